"""Single-GPU runs at the sizes of BASELINE configs 4 / 5 (fp32 state): throughput, energy drift and a finite-state
check over a few hundred ticks.  (The configs themselves are 8-GPU runs; this is the one-GPU evidence.)

    python tools/large_n_run.py N TICKS [mode=float32]
"""
import sys, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import galaxy

n, ticks = int(sys.argv[1]), int(sys.argv[2])
mode = nb.get_mode_from_string(sys.argv[3] if len(sys.argv) > 3 else "float32")
pos, vel, mass = galaxy.create_disk_galaxy(n, seed=42, device="cpu")
sim = nb.GalaxySimulation(pos.cuda(), vel.cuda(), mass.cuda(), precision_mode=mode, profile=True)
e0 = sim.get_total_energy()
print(f"N = {n}, mode {mode.value}, kernel {sim.force_kernel_name()}, E0 = {e0:.9e}", flush=True)
done, t_run = 0, 0.0
sim.kernel_time()
while done < ticks:
    k = min(max(1, ticks // 4), ticks - done)
    t = time.perf_counter(); sim.run(k); sim.synchronize(); t_run += time.perf_counter() - t
    done += k
    e = sim.get_total_energy()
    p = sim.positions
    print(f"tick {done}: drift {(e - e0) / abs(e0):+.4e}  finite {bool(torch.isfinite(p).all())}  "
          f"max|x| {float(p.abs().max()):.2f}  ({t_run:.1f} s of stepping)", flush=True)
ms, launches = sim.kernel_time()
flops = 14.0 * n * n
peak = 78.6 if mode == nb.PrecisionMode.FLOAT64 else 157.3
print(f"{t_run / ticks * 1e3:.2f} ms/step, {n * ticks / t_run:.3e} particle-steps/s, force kernel {ms / launches:.2f} ms = "
      f"{flops / (ms / launches * 1e-3) / 1e12:.1f} TFLOP/s ({flops / (ms / launches * 1e-3) / 1e12 / peak * 100:.1f} % of peak)")
