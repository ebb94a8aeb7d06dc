"""BASELINE config 2 end to end: N = 65 536, FLOAT64 mode, fp32 initial conditions, 2 000 ticks on the GPU next to
the CPU oracle's OpenMP fast path (the reference itself cannot run at this N: 68.7 GB per N x N temporary).
Prints the energy drift of both every 500 ticks, their difference, and the position / velocity deviation.
~8 minutes of host time on 16 cores; run once per round, result recorded in DESIGN.md section 7.

    python tools/config2_energy_check.py [ticks=2000]
"""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.abspath(__file__).rsplit("/", 3)[0])
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import galaxy
from oracle import oracle as O

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = 65536
pos, vel, mass = galaxy.create_disk_galaxy(n, seed=42, device="cpu")
# fp32 initial conditions like main.py:131-133, handed over fp64-typed so that both sides run the fp64 fast
# path from the first force on (the fp32-typed first evaluation is covered at N = 5000 / 65 536 by
# tests/test_gpu_parity.py; the oracle's generic dtype model would need minutes per N^2 pass here)
pos, vel, mass = pos.double(), vel.double(), mass.double()
sim = nb.GalaxySimulation(pos.cuda(), vel.cuda(), mass.cuda(), precision_mode=nb.PrecisionMode.FLOAT64)
p64, v64, m64 = pos.numpy().copy(), vel.numpy().copy(), mass.numpy().copy()
a64 = O.accelerations_f64_fast(p64, m64)


def cpu_energy():
    return 0.5 * float((m64 * (v64 ** 2).sum(1)).sum()) + O.potential_energy_f64_fast(p64, m64)


e0_gpu, e0_cpu = sim.get_total_energy(), cpu_energy()
print(f"tick 0: E gpu {e0_gpu:.12e}  cpu {e0_cpu:.12e}  rel diff {abs(e0_gpu - e0_cpu) / abs(e0_cpu):.2e}", flush=True)
lib = O.lib()
done = 0
t0 = time.perf_counter()
while done < ticks:
    k = min(100, ticks - done)
    sim.run(k)
    rest = k
    lib.nbo_step_f64_fast(n, 2, O._dp(p64), O._dp(v64), O._dp(m64), O._dp(a64), 0.001, 0.1 ** 2, 0.01, rest)
    done += k
    if done % 500 == 0 or done == ticks:
        e_cpu = cpu_energy()
        e_gpu = sim.get_total_energy()
        d_gpu, d_cpu = (e_gpu - e0_gpu) / abs(e0_gpu), (e_cpu - e0_cpu) / abs(e0_cpu)
        gp, gv = sim.positions.cpu().numpy(), sim.velocities.cpu().numpy()
        print(f"tick {done}: drift gpu {d_gpu:+.9e}  cpu {d_cpu:+.9e}  |diff| {abs(d_gpu - d_cpu):.2e}   "
              f"max|dx|/max|x| {np.abs(gp - p64).max() / np.abs(p64).max():.2e}  "
              f"max|dv|/max|v| {np.abs(gv - v64).max() / np.abs(v64).max():.2e}   ({time.perf_counter() - t0:.0f} s)", flush=True)
    else:
        print(f"tick {done} ({time.perf_counter() - t0:.0f} s)", flush=True)
