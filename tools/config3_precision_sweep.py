"""BASELINE config 3: N = 65 536, every precision mode, main.py's flow (fp32 initial conditions, metrics every 100
ticks with the mirrored metrics.py) -- energy drift, r90, bound fraction, velocity dispersion and the rotation
curve against the FLOAT64 run.  The reference cannot run at this N; its own small-N goldens pin the same metrics
in tests/test_gpu_parity.py (config 3 parity test).  Output kept in profiles/.

    python tools/config3_precision_sweep.py [ticks=200]
"""
import sys, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import galaxy, metrics

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = 65536
pos, vel, mass = galaxy.create_disk_galaxy(n, seed=42, device="cpu")
modes = ["float64", "float32", "bfloat16", "float16", "int8", "int4", "custom"]
results, curves = {}, {}
for name in modes:
    mode = nb.get_mode_from_string(name)
    sim = nb.GalaxySimulation(pos.cuda(), vel.cuda(), mass.cuda(), precision_mode=mode)
    m = metrics.SimulationMetrics()
    metrics.collect_metrics(sim, 0, m)
    t = time.perf_counter()
    sim.run(ticks, callback=lambda s, tick: metrics.collect_metrics(s, tick, m), callback_interval=100)
    sim.synchronize()
    dt = time.perf_counter() - t
    results[name] = (m, dt)
    curves[name] = m.rotation_curves[-1]
    sim.close()
print(f"N = {n}, {ticks} ticks, G = 1e-3, softening 0.1, dt 0.01 (wall time includes the metrics every 100 ticks)")
print(f"{'mode':10s} {'E drift':>12s} {'r90':>9s} {'bound':>8s} {'sigma_v':>10s} {'curve: mean dv':>15s} {'max |dv|/v':>11s} {'s':>7s}")
base = curves["float64"]
for name in modes:
    m, dt = results[name]
    e0, e1 = m.total_energy[0], m.total_energy[-1]
    cmp_ = metrics.compare_rotation_curves(base, curves[name])
    v0, v1 = np.array(base["velocities"]), np.array(curves[name]["velocities"])
    ok = ~(np.isnan(v0) | np.isnan(v1)) & (np.array(base["num_stars_per_bin"]) >= 50)
    print(f"{name:10s} {(e1 - e0) / abs(e0):+12.4e} {m.galaxy_radius_90[-1]:9.4f} {m.bound_fraction[-1]:8.4f} "
          f"{m.velocity_dispersion[-1]:10.6f} {cmp_['mean_velocity_diff']:+15.3e} "
          f"{np.abs(v1[ok] - v0[ok]).max() / np.abs(v0[ok]).max():11.2e} {dt:7.2f}")
