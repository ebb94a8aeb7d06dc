"""collect_metrics / native_metrics at one size, for `rocprofv3 --kernel-trace --stats -- python3 tools/metrics_trace.py N`."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import galaxy, metrics
n = int(sys.argv[1])
pos, vel, mass = galaxy.create_disk_galaxy(n, seed=1, device="cpu")
sim = nb.GalaxySimulation(pos.cuda(), vel.cuda(), mass.cuda(), precision_mode=nb.PrecisionMode.FLOAT32)
sim.run(2)
m = metrics.SimulationMetrics()
for _ in range(3): metrics.collect_metrics(sim, 0, m)
sim.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    mm = metrics.native_metrics(None, None, None, simulation=sim)
t1 = time.perf_counter()
for _ in range(20):
    sim.run(1); metrics.collect_metrics(sim, 0, m)
sim.synchronize()
t2 = time.perf_counter()
for _ in range(20):
    sim.run(1); sim.get_total_energy()
sim.synchronize()
t3 = time.perf_counter()
print(f"N={n}: native_metrics {(t1-t0)/20*1e6:.1f} us; step + collect_metrics {(t2-t1)/20*1e6:.1f} us; step + total energy {(t3-t2)/20*1e6:.1f} us")
