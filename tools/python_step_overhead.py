"""Host-side cost of driving the engine step by step from Python (what realtime_visual.py / the experiment scripts of the
reference do: `sim.step()` in a loop, reading `sim.positions` now and then) against one `run(n)` call."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import galaxy

for n in (1024, 3000, 8192, 65536):
    pos, vel, mass = galaxy.create_disk_galaxy(n, seed=1, device="cpu")
    sim = nb.GalaxySimulation(pos.cuda(), vel.cuda(), mass.cuda(), precision_mode=nb.PrecisionMode.FLOAT32)
    sim.run(200); sim.synchronize()
    k = 2000 if n < 10000 else 300
    t0 = time.perf_counter(); sim.run(k); sim.synchronize(); t_run0 = (time.perf_counter() - t0) / k
    t0 = time.perf_counter(); sim.run(k); sim.synchronize(); t_run = (time.perf_counter() - t0) / k
    t0 = time.perf_counter()
    for _ in range(k):
        sim.step()
    sim.synchronize(); t_step = (time.perf_counter() - t0) / k
    t0 = time.perf_counter()
    for _ in range(200):
        sim.step()
        p = sim.positions
    sim.synchronize(); t_read = (time.perf_counter() - t0) / 200
    print(f"N={n}: run(k) first {t_run0 * 1e6:.1f} then {t_run * 1e6:.1f} us/step; step() loop {t_step * 1e6:.1f} us/step; step() + positions read {t_read * 1e6:.1f} us/step")
