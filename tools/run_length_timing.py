import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import galaxy
for n, mode in ((3000, "float32"), (3000, "float64"), (1024, "float32"), (4096, "float32")):
    pos, vel, mass = galaxy.create_disk_galaxy(n, seed=1, device="cpu")
    sim = nb.GalaxySimulation(pos.cuda(), vel.cuda(), mass.cuda(), precision_mode=nb.get_mode_from_string(mode))
    sim.run(200); sim.synchronize()
    out = []
    for k in (100, 500, 1000, 2000, 5000, 1000):
        t0 = time.perf_counter(); sim.run(k); sim.synchronize(); out.append(f"{k}: {(time.perf_counter() - t0) / k * 1e6:.1f}")
    print(f"N={n} {mode} us/step by run length ->", ", ".join(out), sim.force_kernel_name())
