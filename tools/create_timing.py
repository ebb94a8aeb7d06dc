"""Host-visible cost of short simulations (config 1: N = 1024, 200 ticks; the reference's scripts build many short runs):
construction (nb_create + upload + first force), run(200), reading the state back, close()."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import galaxy
for n in [int(a) for a in sys.argv[1:]] or [1024, 3000, 65536]:
    for mode in ("float64", "float32", "int8"):
        pos, vel, mass = galaxy.create_disk_galaxy(n, seed=1, device="cpu")
        pm = nb.get_mode_from_string(mode)
        sim = nb.GalaxySimulation(pos, vel, mass, precision_mode=pm); sim.run(5); sim.synchronize(); sim.close()   # warm the library
        rows = []
        for rep in range(5):
            t0 = time.perf_counter()
            sim = nb.GalaxySimulation(pos, vel, mass, precision_mode=pm)
            sim.synchronize(); t1 = time.perf_counter()
            sim.run(200); sim.synchronize(); t2 = time.perf_counter()
            st = sim.get_state(); t3 = time.perf_counter()
            sim.close(); t4 = time.perf_counter()
            rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
        best = [min(r[i] for r in rows) * 1e6 for i in range(4)]
        print(f"N={n} {mode:8s}: create {best[0]:8.1f} us  run(200) {best[1]:8.1f} us  get_state {best[2]:7.1f} us  close {best[3]:7.1f} us")
