"""One grid-mode run at a chosen size, for `rocprofv3 --kernel-trace --stats -- python3 tools/trace_mid.py N MODE`
(per-kernel durations of a mid / small-N step: profiles/README.md, round 3)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import galaxy
n = int(sys.argv[1]); mode = sys.argv[2]
pos, vel, m = galaxy.create_disk_galaxy(n, seed=42, device="cpu")
sim = nb.GalaxySimulation(pos.cuda(), vel.cuda(), m.cuda(), precision_mode=nb.get_mode_from_string(mode))
sim.run(300); sim.synchronize()
t=time.perf_counter(); sim.run(500); sim.synchronize(); print(n, mode, (time.perf_counter()-t)/500*1e6, "us/step")
