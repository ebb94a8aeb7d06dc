"""Whole-step time of the grid modes (force kernel + max-r2 search + tables + reduction + force quantisation + kicks):
N = 65 536 (BASELINE config 3), equal and unequal masses, and main.py's default N = 3000 (one-launch path).
NB_NO_TRACK=1 gives the round-2 launch train (pruned search from scratch every step) for an A/B on the same box.

    python tools/grid_step_timing.py
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import galaxy

def timed(n, mode, mass=None, steps=400):
    pos, vel, m = galaxy.create_disk_galaxy(n, seed=42, device="cpu")
    if mass is not None:
        m = mass
    sim = nb.GalaxySimulation(pos.cuda(), vel.cuda(), m.cuda(), precision_mode=nb.get_mode_from_string(mode))
    t = time.perf_counter()
    while time.perf_counter() - t < 0.3:
        sim.run(50); sim.synchronize()
    t = time.perf_counter(); sim.run(steps); sim.synchronize(); dt = time.perf_counter() - t
    return dt / steps, sim.force_kernel_name()

tag = ("NB_NO_TRACK" if os.environ.get("NB_NO_TRACK") else "tracked") + (" " + os.path.basename(os.environ["NBODY_LIB"]) if os.environ.get("NBODY_LIB") else "")
sizes = [int(x) for x in os.environ.get("GRID_TIMING_NS", "65536,1024,2048,3000,6000,12000").split(",")]
for n, steps in ((65536, 400), (1024, 4000), (2048, 4000), (3000, 4000), (6000, 2000), (12000, 1000)):
    if n not in sizes:
        continue
    uneq = 0.5 + torch.rand(n, generator=torch.Generator().manual_seed(1))
    for mode in ("int8", "int4", "custom", "float32"):
        for label, mass in (("equal", None), ("unequal", uneq)):
            if label == "unequal" and (n != 65536 or mode == "float32"):
                continue
            dt, k = timed(n, mode, mass, steps)
            print(f"[{tag}] N={n} {mode:7s} {label:7s}: {dt * 1e6:9.1f} us/step  ({k})", flush=True)
