"""A/B of the grid modes' table-free pair path against the table path (NB_NO_GRID_FAST) on single evaluations, and both
against the oracle.  Test infrastructure (it uses the oracle): run from the repo root on a GPU box,
    python tests/tools/grid_fast_ab.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import galaxy
from oracle import oracle as O
for mode, n, seed in (("custom", 9000, 78), ("int4_sim", 9000, 78), ("custom", 30000, 3)):
    pos, vel, mass = galaxy.create_disk_galaxy(n, seed=seed, device="cpu")
    os.environ.pop("NB_NO_GRID_FAST", None)
    a = nb.GalaxySimulation(pos, vel, mass, precision_mode=nb.PrecisionMode(mode))
    da = a.quant_debug()
    os.environ["NB_NO_GRID_FAST"] = "1"
    b = nb.GalaxySimulation(pos, vel, mass, precision_mode=nb.PrecisionMode(mode))
    db = b.quant_debug()
    fa, fb = a.accelerations.numpy().astype(np.float64), b.accelerations.numpy().astype(np.float64)
    d = np.abs(fa - fb)
    i = np.unravel_index(d.argmax(), d.shape)
    print(mode, n, "fast", da["fast_path"], db["fast_path"], "max|da|", d.max(), "rel", d.max() / np.abs(fb).max(), "at", i, fa[i], fb[i],
          "n>1e-5rel", int((d > 1e-5 * np.abs(fb).max()).sum()))
    if n <= 9000:
        ref, dbg = O.accelerations(pos.numpy(), mass.numpy(), mode, debug=True)
        print("   vs oracle: fast", np.abs(fa - ref).max() / np.abs(ref).max(), "slow", np.abs(fb - ref).max() / np.abs(ref).max())
