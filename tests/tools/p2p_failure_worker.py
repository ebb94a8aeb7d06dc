"""Worker of test_direct_allreduce_failure_paths: two processes share ONE GPU (torchrun, gloo transport).

scenario "dead-peer":  both ranks build the same multi-rank simulation on the direct-only communicator (NB_COMM=direct) with
    a short barrier timeout (NB_P2P_TIMEOUT_S); rank 0 then takes a step that rank 1 never joins.  Rank 0's kernels must
    leave their barrier after the timeout, and EVERY entry point that hands state or scalars back (positions,
    energy, metrics, synchronize) must raise NB_ERR_COMM instead of returning garbage.  After the collective
    shutdown a simulation without a communicator steps bit-identically to one taken before the failure.
scenario "vote-no":    the self-test "fails" on rank 1 (NB_TEST_P2P_FAIL_RANK): every rank must come back with the same
    verdict -- direct path off, nobody left waiting in a collective -- and the single-GPU engine is untouched.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ["NB_ROOT"])
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
scenario = os.environ["NB_SCENARIO"]
dist.init_process_group("gloo")
os.environ["NB_COMM"] = "direct"
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import _native as N, galaxy, runtime

dev = torch.device("cuda", 0)
pos, vel, mass = galaxy.create_disk_galaxy(9000, seed=5, device="cpu")
out = {"scenario": scenario}


def single_run():
    runtime.reset_distributed()
    s = nb.GalaxySimulation(pos, vel, mass, precision_mode=nb.PrecisionMode.FLOAT64, device=dev)
    s.run(3)
    x = s.positions.cpu().numpy().copy()
    s.close()
    return x


before = single_run()
runtime.init_distributed(device=0)
if scenario == "dead-peer":
    multi = nb.GalaxySimulation(pos, vel, mass, precision_mode=nb.PrecisionMode.FLOAT64, device=dev)    # collective: fine
    multi.run(1)                                   # a step both ranks take: fine
    ok_x = multi.positions.cpu().numpy().copy()
    dist.barrier()
    if rank == 0:
        t0 = time.perf_counter()
        multi.run(1)                               # rank 1 never joins this one
        errors = {}
        for name, call in (("positions", lambda: multi.positions), ("energy", multi.get_total_energy),
                           ("synchronize", multi.synchronize)):
            try:
                call()
                errors[name] = "no error"
            except N.NativeError as e:
                errors[name] = [e.code, str(e)]
        out["errors"] = errors
        out["seconds"] = time.perf_counter() - t0
    else:
        time.sleep(float(os.environ["NB_P2P_TIMEOUT_S"]) * 3 + 2.0)
    dist.barrier()
    multi.close()
    out["finite_before_failure"] = bool(np.isfinite(ok_x).all())
else:
    try:
        nb.GalaxySimulation(pos, vel, mass, precision_mode=nb.PrecisionMode.FLOAT64, device=dev)
        out["setup"] = "unexpectedly succeeded"
    except RuntimeError as e:                      # NB_COMM=direct has no RCCL to fall back on: the constructor says so
        out["setup"] = str(e)
    out["p2p_state"] = N.lib().nb_comm_p2p_state()
    out["log"] = runtime._p2p_log["state"]
    dist.barrier()
runtime.shutdown()
after = single_run()
out["single_gpu_unaffected"] = bool(np.array_equal(before, after))
gathered = [None] * world
dist.all_gather_object(gathered, out)
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    json.dump(gathered, open(os.environ["NB_OUT"], "w"))
    print("P2P-FAILURE-OK", scenario)
