"""Worker of test_multi_rank_product_path_on_one_gpu: W processes (torchrun, gloo transport) share ONE GPU and run the
real multi-rank step of the engine -- nb_create with nranks = W, the snake-dealt super-row plan of every rank, deferred
kicks, force quantisation after the sum, the potential-energy sum -- with the direct all-reduce as the only carrier
(NB_COMM=direct: RCCL refuses two ranks on one GPU).  Each rank also runs the single-GPU engine on the same input."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.environ["NB_ROOT"])
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
os.environ["NB_COMM"] = "direct"
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import _native as N, galaxy, runtime

dev = torch.device("cuda", 0)
out = {}
cases = (("f64", 9000, nb.PrecisionMode.FLOAT64), ("f32", 9000, nb.PrecisionMode.FLOAT32),
         ("f16", 9000, nb.PrecisionMode.FLOAT16), ("int4", 3000, nb.PrecisionMode.INT4_SIM),
         ("int8_big", 9000, nb.PrecisionMode.INT8_SIM), ("f64_onesided", 3000, nb.PrecisionMode.FLOAT64))
cases += (("f64_n65536", 65536, nb.PrecisionMode.FLOAT64),        # the headline size: production plan of every rank
          ("f64_d3_unequal", 20481, nb.PrecisionMode.FLOAT64),    # 3-D, four targets per lane, general-mass kernel
          ("f32_unequal", 20480, nb.PrecisionMode.FLOAT32))
for name, n, mode in cases:
    pos, vel, mass = galaxy.create_disk_galaxy(n, seed=5, device="cpu")
    if name == "f64_d3_unequal":
        g = torch.Generator().manual_seed(7)
        pos = torch.cat([pos, 0.3 * torch.randn(n, 1, generator=g)], 1)
        vel = torch.cat([vel, torch.zeros(n, 1)], 1)
    if "unequal" in name:
        mass = 0.5 + torch.rand(n, generator=torch.Generator().manual_seed(8))
    runtime.reset_distributed()
    single = nb.GalaxySimulation(pos, vel, mass, precision_mode=mode, device=dev)
    single.run(3); single.run(2)
    runtime.init_distributed(device=0)
    multi = nb.GalaxySimulation(pos, vel, mass, precision_mode=mode, device=dev)
    multi.run(3); multi.run(2)                      # two native calls: the second starts from settled dtypes
    p1, p2 = single.positions.cpu().numpy().astype(np.float64), multi.positions.cpu().numpy().astype(np.float64)
    v1, v2 = single.velocities.cpu().numpy().astype(np.float64), multi.velocities.cpu().numpy().astype(np.float64)
    out[name] = {"relerr_x": float(np.abs(p1 - p2).max() / np.abs(p1).max()),
                 "relerr_v": float(np.abs(v1 - v2).max() / np.abs(v1).max()),
                 "energy": [single.get_total_energy(), multi.get_total_energy()],
                 "kernel": multi.force_kernel_name(),
                 "same_bits": bool(np.array_equal(single.positions.cpu().numpy(), multi.positions.cpu().numpy()) and
                                   np.array_equal(single.velocities.cpu().numpy(), multi.velocities.cpu().numpy())),
                 "hash": hashlib.sha256(multi.positions.cpu().numpy().tobytes() + multi.velocities.cpu().numpy().tobytes()).hexdigest()}
    single.close(); multi.close()
assert N.lib().nb_comm_ready() == world and N.lib().nb_comm_p2p_state() == 2
label = runtime.allreduce_label()
gathered = [None] * world
dist.all_gather_object(gathered, out)
runtime.shutdown()
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    json.dump({"ranks": gathered, "label": label}, open(os.environ["NB_OUT"], "w"))
    print("MULTIRANK-OK", world)
