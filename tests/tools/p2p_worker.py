"""Worker of test_direct_allreduce_between_processes: W processes (torchrun, gloo transport) share ONE GPU and run the
direct all-reduce of csrc/nb_p2p.hip between them -- HIP IPC handles, per-workgroup epoch barriers, rank-ordered sums.
The peers' memory is then local HBM instead of xGMI, everything else (protocol, ordering, process boundaries) is what an
8-GPU node runs.  Every result is compared bit for bit with the rank-ordered sum computed on the host."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ["NB_ROOT"])
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
import ctypes as C
from nbody_cosmological_simulation_amd import _native as N, runtime

runtime.init_distributed(device=0)
L = N.lib()
cap = 1 << 20
assert runtime.attach_direct_allreduce(0, world, rank, capacity_bytes=cap, rounds=2, timeout_s=10.0), runtime._p2p_log
assert L.nb_comm_p2p_state() == 2


def gather(a):
    out = [None] * world
    dist.all_gather_object(out, a)
    return out


rng = np.random.default_rng(100 + rank)
cases = [(np.float64, 1), (np.float64, 7), (np.float64, 1000), (np.float64, 131072), (np.float64, cap // 8),
         (np.float32, 2), (np.float32, 6), (np.float32, 196608), (np.float32, cap // 4)]
for it in range(12):
    for dt, count in cases:
        if it >= 2 and count > 1000:
            continue                                   # the long vectors twice, the short ones every round
        x = (rng.standard_normal(count) * 10 ** rng.uniform(-3, 3)).astype(dt)
        parts = gather(x)
        want = parts[0].copy()
        for q in range(1, world):
            want = want + parts[q]                     # rank order, one rounding per addition: what the kernel does
        got = x.copy()
        N.check(L.nb_comm_p2p_allreduce(got.ctypes.data_as(C.c_void_p), count, N.NB_F64 if dt == np.float64 else N.NB_F32, 10.0))
        assert np.array_equal(got, want), (it, dt, count, int((got != want).sum()))
# back-to-back launches without host synchronisation in between are covered by the self-test (15 launches per round)
N.check(L.nb_comm_p2p_selftest(4, 10.0))
us = C.c_double(0.0)
N.check(L.nb_comm_allreduce_time(None, 1, 300, C.byref(us)))
lat = gather(us.value)
if rank == 0:
    print("P2P-LATENCY-US 1MiB same-GPU", world, "ranks:", " ".join(f"{v:.1f}" for v in lat))
# errors: too long, odd fp32 count
bad = np.zeros(cap // 8 + 1)
assert L.nb_comm_p2p_allreduce(bad.ctypes.data_as(C.c_void_p), bad.size, N.NB_F64, 1.0) != 0
odd = np.zeros(3, np.float32)
assert L.nb_comm_p2p_allreduce(odd.ctypes.data_as(C.c_void_p), 3, N.NB_F32, 1.0) != 0
dist.barrier()
runtime.shutdown()
assert L.nb_comm_p2p_state() == 0
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("P2P-OK", world)
