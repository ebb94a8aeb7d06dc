"""Process-level runtime context: which HIP device this process drives and, under
`torch.distributed` (one process per GPU), which block of sources it owns.

Multi-GPU layout (DESIGN.md "multi-GPU"): every rank holds the full O(N) state, computes
partial accelerations of ALL targets over its contiguous source block
[rank*N/P, (rank+1)*N/P) and the per-particle force vectors are summed with one RCCL
all-reduce per step inside libnbody_amd (nb_comm_init).  torch.distributed is plumbing only:
it carries the 128-byte RCCL unique id from rank 0 to the other ranks.
"""
import ctypes as C
import os

from . import _native as N

_ctx = {"rank": 0, "world": 1, "device": None, "group": None}


def default_hip_device() -> int:
    if _ctx["device"] is not None:
        return _ctx["device"]
    for key in ("NBODY_DEVICE", "LOCAL_RANK"):
        if key in os.environ:
            try:
                return int(os.environ[key])
            except ValueError:
                pass
    return 0


def shard_range(n: int, rank: int, world: int):
    """Contiguous source block of `rank` -- the same integer arithmetic as nb_api.cpp:compute_geometry."""
    return (rank * n) // world, ((rank + 1) * n) // world


def init_distributed(device: int = None, group=None):
    """Adopt the current torch.distributed process group (if any) for new simulations."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        _ctx["rank"] = dist.get_rank(group)
        _ctx["world"] = dist.get_world_size(group)
        _ctx["group"] = group
    else:
        _ctx["rank"], _ctx["world"], _ctx["group"] = 0, 1, None
    if device is not None:
        _ctx["device"] = int(device)
    return _ctx["rank"], _ctx["world"]


def reset_distributed():
    _ctx.update(rank=0, world=1, device=None, group=None)


def rank_world():
    return _ctx["rank"], _ctx["world"]


def force_comm() -> bool:
    """NBODY_FORCE_COMM=1: create the RCCL communicator even for one rank, so a single-GPU box
    exercises nb_comm_init and the all-reduce calls of the multi-GPU step."""
    return os.environ.get("NBODY_FORCE_COMM", "0") == "1" and _ctx["world"] >= 1 and _dist_ready()


def _dist_ready() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def _draw_unique_id() -> bytes:
    buf = C.create_string_buffer(256)
    size = C.c_int32(256)
    N.check(N.lib().nb_comm_unique_id(buf, C.byref(size)))
    return bytes(buf.raw[: size.value])


def exchange_unique_id(draw=_draw_unique_id) -> bytes:
    """Rank 0 draws an RCCL unique id (nb_comm_unique_id); everyone receives its bytes."""
    import torch.distributed as dist
    payload = [None]
    if _ctx["rank"] == 0:
        payload[0] = draw()
    dist.broadcast_object_list(payload, src=0, group=_ctx["group"])
    return payload[0]
