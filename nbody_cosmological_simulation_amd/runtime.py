"""Process-level runtime context: which HIP device this process drives and, under
`torch.distributed` (one process per GPU), which block of sources it owns.

Multi-GPU layout (DESIGN.md "multi-GPU"): every rank holds the full O(N) state and computes
partial accelerations of ALL particles over its share of the pair work -- snake-dealt target
super-rows of the pair-symmetric kernel (the partition at every benchmark size), or a contiguous
source block [rank*N/P, (rank+1)*N/P) on the one-sided kernels (small N, fp64 state under a cast
mode) -- and the per-particle force vectors are summed with one all-reduce per step (RCCL, or the
direct xGMI kernel of csrc/nb_p2p.hip for short vectors) inside
libnbody_amd.  torch.distributed is plumbing only: it carries the 128-byte RCCL unique id from
rank 0 to the other ranks, once per process: all simulations of a process share ONE communicator
(nb_comm_init), which only `shutdown()` destroys (collective; never a garbage collector).
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


def partition_label(world: int = None) -> str:
    """Human-readable name of the pair-work partition (bench.py / logs)."""
    world = _ctx["world"] if world is None else world
    if world <= 1:
        return "single GPU"
    return (f"snake-dealt target super-rows (pair-symmetric) x{world}; one-sided kernels: source j-blocks "
            f"+ one all-reduce of the force vectors per step")


def attach_communicator(handle):
    """Give `handle` the process communicator, creating it on first use (collective: every rank creates its
    first communicating simulation at the same point of the program, like any other collective)."""
    L = N.lib()
    if L.nb_comm_ready() > 0:
        N.check(L.nb_comm_init(handle, None, 0))
        return
    if os.environ.get("NB_COMM") == "direct":
        # no RCCL at all: the ranks of one node (also several processes sharing one GPU, which RCCL refuses) sum
        # everything through the direct all-reduce; vectors above its capacity are an error
        world = _ctx["world"] if _dist_ready() else 1
        ok = attach_direct_allreduce(_ctx["device"] if _ctx["device"] is not None else default_hip_device(), world,
                                     _ctx["rank"] if world > 1 else 0)
        if not ok:
            raise RuntimeError("NB_COMM=direct, but the direct all-reduce could not be set up: " + _p2p_log["state"])
        N.check(L.nb_comm_init(handle, None, 0))
        return
    uid = exchange_unique_id()
    N.check(L.nb_comm_init(handle, uid, len(uid)))
    # The direct xGMI all-reduce is OPT-IN (NB_P2P=auto / force): it has only ever run between virtual ranks and
    # between processes sharing one GPU's memory system -- until a run over real xGMI links is
    # on record, RCCL carries every step by default.  auto: self-test + vote + timing comparison; force: no comparison.
    p2p = os.environ.get("NB_P2P", "0")
    if os.environ.get("NB_NO_P2P") is None and p2p in ("auto", "force", "1"):
        attach_direct_allreduce(_ctx["device"] if _ctx["device"] is not None else default_hip_device(),
                                L.nb_comm_ready(), _ctx["rank"] if L.nb_comm_ready() > 1 else 0,
                                compare_with_rccl=p2p != "force")


P2P_CAPACITY_BYTES = 4 << 20      # force vectors up to N*D = 524 288 doubles; longer ones are bandwidth-bound: RCCL
_p2p_log = {"state": "not attempted"}


def _all_gather(obj, world):
    """All ranks' objects in rank order over torch.distributed (the only transport this package uses)."""
    if world <= 1:
        return [obj]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, obj, group=_ctx["group"])
    return out


def _node_identity() -> str:
    try:
        with open("/proc/sys/kernel/random/boot_id") as f:
            return f.read().strip()
    except OSError:
        import socket
        return socket.gethostname()


def attach_direct_allreduce(device: int, world: int, rank: int, capacity_bytes: int = P2P_CAPACITY_BYTES,
                            rounds: int = 2, timeout_s: float = 2.0, compare_with_rccl: bool = False) -> bool:
    """Set up the direct xGMI all-reduce of libnbody_amd (include/nbody_amd.h, nb_comm_p2p_*) between the ranks of
    one node: export / all-gather / import of the HIP IPC handles, a collective self-test, and a unanimous vote.
    Any failure on any rank leaves every rank on RCCL.  compare_with_rccl: both carriers are then timed on a 1 MiB
    vector (200 back-to-back calls each, slowest rank counts) and the direct path is enabled only if it is the
    faster one on this node (NB_P2P=force skips the comparison, NB_P2P=0 the whole setup).  Collective; returns
    whether the direct path is enabled."""
    L = N.lib()
    if L.nb_comm_p2p_state() != 0:
        return L.nb_comm_p2p_state() == 2
    if world > 8:
        _p2p_log["state"] = "more than 8 ranks"
        return False
    handle = C.create_string_buffer(128)
    size = C.c_int32(128)
    rc = L.nb_comm_p2p_export(device, rank, world, capacity_bytes, handle, C.byref(size))
    mine = {"ok": rc == 0, "node": _node_identity(), "handle": bytes(handle.raw[: size.value]) if rc == 0 else b"",
            "err": "" if rc == 0 else N.last_error()}
    everyone = _all_gather(mine, world)
    ok = all(e["ok"] for e in everyone) and len({e["node"] for e in everyone}) == 1
    if ok:
        blob = b"".join(e["handle"] for e in everyone)
        ok = L.nb_comm_p2p_import(blob, world) == 0
    # every rank must take the same branch from here on: vote before the self-test (which waits on the peers)
    ok = all(_all_gather(bool(ok), world))
    if ok:
        ok = L.nb_comm_p2p_selftest(rounds, timeout_s) == 0
        if os.environ.get("NB_TEST_P2P_FAIL_RANK") == str(rank):
            # test hook: this rank reports a failed self-test -- every rank must then vote the direct path down and
            # leave the step to RCCL (tests/test_gpu_parity.py::test_direct_allreduce_failure_paths)
            ok = False
            mine["err"] = "self-test failure injected on rank %d (NB_TEST_P2P_FAIL_RANK)" % rank
            everyone = _all_gather(mine, world)
        elif os.environ.get("NB_TEST_P2P_FAIL_RANK") is not None:
            everyone = _all_gather(mine, world)
        ok = all(_all_gather(bool(ok), world))
    if not ok:
        errs = [e["err"] for e in everyone if e["err"]] or [N.last_error()]
        _p2p_log["state"] = "disabled: " + errs[0]
    else:
        _p2p_log["state"] = "enabled"
        if compare_with_rccl and L.nb_comm_ready() > 0:
            t = {}
            for name, which in (("direct", 1), ("rccl", 0)):
                us = C.c_double(0.0)
                rc = L.nb_comm_allreduce_time(None, which, 200, C.byref(us))
                t[name] = max(v if v is not None else float("inf")
                              for v in _all_gather(us.value if rc == 0 else None, world))
            _p2p_log.update(direct_us=t["direct"], rccl_us=t["rccl"])
            ok = t["direct"] < t["rccl"]
            _p2p_log["state"] = (f"enabled: 1 MiB all-reduce {t['direct']:.1f} us direct vs {t['rccl']:.1f} us RCCL" if ok else
                                 f"not used: 1 MiB all-reduce {t['direct']:.1f} us direct vs {t['rccl']:.1f} us RCCL")
    N.check(L.nb_comm_p2p_enable(1 if ok else 0))
    return bool(ok)


def allreduce_label() -> str:
    """Which collective carries the force vectors (bench.py / logs)."""
    L = N.lib()
    if L.nb_comm_ready() <= 0:
        return "none"
    if L.nb_comm_p2p_state() == 2 and os.environ.get("NB_COMM") == "direct":
        return "direct loads only (nb_p2p; no RCCL communicator)"
    if L.nb_comm_p2p_state() == 2:
        return f"direct xGMI loads (nb_p2p; {_p2p_log['state']}); RCCL for scalars and long vectors"
    return f"RCCL (direct path {_p2p_log['state']})"


def shutdown():
    """Destroy the process communicator (collective).  Call on every rank after the last step and before
    torch.distributed is torn down; simulations created afterwards would need a new communicator."""
    L = N.lib()
    N.check(L.nb_comm_quiesce())
    if _dist_ready() and _ctx["world"] > 1:
        import torch.distributed as dist
        dist.barrier(group=_ctx["group"])       # no rank frees the shared buffers while a peer's kernel reads them
    N.check(L.nb_comm_shutdown())


def trim_cache() -> int:
    """Release the streams and device allocations that closed simulations left in the library's bounded cache
    (include/nbody_amd.h: nb_cache_trim); returns the bytes of device memory given back."""
    freed = C.c_int64(0)
    N.check(N.lib().nb_cache_trim(C.byref(freed)))
    return int(freed.value)


def rank_world():
    return _ctx["rank"], _ctx["world"]


def force_comm() -> bool:
    """NBODY_FORCE_COMM=1: create the RCCL communicator even for one rank, so a single-GPU box
    exercises nb_comm_init and the all-reduce calls of the multi-GPU step."""
    return os.environ.get("NBODY_FORCE_COMM", "0") == "1"


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
    if _ctx["world"] <= 1 and not _dist_ready():
        return draw()                       # a 1-rank communicator needs no transport
    import torch.distributed as dist
    payload = [None]
    if _ctx["rank"] == 0:
        payload[0] = draw()
    dist.broadcast_object_list(payload, src=0, group=_ctx["group"])
    return payload[0]
