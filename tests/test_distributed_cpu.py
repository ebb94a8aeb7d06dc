"""world_size-2 `gloo` tests of the multi-GPU plumbing, on CPU.

The device work of a rank (partial forces over its source block) is stood in for by the
oracle's j-range partial sums; what is under test is the host logic every rank runs:
runtime.init_distributed / shard_range / the unique-id broadcast, and that block partials
all-reduced with SUM reproduce the full force (the invariant nb_step relies on).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nbody_cosmological_simulation_amd import runtime
        from oracle import oracle as O
        r, w = runtime.init_distributed()
        assert (r, w) == (rank, world)
        uid = runtime.exchange_unique_id(draw=lambda: bytes(range(128)))
        assert uid == bytes(range(128))                      # every rank holds rank 0's id

        g = np.load(os.path.join(ROOT, "tests", "golden", "g1_n257_d2_e0.05.npz"))
        pos, mass = g["pos"].astype(np.float64), g["mass"].astype(np.float64)
        n = pos.shape[0]
        j0, j1 = runtime.shard_range(n, rank, world)
        part = O.accelerations_f64_fast(pos, mass, softening=0.05, j_range=(j0, j1))
        t = torch.from_numpy(part.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)             # C1: per-particle force vectors
        full = O.accelerations_f64_fast(pos, mass, softening=0.05)
        err = np.abs(t.numpy() - full).max() / np.abs(full).max()

        # C2: global max of r2 across source blocks (grid modes)
        p32 = g["pos"]
        d = p32[None, j0:j1, :] - p32[:, None, :]
        r2max = torch.tensor([float(((d * d).sum(-1)).max())])
        dist.all_reduce(r2max, op=dist.ReduceOp.MAX)
        dd = p32[None, :, :] - p32[:, None, :]
        ok_max = float(r2max) == float((dd * dd).sum(-1).max())

        # C3: potential-energy partials
        pe = torch.tensor([O.lib().nbo_potential_energy(n, 2, O.F64, O._dp(pos), O.F64, O._dp(mass), 0.001,
                                                        0.05 ** 2, j0, j1)], dtype=torch.float64)
        dist.all_reduce(pe)
        pe_full = O.potential_energy_f64_fast(pos, mass, softening=0.05)
        out[rank] = (float(err), bool(ok_max), abs(float(pe) - pe_full) / abs(pe_full), j0, j1)
        runtime.reset_distributed()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_j_block_sharding_with_gloo(world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    edges = sorted((v[3], v[4]) for v in out.values())
    assert edges[0][0] == 0 and edges[-1][1] == 257
    for rank in range(world):
        err, ok_max, pe_err, _, _ = out[rank]
        assert err < 1e-14, "all-reduced block partials must equal the full force"
        assert ok_max
        assert pe_err < 1e-14


# --------------------------------------------------------------------------- the partition actually used
def _plan(n, dim, rank, world, is_f64=True, multi=True, mode=0, cus=256):
    """Work plan of one rank through the device-free C-ABI entry nb_plan_debug (what nb_set_state uploads)."""
    import ctypes as C
    from nbody_cosmological_simulation_amd import _native as N
    L = N.lib()
    cfg = N.NbConfig(n=n, dim=dim, mode=mode, levels=0, G=1e-3, softening_sq=0.01, dt=0.01, device=0, rank=rank,
                     nranks=world, flags=0)
    info = (C.c_int32 * 16)()
    N.check(L.nb_plan_debug(C.byref(cfg), int(is_f64), int(multi), cus, info, None, 0, None, None, None, None, None))
    keys = ["enabled", "r", "tile_b", "tiles", "np", "nwork", "nslots", "ncol", "cl", "nchunks", "col_mib", "row_mib"]
    out = dict(zip(keys, list(info)))
    if not out["enabled"]:
        return out
    work = np.zeros((out["nwork"], 8), np.int32)
    arr = {k: np.zeros(out["tiles"], np.int32) for k in ("row_slot0", "row_nslots", "col_upto")}
    cw = np.zeros(out["nchunks"] + 1, np.int32)
    ct = np.zeros(out["nchunks"] + 1, np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    N.check(L.nb_plan_debug(C.byref(cfg), int(is_f64), int(multi), cus, info, ip(work), out["nwork"], ip(arr["row_slot0"]),
                            ip(arr["row_nslots"]), ip(arr["col_upto"]), ip(cw), ip(ct)))
    out.update(arr, work=work, chunk_work=cw, chunk_tile=ct)
    return out


@pytest.mark.parametrize("n,is_f64", [(9000, True), (65536, True), (262144, False), (20481, True), (1024, True), (1100, False),
                                      # mid sizes whose sweeps the cost model cuts into 3 / 5 / 6 pieces (unequal lengths)
                                      (8192, True), (6656, True), (5632, True), (12288, False)])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_symmetric_plans_cover_every_tile_pair_once(monkeypatch, n, is_f64, world):
    """The multi-GPU partition of the headline sizes is snake-dealt target super-rows with pair symmetry
    (nb_plan.cpp), not j-blocks.  Without a GPU: the union over ranks of the work items must sweep every tile
    pair (I, J >= I) through all 64 rotation steps exactly once; row slots must be disjoint and match the
    reduction's index tables; column-slab prefixes must hold exactly the entries a tile needs."""
    plans = [_plan(n, 2, r, world, is_f64=is_f64) for r in range(world)]
    if not plans[0]["enabled"]:
        # fewer super-rows than ranks: EVERY rank falls back to the one-sided source blocks
        # (n <= 1100: tiles of 64 / 128; the fp64 mid sizes: tiles of 256 -> N = 6656 has 7 super-rows, 5632 has 6)
        sr_max = -(-(-(-n // 256)) // 4) if n > 1100 else 5
        assert world > sr_max or (n <= 1100 and world >= 3)
        assert not any(p["enabled"] for p in plans)
        return
    assert all(p["enabled"] for p in plans)
    p0 = plans[0]
    B, tiles = p0["tile_b"], p0["tiles"]
    T = (n + B - 1) // B
    assert tiles % 4 == 0 and tiles >= T and p0["np"] == tiles * B
    steps = np.zeros((T, T), np.uint64)                 # bit s set: rotation step s of (I, J) is covered
    pairs_of_rank = []
    for p in plans:
        assert (p["tile_b"], p["tiles"], p["nchunks"]) == (B, tiles, 1)
        assert list(p["chunk_tile"]) == [0, tiles] and list(p["chunk_work"]) == [0, p["nwork"]]
        w = p["work"]
        used_slots = np.zeros(p["nslots"], np.int32)
        npairs = 0
        col_of_row = {}
        for idx, (ti, jb, je, slot, stride, col, sb, sc) in enumerate(w):
            rowsplit = stride < 0        # one target tile per workgroup, its four waves share the steps [sb, sb + sc)
            assert (rowsplit or ti % 4 == 0) and ti <= jb < je <= T and 0 <= sb and sc > 0 and sb + sc <= 64
            mask = np.uint64(((1 << int(sc)) - 1) << int(sb))
            for wv in range(1 if rowsplit else 4):
                I = ti + wv
                sl = slot if rowsplit else slot + wv * stride
                assert 0 <= sl < p["nslots"]
                used_slots[sl] += 1
                assert p["row_slot0"][I] <= sl < p["row_slot0"][I] + p["row_nslots"][I]
                if I >= T:
                    continue
                for J in range(max(jb, I), je):
                    assert steps[I, J] & mask == 0, f"tile pair ({I},{J}) swept twice"
                    steps[I, J] |= mask
                    npairs += int(sc)
            col_of_row.setdefault(int(ti), set()).add(int(col))
        assert np.all(used_slots == 1), "row slots must be written exactly once"
        assert int(p["row_nslots"].sum()) == p["nslots"]
        # column-slab prefix of tile J = entries of the owned super-rows that start above J
        for J in range(tiles):
            need = set()
            for ti, cols in col_of_row.items():
                if ti < J:
                    need |= cols
            assert need == set(range(int(p["col_upto"][J]))), f"col_upto[{J}]"
        pairs_of_rank.append(npairs)
    full = np.uint64(0xFFFFFFFFFFFFFFFF)
    iu = np.triu_indices(T)
    assert np.all(steps[iu] == full), "every tile pair (I, J >= I) must be swept through all 64 steps"
    assert np.all(steps[np.tril_indices(T, -1)] == 0)
    # snake dealing: equal pair work per rank to within one super-row
    assert max(pairs_of_rank) - min(pairs_of_rank) <= 4 * T * 64


@pytest.mark.parametrize("split", [3, 5, 7, 11, 16])
def test_sweep_pieces_need_not_divide_64(monkeypatch, split):
    """NB_SYM_SPLIT = any 1 ... 16: the pieces of a sweep are [64 q / nsp, 64 (q + 1) / nsp) -- together all 64 rotation
    steps of every work item, lengths differing by at most one."""
    monkeypatch.setenv("NB_SYM_SPLIT", str(split))
    p = _plan(8192, 2, 0, 1, is_f64=True)
    assert p["enabled"] and p["nwork"] % split == 0
    by_item = {}
    for ti, jb, je, slot, stride, col, sb, sc in p["work"]:
        by_item.setdefault((int(ti), int(jb), int(je)), []).append((int(sb), int(sc)))
    for pieces in by_item.values():
        pieces.sort()
        assert len(pieces) == split and pieces[0][0] == 0 and pieces[-1][0] + pieces[-1][1] == 64
        assert all(a[0] + a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
        lens = [c for _, c in pieces]
        assert max(lens) - min(lens) <= 1


def test_mid_size_pieces_follow_the_cost_model():
    """fp64 mid sizes: classic pieces per sweep or row-split work items, chosen by the measured cost model of nb_plan.cpp
    (profiles/r03_mid_split_sweep.txt, profiles/r03_rowsplit_sweep.txt).  Row-split (one target tile per workgroup, its
    four waves share the steps, slot_stride < 0) only for FLOAT64 plans in 2-D; large systems keep whole sweeps."""
    def shape(n, **kw):
        p = _plan(n, kw.pop("dim", 2), 0, 1, is_f64=kw.pop("is_f64", True))
        w = p["work"]
        items = {(int(x[0]), int(x[1])) for x in w}
        return ("rowsplit" if int(w[0][4]) < 0 else "classic", p["nwork"] // len(items))
    assert shape(5120) == ("rowsplit", 2) and shape(8192) == ("rowsplit", 2)
    assert shape(9216) == ("rowsplit", 1) and shape(12288) == ("rowsplit", 1) and shape(16384) == ("rowsplit", 1)
    assert shape(14336) == ("classic", 3)
    assert shape(65536)[0] == "classic" and shape(24576)[0] == "classic"
    assert shape(8192, dim=3)[0] == "classic" and shape(8192, is_f64=False)[0] == "classic"


def test_rowsplit_knob(monkeypatch):
    """NB_SYM_ROWSPLIT=0 keeps the classic work items, n > 0 forces row-split items with n step pieces."""
    monkeypatch.setenv("NB_SYM_ROWSPLIT", "0")
    assert int(_plan(8192, 2, 0, 1, is_f64=True)["work"][0][4]) > 0
    monkeypatch.setenv("NB_SYM_ROWSPLIT", "3")
    p = _plan(8192, 2, 0, 1, is_f64=True)
    assert all(int(w[4]) < 0 for w in p["work"]) and p["nwork"] == 3 * (32 * 33 // 2)


def _p2p_vote_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nbody_cosmological_simulation_amd import _native as N, runtime
        runtime.init_distributed(device=0)
        # no GPU here: the export fails on every rank; the setup must come back False on every rank (unanimous vote,
        # nobody left waiting in a collective) and leave the library without a direct path
        ok = runtime.attach_direct_allreduce(0, world, rank)
        state = N.lib().nb_comm_p2p_state()
        label = runtime._p2p_log["state"]
        torch.save({"ok": ok, "state": state, "label": label}, os.path.join(out, f"p2p_{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the behaviour without a GPU")
def test_direct_allreduce_setup_votes_no_without_a_gpu(tmp_path):
    """runtime.attach_direct_allreduce between two gloo ranks on a machine without a GPU: every rank's export fails,
    the all-gathered verdict is a unanimous no, no rank hangs, RCCL stays the carrier."""
    world, port, out = 2, _free_port(), str(tmp_path)
    mp.spawn(_p2p_vote_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(os.path.join(out, f"p2p_{r}.pt"))
        assert got["ok"] is False and got["state"] == 0
        assert got["label"].startswith("disabled")


def test_plan_choice_is_rank_independent():
    """Every rank must take the same decision between the symmetric plan and the one-sided source blocks."""
    for n in (700, 3000, 4096, 9000, 40000):
        for world in (2, 5, 8):
            for is_f64 in (True, False):
                en = {_plan(n, 2, r, world, is_f64=is_f64)["enabled"] for r in range(world)}
                assert len(en) == 1, (n, world, is_f64)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without RANK in the environment must start the N ranks itself (child
    torch.distributed.run, rendezvous on 127.0.0.1) and relay rank 0's JSON line -- the driver's command shape."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"dry_run": True, "n_gpus": 2, "gpus_arg": 2}
