"""Quant-bin assignments read out of the PRODUCTION pair loops.

north_star asks for "bit-identical quant-bin assignments for int8/int4": the index round(normalized * (levels - 1)) of
reference quantization.py:119-121.  `quant_debug` / `quant_bins_rows` answer from a kernel of their own that walks the
exact threshold tables; the force kernels decide differently (v_log_f32 estimate -> rint -> v_exp_f32, one ballot per
wave, threshold fallback only near a bin edge).  `nb_quant_bin_sums` therefore runs the SAME kernel templates the force
path launches -- sweep / sweep_pk of force_sym_kernel, force_f32_kernel, small_step_kernel -- instantiated with an
integer read-out where each pair's bin is decided, and returns per particle p

    s1[p] = sum_q k(p, q)            s2[p] = sum_q k(p, q) * ((q mod 65521) + 1)

(exact int64 sums, order-free).  They are compared here, bit for bit, with the same sums over the rows of the
REFERENCE's own bin matrices (golden g20: tests/golden/make_golden.py g20, from the stored g13 / g16 positions) and,
where the reference has no fixture (CUSTOM grids of 1000 / 4096 levels, adversarial bin-edge layouts), with the
oracle's bin matrix.  Bins depend on positions and softening only, so the same checksums serve runs with equal and
with unequal masses -- which select different kernels (packed uniform-mass sweep vs scalar general-mass sweep).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

GRID = ["int8_sim", "int4_sim", "custom"]


@pytest.fixture(scope="module")
def nb():
    import nbody_cosmological_simulation_amd as pkg
    assert pkg._native.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return pkg


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def checksums(k):
    k64 = np.asarray(k, np.int64)
    w = (np.arange(k64.shape[1], dtype=np.int64) % 65521) + 1
    return k64.sum(axis=1), (k64 * w[None, :]).sum(axis=1)


# kernel variants: environment knobs read at handle creation (DESIGN.md "tuning and test knobs") + what must come out
VARIANTS = {
    # name: (env, which, expectations on the read-out's info)
    "default": ({}, "tiled", {}),
    "sym_r2": ({"NB_SYM": "1", "NB_SYM_R": "2"}, "tiled", {"path": "sym", "shape": 2}),
    "sym_r4": ({"NB_SYM": "1", "NB_SYM_R": "4"}, "tiled", {"path": "sym", "shape": 4}),
    "sym_r4_general": ({"NB_SYM": "1", "NB_SYM_R": "4", "NB_NO_UNIFORM": "1"}, "tiled",
                       {"path": "sym", "shape": 4, "uniform_kernel": False}),
    "sym_r4_tables": ({"NB_SYM": "1", "NB_SYM_R": "4", "NB_NO_GRID_FAST": "1"}, "tiled",
                      {"path": "sym", "shape": 4, "fast_path": False}),
    "sym_r2_tables": ({"NB_SYM": "1", "NB_SYM_R": "2", "NB_NO_GRID_FAST": "1"}, "tiled",
                      {"path": "sym", "shape": 2, "fast_path": False}),
    "sym_r4_split": ({"NB_SYM": "1", "NB_SYM_R": "4", "NB_SYM_SPLIT": "4"}, "tiled", {"path": "sym", "shape": 4}),
    "onesided": ({"NB_SYM": "0"}, "tiled", {"path": "onesided"}),
    "small": ({"NB_SMALL_MAX": "4096"}, "small", {"path": "small"}),
    "small_tables": ({"NB_SMALL_MAX": "4096", "NB_NO_GRID_FAST": "1"}, "small", {"path": "small", "fast_path": False}),
    "small_16": ({"NB_SMALL_MAX": "4096", "NB_SMALL_LANES": "16"}, "small", {"path": "small", "shape": 16}),
}


def _check(bs, want1, want2, expect, n_pairs_min):
    for key, val in expect.items():
        assert bs[key] == val, (key, bs[key], val, bs)
    bad = np.nonzero((bs["sum_k"] != want1) | (bs["sum_kw"] != want2))[0]
    assert bad.size == 0, (f"{bad.size} particles with a wrong bin checksum, first {bad[:5]}: "
                           f"got {bs['sum_k'][bad[:5]]} want {np.asarray(want1)[bad[:5]]}")
    total = bs["pairs_table_free"] + bs["pairs_table"]
    assert total >= n_pairs_min, bs
    if bs["fast_path"] and bs["path"] != "onesided":      # (the one-sided kernel always reads its tables)
        # the table-free route is what bins nearly every pair: that is the decision being pinned here
        assert bs["pairs_table_free"] > 0.5 * total, bs
    else:
        assert bs["pairs_table_free"] == 0, bs


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("masses", ["uniform", "unequal"])
@pytest.mark.parametrize("mode", GRID)
@pytest.mark.parametrize("case", ["n4096_d2", "n2048_d3"])
def test_production_pair_loops_bin_like_the_reference(nb, monkeypatch, case, mode, masses, variant):
    """Every row of the reference's N x N bin matrix (N = 4096 D = 2 disk galaxy, softening 0.1; N = 2048 D = 3,
    softening 0.05: softening^2 below the grid floor, the clamp variant of the table-free path) against the read-out of
    each production kernel: R = 2 / R = 4 tilings, packed uniform-mass and scalar general-mass sweeps, split sweeps,
    table-free (GRID_FAST / GRID_FAST_CLAMP) and table (GRID_EST) variants, the one-sided kernel, and the one-launch
    small-system kernel with 64 / 32 / 16 lanes per target.  (Bins depend on positions and softening only: both mass
    choices are checked against the same reference checksums.)"""
    env, which, expect = VARIANTS[variant]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g = load_golden(f"g13_bins_{case}.npz")
    ref = load_golden("g20_bin_checksums.npz")
    n = g["pos"].shape[0]
    uniform = masses == "uniform"
    mass = np.ones(n, np.float32) if uniform else (0.5 + np.random.default_rng(3).random(n)).astype(np.float32)
    sim = nb.GalaxySimulation(T(g["pos"]), T(g["vel"]), T(mass), precision_mode=nb.PrecisionMode(mode),
                              G=float(g["G"]), softening=float(g["eps"]), dt=float(g["dt"]))
    dbg = sim.quant_debug()
    assert np.float32(dbg["lmin"]) == np.float32(g[f"{mode}/lmin"]) and np.float32(dbg["lmax"]) == np.float32(g[f"{mode}/lmax"])
    acc_before = sim.accelerations.numpy().copy()
    bs = sim.quant_bin_sums(which)
    expect = dict(expect)
    if expect.get("path") == "sym":
        # the packed uniform-mass kernel serves equal masses on the R = 4 tiling (force_eval's rule)
        expect.setdefault("uniform_kernel", uniform and expect["shape"] == 4 and "NB_NO_UNIFORM" not in env)
    expect.setdefault("fast_path", True)
    _check(bs, ref[f"g13_{case}/{mode}/s1"], ref[f"g13_{case}/{mode}/s2"], expect, n * (n - 1) // 2)
    # the read-out recomputes the forces and must leave the state alone
    assert np.array_equal(sim.accelerations.numpy(), acc_before)
    if which == "small":
        # ... and after real steps the small-system kernel is what force_kernel_name() reports: read-out "as the last step"
        sim.run(2)
        assert sim.force_kernel_name() == "small_step_kernel"
        assert sim.quant_bin_sums("last")["path"] == "small"


@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("mode", GRID)
def test_config3_full_size_production_bins_vs_reference_rows(nb, monkeypatch, mode, general):
    """BASELINE config 3's real size, N = 65 536: the production plan (tiles of 256, pruned max-r2 search, packed
    uniform-mass kernel -- or the scalar general-mass one) against the reference's bins of six target rows, one of them
    a row of the farthest pair (golden g16 / g20).  The checksums of those rows collect the kernel's decisions from
    every source tile, from the target side AND the mirrored source side of the symmetric sweep."""
    if general:
        monkeypatch.setenv("NB_NO_UNIFORM", "1")
    g16 = load_golden("g16_bins_n65536_rows.npz")
    ref = load_golden("g20_bin_checksums.npz")
    pos = T(g16["pos"])
    n = pos.shape[0]
    sim = nb.GalaxySimulation(pos, torch.zeros_like(pos), torch.ones(n), precision_mode=nb.PrecisionMode(mode))
    assert sim.force_kernel_name().startswith("force_sym_kernel<float")
    bs = sim.quant_bin_sums("last")
    assert bs["path"] == "sym" and bs["shape"] == 4 and bs["fast_path"] and bs["uniform_kernel"] == (not general), bs
    rows = ref["g16/rows"]
    assert np.array_equal(bs["sum_k"][rows], ref[f"g16/{mode}/s1"]), (bs["sum_k"][rows], ref[f"g16/{mode}/s1"])
    assert np.array_equal(bs["sum_kw"][rows], ref[f"g16/{mode}/s2"])
    total = bs["pairs_table_free"] + bs["pairs_table"]
    assert total >= n * (n - 1) // 2 and bs["pairs_table_free"] > 0.8 * total, bs
    # every other row against the table walk of nb_quant_bins_rows on a sample (the two read-outs must agree everywhere)
    sample = np.array([7, 4099, 31111, 65535])
    for i in sample:
        k = sim.quant_bins_rows(int(i), int(i) + 1)
        s1, s2 = checksums(k)
        assert bs["sum_k"][i] == s1[0] and bs["sum_kw"][i] == s2[0]


@pytest.mark.parametrize("uniform", [True, False])
@pytest.mark.parametrize("levels", [1000, 4096, 3])
@pytest.mark.parametrize("n,d", [(3000, 2), (2500, 3)])
def test_custom_grids_beyond_the_small_tables_vs_oracle(nb, monkeypatch, n, d, levels, uniform):
    """CUSTOM grids of 1000 / 4096 levels (the 33 KB table variant of the kernels: estimate + threshold compare, or the
    binary search when the grid is too narrow for the estimate) and of 3 levels, against the oracle's bin matrix."""
    from oracle import oracle as O
    monkeypatch.setenv("NB_SYM", "1")
    rng = np.random.default_rng(n + levels)
    pos = (rng.standard_normal((n, d)) * 3).astype(np.float32)
    mass = np.ones(n, np.float32) if uniform else (0.5 + rng.random(n)).astype(np.float32)
    _, dbg = O.accelerations(pos, mass, "custom", levels=levels, debug=True)
    want1, want2 = checksums(dbg["d2bins"])
    sim = nb.GalaxySimulation(T(pos), torch.zeros(n, d), T(mass), precision_mode=nb.PrecisionMode.CUSTOM,
                              custom_levels=levels)
    bs = sim.quant_bin_sums("tiled")
    assert bs["path"] == "sym" and bs["levels"] == levels
    if levels > 256:
        assert not bs["fast_path"]          # the table-free path needs single-block tables (<= 256 levels)
    _check(bs, want1, want2, {}, n * (n - 1) // 2)


def test_degenerate_and_unsupported_cases_say_so(nb):
    """A degenerate grid (all pairs inside the floor: lmax - lmin < 1e-10) passes values through -- there are no bins
    and the read-out says so; FLOAT32 mode has no bins at all."""
    from nbody_cosmological_simulation_amd import _native
    n = 300
    pos = (np.random.default_rng(0).random((n, 2)) * 1e-3).astype(np.float32)        # every r2 < 0.01: clamped to the floor
    sim = nb.GalaxySimulation(T(pos), torch.zeros(n, 2), torch.ones(n), precision_mode=nb.PrecisionMode.INT8_SIM, softening=0.01)
    with pytest.raises(_native.NativeError, match="degenerate"):
        sim.quant_bin_sums("tiled")
    sim32 = nb.GalaxySimulation(T(pos), torch.zeros(n, 2), torch.ones(n), precision_mode=nb.PrecisionMode.FLOAT32)
    with pytest.raises(_native.NativeError, match="grid modes"):
        sim32.quant_bin_sums("tiled")


@pytest.mark.parametrize("world", [2, 3])
def test_bin_checksums_of_comm_less_shards_add_up(nb, monkeypatch, world):
    """Multi-GPU by construction: the work lists the ranks would run (snake-dealt super-rows) executed one after the other
    as comm-less shards -- every pair is binned by exactly one rank, so the ranks' checksums must ADD UP to the reference's
    (the same property the partial forces have, now for the bin decisions themselves)."""
    from nbody_cosmological_simulation_amd import _native as N
    monkeypatch.setenv("NB_SYM", "2")
    g = load_golden("g13_bins_n4096_d2.npz")
    ref = load_golden("g20_bin_checksums.npz")
    n = g["pos"].shape[0]
    s1 = np.zeros(n, np.int64)
    s2 = np.zeros(n, np.int64)
    for rank in range(world):
        sim = nb.GalaxySimulation(T(g["pos"]), T(g["vel"]), torch.ones(n), precision_mode=nb.PrecisionMode.INT8_SIM,
                                  G=float(g["G"]), softening=float(g["eps"]), dt=float(g["dt"]), shard=(rank, world))
        bs = sim.quant_bin_sums("tiled")
        assert bs["path"] == "sym"
        s1 += bs["sum_k"]
        s2 += bs["sum_kw"]
    assert np.array_equal(s1, ref["g13_n4096_d2/int8_sim/s1"]) and np.array_equal(s2, ref["g13_n4096_d2/int8_sim/s2"])


@pytest.mark.parametrize("uniform", [True, False])
@pytest.mark.parametrize("r", [2, 4])
@pytest.mark.parametrize("case", ["narrow", "degenerate"])
def test_packed_grid_kernels_on_narrow_and_degenerate_grids(nb, monkeypatch, case, r, uniform):
    """States of the tables the uniform-mass packed kernel used to leave to a general-mass twin launch (round 2) and now
    serves itself: a grid too NARROW for the bin estimate (softening 1.0, all stars within 0.05: r2 in [1, 1.01] ->
    binary search over the thresholds, GRID_SEARCH) and a DEGENERATE grid (every r2 below the floor 0.01: values pass
    through, quantization.py:115-116).  Forces against the oracle; bins (narrow case) against the oracle's bin matrix."""
    from oracle import oracle as O
    from nbody_cosmological_simulation_amd import _native
    monkeypatch.setenv("NB_SYM", "1")
    monkeypatch.setenv("NB_SYM_R", str(r))
    rng = np.random.default_rng(17 + r)
    n = 700
    if case == "narrow":
        pos, eps = (rng.random((n, 2)) * 0.05).astype(np.float32), 1.0
    else:
        pos, eps = (rng.random((n, 2)) * 2e-3).astype(np.float32), 0.05
    mass = np.full(n, 0.8, np.float32) if uniform else (0.5 + rng.random(n)).astype(np.float32)
    ref, dbg = O.accelerations(pos, mass, "custom", softening=eps, levels=200, debug=True)
    sim = nb.GalaxySimulation(T(pos), torch.zeros(n, 2), T(mass), precision_mode=nb.PrecisionMode.CUSTOM, custom_levels=200,
                              softening=eps)
    assert sim.force_kernel_name().startswith("force_sym_kernel<float")
    acc = sim.accelerations.numpy().astype(np.float64)
    assert np.abs(acc - ref).max() <= 2e-6 * np.abs(ref).max()
    if case == "narrow":
        got = sim.quant_debug()
        assert not got["fast_path"]
        bs = sim.quant_bin_sums("tiled")
        want1, want2 = checksums(dbg["d2bins"])
        assert np.array_equal(bs["sum_k"], want1) and np.array_equal(bs["sum_kw"], want2)
        assert bs["pairs_table_free"] == 0
    else:
        with pytest.raises(_native.NativeError, match="degenerate"):
            sim.quant_bin_sums("tiled")
