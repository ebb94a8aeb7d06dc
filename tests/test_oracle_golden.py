"""Pins the CPU oracle (oracle/) to golden vectors produced by RUNNING the reference
(tests/golden/make_golden.py).  CPU only.

Tolerances (relative to max|reference| of the compared array unless noted):
  * FLOAT64 mode: 1e-13 single evaluation / short runs (observed <= 1e-15), trajectory
    bounds follow the reference's own self-noise floor.
  * fp32-family modes: 1e-6 single evaluation (observed <= 3e-7: torch SLEEF pow/exp vs
    correctly rounded, and torch's fp32 cascade sum vs double accumulation).
  * distance-bin indices, lmin/lmax: bit-exact.  Force bins: mismatch count reported, and
    bounded (they depend on the fp32 summation order of the reference).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from oracle import oracle as O

MODES = ["float64", "float32", "bfloat16", "float16", "int8_sim", "int4_sim", "custom"]
GRID = ["int8_sim", "int4_sim", "custom"]
G1 = ["g1_n64_d2_e0.1.npz", "g1_n257_d2_e0.05.npz", "g1_n64_d3_e0.01.npz", "g1_n257_d3_e0.1.npz"]


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def tol_single(mode):
    return 1e-13 if mode == "float64" else 1e-6


@pytest.mark.parametrize("fname", G1)
@pytest.mark.parametrize("mode", MODES)
def test_g1_single_evaluation(fname, mode):
    g = load_golden(fname)
    eps = float(g["eps"])
    acc, dbg = O.accelerations(g["pos"], g["mass"], mode, softening=eps, debug=True)
    ref = g[f"{mode}/acc0"]
    assert acc.dtype == ref.dtype
    if mode in GRID:
        assert dbg["lmin"] == float(g[f"{mode}/lmin"])
        assert dbg["lmax"] == float(g[f"{mode}/lmax"])
        assert np.array_equal(dbg["d2bins"], g[f"{mode}/d2bins"])          # bit-identical bins
    if mode in ("int8_sim", "int4_sim"):
        assert relerr(dbg["acc_prequant"], g[f"{mode}/acc0_prequant"]) < 1e-6
        assert abs(dbg["fmin"] - float(g[f"{mode}/fmin"])) <= 1e-6 * abs(float(g[f"{mode}/fmin"]))
        assert abs(dbg["fmax"] - float(g[f"{mode}/fmax"])) <= 1e-6 * abs(float(g[f"{mode}/fmax"]))
        flips = int((dbg["fbins"] != g[f"{mode}/fbins"]).sum())
        assert flips <= 2, f"{flips} force-bin flips"
        if flips == 0:
            assert relerr(acc, ref) < 1e-6
    else:
        assert relerr(acc, ref) < tol_single(mode)


@pytest.mark.parametrize("fname", G1)
def test_g1_r2_bit_exact(fname):
    """fp32 distance-squared matrix (no FMA, reference op order) is bit-identical."""
    g = load_golden(fname)
    pos = g["pos"].astype(np.float32)
    diff = pos[None, :, :] - pos[:, None, :]
    sq = diff * diff
    s = sq[..., 0]
    for k in range(1, pos.shape[1]):
        s = s + sq[..., k]
    r2 = s + np.float32(float(g["eps"]) ** 2)
    assert np.array_equal(r2, g["r2_f32"])
    # and the oracle's own rounding model agrees with native float arithmetic
    out, bins, lmin, lmax = O.grid_quantize_safe(g["r2_f32"], 256, bins=True)
    assert np.array_equal(bins, g["int8_sim/d2bins"])


@pytest.mark.parametrize("fname", G1)
@pytest.mark.parametrize("mode", MODES)
def test_g1_step_and_energy(fname, mode):
    g = load_golden(fname)
    sim = O.OracleSim(g["pos"], g["vel"], g["mass"], mode, G=float(g["G"]), softening=float(g["eps"]),
                      dt=float(g["dt"]))
    tol = 1e-13 if mode == "float64" else 2e-6
    assert abs(sim.get_potential_energy() - float(g[f"{mode}/pe0"])) <= 2e-6 * abs(float(g[f"{mode}/pe0"]))
    assert abs(sim.get_kinetic_energy() - float(g[f"{mode}/ke0"])) <= 2e-6 * abs(float(g[f"{mode}/ke0"]))
    sim.step()
    assert sim.positions.dtype == g[f"{mode}/pos1"].dtype
    assert sim.velocities.dtype == g[f"{mode}/vel1"].dtype
    assert relerr(sim.positions, g[f"{mode}/pos1"]) < tol
    assert relerr(sim.velocities, g[f"{mode}/vel1"]) < tol
    if mode not in ("int8_sim", "int4_sim"):
        assert relerr(sim.accelerations, g[f"{mode}/acc1"]) < max(tol, 1e-6 if mode != "float64" else 0)
    sim.run(9)
    t10 = 1e-12 if mode == "float64" else (1e-5 if mode not in ("int8_sim", "int4_sim") else 2e-3)
    assert relerr(sim.positions, g[f"{mode}/pos10"]) < t10
    assert relerr(sim.velocities, g[f"{mode}/vel10"]) < t10 * 50


def test_g1_fp64_energy_exact():
    g = load_golden(G1[1])
    sim = O.OracleSim(g["pos"], g["vel"], g["mass"], "float64", softening=float(g["eps"]))
    # fp32 state at tick 0: energies are fp32-typed in the reference
    assert abs(sim.get_potential_energy() - float(g["float64/pe0"])) <= 1e-6 * abs(float(g["float64/pe0"]))
    sim.run(10)
    ke, pe = g["float64/e10"]
    assert abs(sim.get_kinetic_energy() - ke) <= 1e-13 * abs(ke)
    assert abs(sim.get_potential_energy() - pe) <= 1e-13 * abs(pe)


@pytest.mark.parametrize("L", [4, 64, 1000])
def test_g1c_custom_levels_override_semantics(L):
    """sensitivity_test.py:55-76 pattern == CUSTOM mode with `levels`, no force quantisation."""
    g = load_golden("g1c_custom_levels.npz")
    sim = O.OracleSim(g["pos"], g["vel"], g["mass"], "custom", levels=L)
    assert relerr(sim.accelerations, g[f"L{L}/acc0"]) < 1e-6
    sim.run(5)
    assert relerr(sim.positions, g[f"L{L}/pos5"]) < 1e-5


def test_g2_config1_fp64_trajectory():
    """Config 1 (N=1024, 200 ticks, float64): particle-level parity at every snapshot."""
    g = load_golden("g2_config1_n1024.npz")
    sim = O.OracleSim(g["pos"], g["vel"], g["mass"], "float64")
    scale = np.abs(g["float64/pos200"]).max()
    bounds = {0: 1e-15, 1: 1e-14, 10: 1e-13, 100: 1e-12, 200: 1e-11}   # reference self-noise: 1.2e-15 @200
    t = 0
    for snap in (0, 1, 10, 100, 200):
        sim.run(snap - t)
        t = snap
        assert np.abs(sim.positions.astype(np.float64) - g[f"float64/pos{snap}"]).max() / scale < bounds[snap]
        assert relerr(sim.velocities, g[f"float64/vel{snap}"]) < bounds[snap] * 100
        if snap in (0, 100, 200):
            for key, fn in (("ke", sim.get_kinetic_energy), ("pe", sim.get_potential_energy)):
                ref = float(g[f"float64/diag{snap}/{key}"])
                tol = 1e-6 if snap == 0 else 1e-12     # tick 0 energies are fp32-typed upstream
                assert abs(fn() - ref) <= tol * abs(ref)
    e0 = float(g["float64/diag100/e"])
    e1 = float(g["float64/diag200/e"])
    drift_ref = (e1 - e0) / abs(e0)
    # energy-drift parity (north_star: 1e-10)
    sim2 = O.OracleSim(g["pos"], g["vel"], g["mass"], "float64")
    sim2.run(100)
    a = sim2.get_total_energy()
    sim2.run(100)
    b = sim2.get_total_energy()
    assert abs((b - a) / abs(a) - drift_ref) < 1e-10


@pytest.mark.parametrize("mode", MODES[1:])
def test_g2_config1_other_modes_short_horizon(mode):
    g = load_golden("g2_config1_n1024.npz")
    sim = O.OracleSim(g["pos"], g["vel"], g["mass"], mode)
    sim.run(10)
    tol = 5e-6 if mode not in ("int8_sim", "int4_sim") else 5e-3
    assert relerr(sim.positions, g[f"{mode}/pos10"]) < tol


@pytest.mark.parametrize("n", [4096])
def test_g3_fast_path_matches_reference(n):
    g = load_golden(f"g3_fp64_n{n}.npz")
    pos, mass = g["pos"], g["mass"]
    # first evaluation in FLOAT64 mode uses fp32 d/r2: generic oracle
    acc0 = O.accelerations(pos[:512] if False else pos, mass, "float64")
    assert relerr(acc0, g["acc0"]) < 1e-13
    # all-fp64 fast path vs the generic restatement on the final (fp64) state
    pf = g["pos_final"]
    fast = O.accelerations_f64_fast(pf, mass.astype(np.float64))
    assert relerr(fast, g["acc_final"]) < 1e-13
    pe = O.potential_energy_f64_fast(pf, mass.astype(np.float64))
    assert abs(pe - g["e1"][1]) <= 1e-13 * abs(g["e1"][1])


def test_g3_n8192_final_forces():
    g = load_golden("g3_fp64_n8192.npz")
    fast = O.accelerations_f64_fast(g["pos_final"], g["mass"].astype(np.float64))
    assert relerr(fast, g["acc_final"]) < 1e-13


def test_g4_dtype_state_machine_and_mixed_inputs():
    api = json.load(open(os.path.join(GOLDEN, "api.json")))
    g = load_golden("g4_api.npz")
    names = {O.F16: "torch.float16", O.BF16: "torch.bfloat16", O.F32: "torch.float32", O.F64: "torch.float64"}
    for mode in MODES:
        sim = O.OracleSim(g["pos"], g["vel"], g["mass"], mode)
        rows = [[names[c] for c in (sim.codes[0], sim.codes[1], sim.codes[3])]]
        for _ in range(2):
            sim.step()
            rows.append([names[c] for c in (sim.codes[0], sim.codes[1], sim.codes[3])])
        assert rows == api["dtype_timeline_fp32_inputs"][mode], mode
        # fp64 inputs
        sim = O.OracleSim(g["pos"].astype(np.float64), g["vel"].astype(np.float64),
                          g["mass"].astype(np.float64), mode)
        tol = 1e-13 if mode == "float64" else 1e-6
        if mode not in ("int8_sim", "int4_sim"):
            assert relerr(sim.accelerations, g[f"in64/{mode}/acc0"]) < tol, mode
        r0 = [names[c] for c in (sim.codes[0], sim.codes[1], sim.codes[3])]
        sim.run(3)
        r1 = [names[c] for c in (sim.codes[0], sim.codes[1], sim.codes[3])]
        assert [r0, r1] == api["dtype_timeline_fp64_inputs"][mode], mode
        if mode not in ("int8_sim", "int4_sim"):
            assert relerr(sim.positions, g[f"in64/{mode}/pos3"]) < 1e-6
            ke, pe = g[f"in64/{mode}/e3"]
            assert abs(sim.get_potential_energy() - pe) <= 1e-6 * abs(pe)
            assert abs(sim.get_kinetic_energy() - ke) <= 1e-5 * abs(ke)


@pytest.mark.parametrize("name,code", [("float16", O.F16), ("bfloat16", O.BF16)])
def test_g4_half_precision_state(name, code):
    """omega_point_test.py:722-733: f16/bf16 state tensors in FLOAT32 mode."""
    api = json.load(open(os.path.join(GOLDEN, "api.json")))
    g = load_golden("g4_api.npz")
    import torch
    tdt = getattr(torch, name)
    rd = lambda a: torch.from_numpy(a).to(tdt).float().numpy()   # values representable in the half type
    sim = O.OracleSim(rd(g["pos"]), rd(g["vel"]), rd(g["mass"]), "float32", codes=(code, code, code))
    ref = g[f"in_{name}/acc0"]
    assert relerr(sim.accelerations, ref) < 2e-6
    assert abs(sim.get_potential_energy() - float(g[f"in_{name}/pe0"])) <= 4e-3 * abs(float(g[f"in_{name}/pe0"]))
    sim.run(3)
    names = {O.F16: "torch.float16", O.BF16: "torch.bfloat16", O.F32: "torch.float32", O.F64: "torch.float64"}
    assert [names[c] for c in (sim.codes[0], sim.codes[1], sim.codes[3])] == \
        api["dtype_timeline_half_inputs_float32_mode"][name][1]
    assert relerr(sim.positions, g[f"in_{name}/pos3"]) < 1e-5


def test_g5_tensor_hooks():
    g = load_golden("g5_hooks.npz")
    d2, force = g["d2"], g["force"]
    for mode in MODES:
        out = O.quantize_distance_squared(d2, mode)
        ref = g[f"qd2/{mode}"]
        assert out.dtype == ref.dtype
        if mode in GRID:
            assert relerr(out, ref) < 1e-6
        else:
            assert np.array_equal(out, ref)          # casts are exact (inf on fp16 overflow too)
        outf = O.quantize_force(force, mode)
        reff = g[f"qf/{mode}"]
        if mode in GRID:
            assert relerr(outf, reff) < 1e-6
        else:
            assert np.array_equal(outf, reff)
    for L in (4, 16, 100, 1000):
        assert relerr(O.quantize_distance_squared(d2, "custom", custom_levels=L), g[f"qd2/custom{L}"]) < 1e-6
        assert relerr(O.grid_quantize_safe(d2, L, min_val=0.5), g[f"safe/L{L}_min0.5"]) < 1e-6
        assert relerr(O.grid_quantize(force, L), g[f"lin/L{L}"]) < 1e-6
    const = np.full((7, 7), 3.0, np.float32)
    assert np.array_equal(O.grid_quantize_safe(const, 16), g["safe/const"])
    assert np.array_equal(O.grid_quantize(const, 16), g["lin/const"])
    assert relerr(O.quantize_distance_squared(d2.astype(np.float64), "int8_sim"), g["qd2_64/int8_sim"]) < 1e-13
    assert np.array_equal(O.quantize_distance_squared(d2.astype(np.float64), "float16"), g["qd2_64/float16"])


def test_j_range_partials_sum_to_full():
    g = load_golden(G1[1])
    pos, mass = g["pos"], g["mass"]
    full = O.accelerations(pos, mass, "float64", softening=0.05)
    n = pos.shape[0]
    parts = [O.accelerations(pos, mass, "float64", softening=0.05, j_range=(a, b), force_quant=False)
             for a, b in ((0, 100), (100, 200), (200, n))]
    assert relerr(sum(parts), full) < 1e-14
    f64 = O.accelerations_f64_fast(pos.astype(np.float64), mass.astype(np.float64), softening=0.05)
    p64 = sum(O.accelerations_f64_fast(pos.astype(np.float64), mass.astype(np.float64), softening=0.05, j_range=r)
              for r in ((0, 128), (128, n)))
    assert relerr(p64, f64) < 1e-14


@pytest.mark.parametrize("mode", ["float64", "float32", "bfloat16", "float16"])
def test_torch_materialised_restatement_agrees(mode):
    """oracle/torch_materialised.py (used by bench.py as 'the reference's PyTorch-CPU path') reproduces the
    golden trajectory bit for bit in FLOAT64 mode and agrees with the C oracle in the cast modes."""
    import torch
    from oracle import torch_materialised as TM
    g = load_golden("g2_config1_n1024.npz")
    pos, vel, m = (torch.from_numpy(g[k]) for k in ("pos", "vel", "mass"))
    st = dict(pos=pos, vel=vel, masses=m)
    st["acc"] = TM.accelerations(pos, m, mode=mode)
    for _ in range(10):
        TM.step(st, mode=mode)
    ref = g[f"{mode}/pos10"]
    assert st["pos"].numpy().dtype == ref.dtype
    if mode == "float64":
        assert np.array_equal(st["pos"].numpy(), ref)
    else:
        assert relerr(st["pos"].numpy(), ref) < 1e-6


@pytest.mark.parametrize("n", [96, 700])
@pytest.mark.parametrize("hname,code", [("float16", O.F16), ("bfloat16", O.BF16)])
@pytest.mark.parametrize("mode", ["float32", "bfloat16", "float16", "float64"])
def test_g7_half_typed_state(n, hname, code, mode):
    """Half-typed state through every cast mode and FLOAT64 (reference run, g7): energies before the promotion
    (half arithmetic, incl. the float16 overflow of the N = 700 potential sum), after it (masses still half:
    mass products rounded to half, simulation.py:185), and the state after three steps."""
    g = load_golden("g7_half_state.npz")
    key = f"n{n}/{hname}/{mode}"
    sim = O.OracleSim(g[f"n{n}/{hname}/pos"], g[f"n{n}/{hname}/vel"], g[f"n{n}/{hname}/mass"], mode,
                      codes=(code, code, code))

    def close(got, want, tol):
        return got == want or abs(got - want) <= tol * abs(want)

    ke0, pe0 = g[key + "/e0"]
    assert close(sim.get_kinetic_energy(), ke0, 4e-3) and close(sim.get_potential_energy(), pe0, 8e-3)
    sim.run(3)
    ke3, pe3 = g[key + "/e3"]
    tol = 1e-12 if mode == "float64" else 2e-6
    assert close(sim.get_kinetic_energy(), ke3, max(tol, 1e-9)) and close(sim.get_potential_energy(), pe3, max(tol, 1e-9))
    assert relerr(sim.positions, g[key + "/pos3"]) < (1e-12 if mode == "float64" else 1e-5)
    assert relerr(sim.velocities, g[key + "/vel3"]) < (1e-10 if mode == "float64" else 1e-4)
    names = {O.F16: "torch.float16", O.BF16: "torch.bfloat16", O.F32: "torch.float32", O.F64: "torch.float64"}
    assert [names[c] for c in (sim.codes[0], sim.codes[1], sim.codes[3], sim.codes[2])] == list(g[key + "/dtypes3"])


def _g8_check(get_sim, key, g, f64):
    """Shared by the oracle (here) and the engine (tests/test_gpu_parity.py): one g8 case against the reference."""
    mode, eps, dt = key.split("/")
    eps, dt = float(eps[3:]), float(dt[2:])
    sim = get_sim(mode, eps, dt)
    wild = dt >= 1.0 and eps <= 1e-3          # 2-unit steps through 1e-4-softened encounters: rounding is amplified

    def same(got, want, tol):
        got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
        if not np.array_equal(np.isnan(got), np.isnan(want)):
            return False
        ok = ~np.isnan(want)
        return ok.sum() == 0 or np.abs(got[ok] - want[ok]).max() <= tol * max(np.abs(want[ok]).max(), 1e-300)

    acc = sim.accelerations
    acc = acc.double().numpy() if hasattr(acc, "double") else acc
    assert same(acc, g[key + "/acc0"], 1e-13 if f64 else 2e-6), key
    # tick 0: the state is still fp32-typed in every mode (fp32 inputs), so the energies are fp32 sums
    assert same([sim.get_kinetic_energy(), sim.get_potential_energy()], g[key + "/e0"], 2e-6), key
    sim.run(5)
    pos, vel = sim.positions, sim.velocities
    pos = pos.double().numpy() if hasattr(pos, "double") else pos
    vel = vel.double().numpy() if hasattr(vel, "double") else vel
    tol = 5e-2 if (wild and not f64) else (1e-9 if f64 else 1e-5)
    assert same(pos, g[key + "/pos5"], tol), key
    assert same(vel, g[key + "/vel5"], tol * (1 if f64 or wild else 10)), key
    # (the bf16 / softening 1e-3 run gains 45x its kinetic energy in five steps: a half-precision rounding flip of
    # one close pair's r2 moves it by 1e-5)
    e_tol = 5e-2 if wild and not f64 else (1e-9 if f64 else 1e-4)
    assert same([sim.get_kinetic_energy()], [g[key + "/e5"][0]], e_tol), key
    assert same([sim.get_potential_energy()], [g[key + "/e5"][1]], e_tol), key


@pytest.mark.parametrize("idx", range(12))
def test_g8_parameter_extremes(idx):
    """crash_point_test.py / falsification_tests.py parameter ranges through the stock class: softening 0 (NaN
    forces and NaN potential, like the reference's masked 0/0), 1e-4 (NaN in FLOAT16 mode, where it underflows),
    1.0; dt up to 2.0; grid modes below their 0.01 clamp."""
    g = load_golden("g8_extremes.npz")
    key = str(g["cases"][idx])
    _g8_check(lambda mode, eps, dt: O.OracleSim(g["pos"], g["vel"], g["mass"], mode, G=0.001, softening=eps, dt=dt),
              key, g, key.startswith("float64"))


def _g9_check(get_sim, key, g):
    sname, mode = key.split("/")
    sim = get_sim(g[sname + "/pos"], g[sname + "/vel"], g[sname + "/mass"], mode)
    f64 = mode == "float64"

    def same(got, want, tol):
        got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
        if got.shape != want.shape or not np.array_equal(np.isnan(got), np.isnan(want)):
            return False
        ok = ~np.isnan(want)
        return ok.sum() == 0 or np.abs(got[ok] - want[ok]).max() <= tol * max(np.abs(want[ok]).max(), 1e-300)

    tonp = lambda t: t.double().numpy() if hasattr(t, "double") else t
    assert same(tonp(sim.accelerations), g[key + "/acc0"], 1e-13 if f64 else 2e-6), key
    assert same([sim.get_kinetic_energy(), sim.get_potential_energy()], g[key + "/e0"], 2e-6), key
    sim.run(2)
    assert same(tonp(sim.positions), g[key + "/pos2"], 1e-13 if f64 else 1e-6), key
    assert same(tonp(sim.velocities), g[key + "/vel2"], 1e-12 if f64 else 1e-6), key
    assert same([sim.get_kinetic_energy(), sim.get_potential_energy()], g[key + "/e2"], 1e-12 if f64 else 2e-6), key


@pytest.mark.parametrize("idx", range(35))
def test_g9_degenerate_systems(idx):
    """N = 1, 2, 3, five coincident particles, a massless particle in 3-D, every mode (reference run, g9)."""
    g = load_golden("g9_degenerate.npz")
    _g9_check(lambda p, v, m, mode: O.OracleSim(p, v, m, mode), str(g["cases"][idx]), g)


def _g10_check(fns, g, name):
    """fns: (quantize_distance_squared, quantize_force, grid_quantize_safe, grid_quantize) taking numpy arrays."""
    qd2, qf, safe, lin = fns
    t = g["in/" + name]

    def same(got, want, tol=1e-6):
        got, want = np.asarray(got), np.asarray(want)
        assert got.shape == want.shape and got.dtype == want.dtype, (name, got.shape, want.shape, got.dtype, want.dtype)
        a, b = got.astype(np.float64), want.astype(np.float64)
        assert np.array_equal(np.isnan(a), np.isnan(b)), name
        inf = np.isinf(b)
        assert np.array_equal(a[inf], b[inf]), name
        ok = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), ok), name
        if ok.any():
            assert np.abs(a[ok] - b[ok]).max() <= tol * max(np.abs(b[ok]).max(), 1e-300), name

    for mode in MODES:
        same(qd2(t, mode), g[f"{name}/qd2/{mode}"])
        same(qf(t - np.float32(15.0), mode), g[f"{name}/qf/{mode}"])
    for L in (2, 3, 7):
        same(safe(t, L, 0.01), g[f"{name}/safe/L{L}"])
        same(lin(t - np.float32(15.0), L), g[f"{name}/lin/L{L}"])


@pytest.mark.parametrize("name", ["zero_neg_huge", "inf", "nan", "vec1d", "cube3d", "single"])
def test_g10_hook_edge_values(name):
    """Tensor-level hooks on zeros, negatives, 1e30, inf, NaN, 2-3 levels, 1-D / 3-D / one-element tensors (g10)."""
    g = load_golden("g10_hook_edges.npz")
    _g10_check((O.quantize_distance_squared, O.quantize_force, O.grid_quantize_safe, O.grid_quantize), g, name)


# --------------------------------------------------------------------------- G13: bins at scale, from the reference
G13 = ["g13_bins_n4096_d2.npz", "g13_bins_n2048_d3.npz"]


def row_crcs(bins):
    import zlib
    b = np.ascontiguousarray(np.asarray(bins).astype("<i2"))
    return np.array([zlib.crc32(b[i].tobytes()) for i in range(b.shape[0])], np.uint32)


@pytest.mark.parametrize("fname", G13)
@pytest.mark.parametrize("mode", GRID)
def test_g13_bins_at_scale_vs_reference(fname, mode):
    """Quant-bin assignments of the REFERENCE at N = 4096 (D = 2) / N = 2048 (D = 3, softening below the grid
    floor): every row of the N x N int16 bin matrix by CRC-32, the bin histogram, lmin / lmax, the force grid."""
    g = load_golden(fname)
    acc, dbg = O.accelerations(g["pos"], g["mass"], mode, softening=float(g["eps"]), debug=True)
    assert np.float32(dbg["lmin"]) == np.float32(g[f"{mode}/lmin"])
    assert np.float32(dbg["lmax"]) == np.float32(g[f"{mode}/lmax"])
    levels = {"int8_sim": 256, "int4_sim": 16, "custom": 64}[mode]
    assert np.array_equal(np.bincount(dbg["d2bins"].ravel(), minlength=levels), g[f"{mode}/hist"])
    assert np.array_equal(row_crcs(dbg["d2bins"]), g[f"{mode}/row_crc"]), "a row of bins differs from the reference"
    n = g["pos"].shape[0]
    for r in (0, n // 2, n - 1):
        assert np.array_equal(dbg["d2bins"][r].astype(np.int16), g[f"{mode}/row{r}"])
    if mode != "custom":
        flips = int((dbg["fbins"] != g[f"{mode}/fbins"]).sum())
        print(f"{fname} {mode}: force-bin flips {flips} of {dbg['fbins'].size}")
        assert flips <= 4
        assert abs(dbg["fmin"] - float(g[f"{mode}/fmin"])) <= 2e-6 * abs(float(g[f"{mode}/fmin"]))
        assert abs(dbg["fmax"] - float(g[f"{mode}/fmax"])) <= 2e-6 * abs(float(g[f"{mode}/fmax"]))
    else:
        assert relerr(acc, g[f"{mode}/acc0"]) < 2e-6


@pytest.mark.parametrize("fname", G13)
@pytest.mark.parametrize("mode", ["int8_sim", "int4_sim", "custom"])
def test_g20_bin_checksums_vs_reference(fname, mode):
    """The per-row integer checksums the HIP path's production pair loops report (nb_quant_bin_sums: sum_j k and
    sum_j k * ((j mod 65521) + 1)) computed from the ORACLE's bin matrix equal the ones make_golden.py g20 computed
    from the reference's bin matrix -- so the fixture and the checksum definition are pinned on the CPU too."""
    g = load_golden(fname)
    ref = load_golden("g20_bin_checksums.npz")
    case = fname[len("g13_bins_"):-len(".npz")]
    _, dbg = O.accelerations(g["pos"], g["mass"], mode, softening=float(g["eps"]), debug=True)
    k = dbg["d2bins"].astype(np.int64)
    assert np.all(np.diag(k) == 0)
    w = (np.arange(k.shape[1], dtype=np.int64) % 65521) + 1
    assert np.array_equal(k.sum(axis=1), ref[f"g13_{case}/{mode}/s1"])
    assert np.array_equal((k * w[None, :]).sum(axis=1), ref[f"g13_{case}/{mode}/s2"])


def test_torch_scalar_semantics_on_half_tensors():
    """What the oracle's scalar_as / scalar_mul (nbody_oracle.c) encode, measured on torch itself (no reference code
    involved): a Python scalar ADDED to a half tensor is rounded to the tensor's dtype first; a Python scalar that
    MULTIPLIES / DIVIDES one stays in float; G / t is t.reciprocal() times float(G)."""
    import torch
    torch.manual_seed(0)
    for dt in (torch.float16, torch.bfloat16):
        x = (torch.rand(50000) * 10 + 0.1).to(dt)
        f = x.float()
        assert torch.equal(x * 0.001, (f * np.float32(0.001)).to(dt))
        assert not torch.equal(x * 0.001, (f * torch.tensor(0.001).to(dt).float()).to(dt))
        assert torch.equal(x / 3.3, (f / np.float32(3.3)).to(dt))
        assert torch.equal(0.001 / x, (x.reciprocal().float() * np.float32(0.001)).to(dt))
        y = (torch.rand(50000) * 0.05).to(dt)
        assert torch.equal(y + 0.0123, (y.float() + torch.tensor(0.0123).to(dt).float()).to(dt))
        assert not torch.equal(y + 0.0123, (y.float() + np.float32(0.0123)).to(dt))


_G15_HALF = [(grp, mode) for grp in ("half", "bf16") for mode in ("int8_sim", "int4_sim", "custom")]


@pytest.mark.parametrize("grp,mode", _G15_HALF)
def test_g15_grid_modes_on_half_typed_state_vs_reference(grp, mode):
    """The grid modes on float16 / bfloat16 state tensors (g15: the stock class accepts them): every op of
    quantization.py:106-127 rounds to the half type.  With the scalar semantics above the oracle reproduces the
    reference's forces to fp32 summation level -- before round 3 both the oracle and the HIP path rounded G and
    (levels - 1) to the half type first and sat 4e-4 away, which the 4e-3 bar of round 2 hid."""
    g = load_golden("g15_dtype_combos.npz")
    tag = f"{grp}/{mode}"
    assert int(g[f"{tag}/ok"]) == 1
    code = O.F16 if grp == "half" else O.BF16
    import torch
    cast = (lambda a: torch.from_numpy(a).half().float().numpy()) if grp == "half" else \
           (lambda a: torch.from_numpy(a).bfloat16().float().numpy())
    pos, mass = cast(g["pos"]), cast(g["mass"])
    acc, dbg = O.accelerations(pos, mass, mode, softening=0.1, debug=True, pos_code=code, mass_code=code)
    ref = g[f"{tag}/acc0"]
    if mode == "custom":
        assert relerr(acc, ref) < 2e-6, relerr(acc, ref)
    else:
        levels = 256 if mode == "int8_sim" else 16
        step = (float(g[f"{tag}/fmax"]) - float(g[f"{tag}/fmin"])) / (levels - 1)
        diff = np.abs(np.asarray(acc, np.float64) - ref)
        frac = float((diff > 0.5 * step).mean())
        print(f"{tag}: force values off by more than half a grid step: {frac:.4f}, max {diff.max() / step:.3f} steps")
        assert abs(dbg["fmin"] - float(g[f"{tag}/fmin"])) <= 2e-6 * abs(float(g[f"{tag}/fmin"]))
        assert abs(dbg["fmax"] - float(g[f"{tag}/fmax"])) <= 2e-6 * abs(float(g[f"{tag}/fmax"]))
        noise = load_golden("g21_reference_self_noise.npz")
        ref_frac = max(float(noise[f"half/{grp}/{mode}/{si}/frac_gt_half_step"]) for si in range(8))
        assert frac <= max(2 * ref_frac, 2.0 / diff.size), (frac, ref_frac)
        assert diff.max() <= 1.01 * step


def test_g14_state_hash_matches_reference():
    """checkpoint.state_hash against hashes produced by the reference's reproducibility.hash_tensor_state."""
    import json
    import torch
    from nbody_cosmological_simulation_amd import checkpoint
    want = json.load(open(os.path.join(GOLDEN, "g14_state_hash.json")))
    g = load_golden("g2_config1_n1024.npz")
    pos, vel = torch.from_numpy(g["pos"]), torch.from_numpy(g["vel"])
    assert checkpoint.state_hash(pos, vel) == want["float32"]
    assert checkpoint.state_hash(pos.double(), vel.double()) == want["float64"]
    assert checkpoint.state_hash(pos.half(), vel.half()) == want["float16"]
    assert checkpoint.state_hash(torch.from_numpy(g["float64/pos200"]), torch.from_numpy(g["float64/vel200"])) == \
        want["float64_tick200"]
    # bfloat16 (numpy has no such dtype; upstream's .numpy() raises there): hashed over the raw 16-bit patterns
    hb = checkpoint.state_hash(pos.bfloat16(), vel.bfloat16())
    assert len(hb) == 16 and hb != want["float16"]


def test_metrics_oracle_vs_reference():
    """oracle/metrics_oracle.py (the checker of the native nb_metrics kernels) on the reference's own galaxies (g6):
    rotation curve (20 bins, and 7 bins with a fixed max radius), r90 / r50, bound fraction, velocity dispersion."""
    from oracle import metrics_oracle as MO
    g = load_golden("g6_galaxy_metrics.npz")
    for name in ("disk", "test", "halo"):
        p, v, m = (g[f"{name}/{k}"] for k in ("pos", "vel", "mass"))
        rc = MO.rotation_curve(p, v)
        assert np.allclose(rc["radii"], g[f"{name}/rc_r"], rtol=1e-6)
        assert list(rc["num_stars_per_bin"]) == list(g[f"{name}/rc_n"])
        assert np.allclose(rc["velocities"], g[f"{name}/rc_v"], rtol=2e-6, equal_nan=True)
        rc7 = MO.rotation_curve(p, v, num_bins=7, max_radius=12.5)
        assert np.allclose(rc7["velocities"], g[f"{name}/rc7_v"], rtol=2e-6, equal_nan=True)
        assert abs(MO.galaxy_radius(p, 90) - float(g[f"{name}/r90"])) <= 1e-6 * float(g[f"{name}/r90"])
        assert abs(MO.galaxy_radius(p, 50) - float(g[f"{name}/r50"])) <= 1e-6 * float(g[f"{name}/r50"])
        assert abs(MO.bound_fraction(p, v, m, 0.001) - float(g[f"{name}/bound"])) <= 1e-6
        assert abs(MO.velocity_dispersion(v) - float(g[f"{name}/disp"])) <= 2e-6 * float(g[f"{name}/disp"])


def test_metrics_oracle_non_finite_stars():
    """One star with a NaN position belongs to no bin and one with an infinite speed only spoils its own bin
    (the reference's per-bin masked means, metrics.py:62-68) -- what blown-up INT4 runs produce."""
    from oracle import metrics_oracle as MO
    rng = np.random.default_rng(3)
    p = (rng.standard_normal((500, 2)) * 3).astype(np.float32)
    v = (rng.standard_normal((500, 2)) * 0.2).astype(np.float32)
    v[7, 1] = np.inf
    rc = MO.rotation_curve(p, v, num_bins=5, max_radius=10.0)
    assert np.isinf(rc["velocities"]).sum() == 1 and np.isfinite(rc["velocities"]).sum() == 4
    p[11, 0] = np.nan
    rc2 = MO.rotation_curve(p, v, num_bins=5, max_radius=10.0)
    assert sum(rc2["num_stars_per_bin"]) == sum(rc["num_stars_per_bin"]) - 1
    assert np.isfinite(rc2["velocities"]).sum() == 4

def test_oracle_bins_at_config3_size_vs_reference_rows():
    """g16: distance bins of six target rows at N = 65 536 (BASELINE config 3's real size) as the REFERENCE's own
    quantize_distance_squared assigns them (row block containing the farthest pair: global lmin / lmax; generator:
    tests/golden/make_golden.py g16).  The oracle's rows must match bit for bit -- INT8 here (the all-pairs scan for
    lmax is the cost: once), all three grid modes on the GPU."""
    import hashlib
    import zlib
    g = load_golden("g16_bins_n65536_rows.npz")
    import torch
    pos = torch.from_numpy(g["pos"])              # the golden carries its positions (host-dependent last bits otherwise)
    assert hashlib.sha256(pos.numpy().tobytes()).hexdigest() == str(g["pos_sha256"])
    mass = torch.ones(pos.shape[0])
    rows = [int(r) for r in g["rows"]]
    for mode in ("int8_sim",):
        for idx, r in enumerate(rows[:3] + rows[3:4]):
            ridx = rows.index(r)
            _, dbg = O.accelerations_rows(pos.numpy(), mass.numpy(), mode, r, r + 1, bins=True)
            assert np.float32(dbg["lmin"]) == np.float32(g[f"{mode}/lmin"]) and np.float32(dbg["lmax"]) == np.float32(g[f"{mode}/lmax"])
            k16 = np.ascontiguousarray(dbg["d2bins"][0].astype("<i2"))
            assert zlib.crc32(k16.tobytes()) == int(g[f"{mode}/row_crc"][ridx]), (mode, r)
            if idx == 0:
                break          # one all-pairs scan on the CPU is enough here; the GPU test checks every row and mode
    # accelerations of the sampled rows as the reference's torch expressions give them (simulation.py:83-112 on the row
    # block); the cast modes need no global scan
    p_np, m_np = pos.numpy(), mass.numpy()
    for mode, tol in (("float64", 1e-13), ("float32", 2e-6), ("bfloat16", 2e-6), ("float16", 2e-6)):
        ref = g[f"{mode}/acc_rows"]
        scale = np.abs(ref).max()
        for idx, r in enumerate(rows):
            got, _ = O.accelerations_rows(p_np, m_np, mode, r, r + 1)
            assert np.abs(got[0].astype(np.float64) - ref[idx]).max() <= tol * scale, (mode, r)


def test_oracle_first_evaluation_and_drift_at_config2_size_vs_reference_ops():
    """g17 (N = 65 536, FLOAT64 mode, fp32 initial conditions, zero velocities): the first force evaluation of 2048
    sampled rows and the positions after the opening kick + drift, against the reference's own torch expressions
    evaluated on row blocks (make_golden.py g17).  The full step and the energies need every row's force: GPU test."""
    import torch
    g = load_golden("g17_step_n65536.npz")
    pos = load_golden("g16_bins_n65536_rows.npz")["pos"]
    mass = np.ones(pos.shape[0], np.float32)
    rows = g["rows"]
    scale = np.abs(g["acc0"]).max()
    for i0 in range(0, len(rows), 256):                      # the oracle evaluates contiguous row ranges
        for r in rows[i0:i0 + 8]:                            # a few rows per block keep this a seconds-long test
            got, _ = O.accelerations_rows(pos, mass, "float64", int(r), int(r) + 1)
            idx = int(np.where(rows == r)[0][0])
            assert got.dtype == np.float64
            assert np.abs(got[0] - g["acc0"][idx]).max() <= 1e-13 * scale
            v_half = 0.0 + got[0] * (0.01 / 2)
            x1 = pos[r].astype(np.float64) + v_half * 0.01
            assert np.abs(x1 - g["pos1"][idx]).max() <= 1e-15 * np.abs(g["pos1"]).max()


def test_oracle_prequant_forces_at_config3_size_vs_reference_ops():
    """g18: the forces BEFORE quantize_force of two sampled rows at N = 65 536 in INT8 mode (reference's torch
    expressions on row blocks with the global grid bounds) against the oracle; the full force grid is a GPU test."""
    g = load_golden("g18_force_quant_n65536.npz")
    pos = load_golden("g16_bins_n65536_rows.npz")["pos"]
    mass = np.ones(pos.shape[0], np.float32)
    scale = np.abs(g["int8_sim/pre_rows"]).max()
    for idx in (0, 2047):
        r = int(g["rows"][idx])
        got, _ = O.accelerations_rows(pos, mass, "int8_sim", r, r + 1)
        assert np.abs(got[0].astype(np.float64) - g["int8_sim/pre_rows"][idx]).max() <= 2e-6 * scale
