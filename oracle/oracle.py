"""ctypes front-end of the CPU oracle (oracle/nbody_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of nbody_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.

`OracleSim` restates the reference's GalaxySimulation (simulation.py:12-196) on numpy
arrays with the reference's dtype state machine (SURVEY.md section 8a "Facts").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnbody_oracle.so")

F16, BF16, F32, F64 = 0, 1, 2, 3
MODE_CODES = {"float64": 0, "float32": 1, "bfloat16": 2, "float16": 3,
              "int8_sim": 4, "int4_sim": 5, "custom": 6}
_NP = {F16: np.float16, F32: np.float32, F64: np.float64, BF16: np.float32}
_CODE = {np.dtype(np.float16): F16, np.dtype(np.float32): F32, np.dtype(np.float64): F64}


def build(force=False):
    src = os.path.join(_HERE, "nbody_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.nbo_accelerations.restype = C.c_int
        L.nbo_accelerations.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp, C.c_int, C.c_int,
                                        C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                        dp, dp, ip, ip, dp]
        L.nbo_accelerations_rows.restype = C.c_int
        L.nbo_accelerations_rows.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp, C.c_int, C.c_int,
                                             C.c_double, C.c_double, C.c_int, C.c_int, dp, dp, ip]
        L.nbo_acc_dtype.restype = C.c_int
        L.nbo_acc_dtype.argtypes = [C.c_int, C.c_int, C.c_int]
        L.nbo_axpy.restype = C.c_int
        L.nbo_axpy.argtypes = [C.c_long, C.c_int, dp, C.c_int, dp, C.c_double, dp]
        L.nbo_kinetic_energy.restype = C.c_double
        L.nbo_kinetic_energy.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp]
        L.nbo_potential_energy.restype = C.c_double
        L.nbo_potential_energy.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp, C.c_double,
                                           C.c_double, C.c_int, C.c_int]
        L.nbo_grid_quantize_safe.restype = C.c_int
        L.nbo_grid_quantize_safe.argtypes = [C.c_long, C.c_int, dp, dp, C.c_int, C.c_double, dp, dp, ip]
        L.nbo_grid_quantize.restype = C.c_int
        L.nbo_grid_quantize.argtypes = [C.c_long, C.c_int, dp, dp, C.c_int, dp, dp, ip]
        L.nbo_quantize_distance_squared.restype = C.c_int
        L.nbo_quantize_distance_squared.argtypes = [C.c_long, C.c_int, dp, dp, C.c_int, C.c_int,
                                                    C.c_double, C.POINTER(C.c_int)]
        L.nbo_quantize_force.restype = C.c_int
        L.nbo_quantize_force.argtypes = [C.c_long, C.c_int, dp, dp, C.c_int, C.c_int,
                                         C.POINTER(C.c_int), dp, dp, ip]
        L.nbo_accelerations_f64_fast.restype = None
        L.nbo_accelerations_f64_fast.argtypes = [C.c_int, C.c_int, dp, dp, C.c_double, C.c_double,
                                                 C.c_int, C.c_int, dp]
        fp = C.POINTER(C.c_float)
        L.nbo_accelerations_f32_fast.restype = None
        L.nbo_accelerations_f32_fast.argtypes = [C.c_int, C.c_int, fp, fp, C.c_float, C.c_float,
                                                 C.c_int, C.c_int, fp]
        L.nbo_accelerations_f32_fast_sub.restype = None
        L.nbo_accelerations_f32_fast_sub.argtypes = [C.c_int, C.c_int, fp, fp, C.c_float, C.c_float,
                                                     C.c_int, C.c_int, fp]
        L.nbo_step_f64_fast.restype = None
        L.nbo_step_f64_fast.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, C.c_double, C.c_double,
                                        C.c_double, C.c_int]
        L.nbo_potential_energy_f64_fast.restype = C.c_double
        L.nbo_potential_energy_f64_fast.argtypes = [C.c_int, C.c_int, dp, dp, C.c_double, C.c_double]
        L.nbo_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def dtype_code(arr, bf16=False):
    return BF16 if bf16 else _CODE[np.dtype(arr.dtype)]


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def from_f64(a, code):
    return a.astype(_NP[code])


def mode_code(mode):
    if isinstance(mode, int):
        return mode
    return MODE_CODES[getattr(mode, "value", mode)]


# ---------------------------------------------------------------- functional API

def accelerations(pos, mass, mode, G=0.001, softening=0.1, levels=0, j_range=None,
                  force_quant=True, debug=False, pos_code=None, mass_code=None):
    """simulation.py:74-118.  Returns acc (numpy, reference dtype) [, debug dict]."""
    pc = dtype_code(pos) if pos_code is None else pos_code
    mc = dtype_code(mass) if mass_code is None else mass_code
    n, d = pos.shape
    p64, m64 = as_f64(pos), as_f64(mass)
    out = np.empty((n, d), np.float64)
    j0, j1 = (0, n) if j_range is None else j_range
    dbg = np.zeros(4, np.float64)
    d2b = np.full((n, n), -9, np.int32) if debug else None
    fb = np.full((n, d), -9, np.int32) if debug else None
    pre = np.empty((n, d), np.float64) if debug else None
    A = lib().nbo_accelerations(n, d, pc, _dp(p64), mc, _dp(m64), mode_code(mode), levels,
                                float(G), float(softening) ** 2, j0, j1, int(force_quant),
                                _dp(out), _dp(dbg), _ip(d2b), _ip(fb),
                                None if pre is None else _dp(pre))
    acc = from_f64(out, A)
    if debug:
        return acc, dict(lmin=dbg[0], lmax=dbg[1], fmin=dbg[2], fmax=dbg[3], d2bins=d2b, fbins=fb,
                         acc_prequant=from_f64(pre, A), acc_code=A)
    return acc


def accelerations_rows(pos, mass, mode, i0, i1, G=0.001, softening=0.1, levels=0, bins=False):
    """Rows [i0, i1) of `accelerations` (no force quantisation) at sizes where all N^2 pairs are too slow; the
    grid's global lmin / lmax still come from all pairs.  Returns acc rows, dict(lmin, lmax[, d2bins])."""
    pc, mc = dtype_code(pos), dtype_code(mass)
    n, d = pos.shape
    p64, m64 = as_f64(pos), as_f64(mass)
    out = np.empty((i1 - i0, d), np.float64)
    dbg = np.zeros(4, np.float64)
    d2b = np.full((i1 - i0, n), -9, np.int32) if bins else None
    A = lib().nbo_accelerations_rows(n, d, pc, _dp(p64), mc, _dp(m64), mode_code(mode), levels, float(G),
                                     float(softening) ** 2, int(i0), int(i1), _dp(out), _dp(dbg), _ip(d2b))
    return from_f64(out, A), dict(lmin=dbg[0], lmax=dbg[1], d2bins=d2b)


def grid_quantize_safe(t, levels, min_val=0.01, bins=False):
    code = dtype_code(t)
    a = as_f64(t).ravel()
    out = np.empty_like(a)
    b = np.empty(a.size, np.int32) if bins else None
    lmin, lmax = C.c_double(), C.c_double()
    lib().nbo_grid_quantize_safe(a.size, code, _dp(a), _dp(out), levels, min_val,
                                 C.byref(lmin), C.byref(lmax), _ip(b))
    res = from_f64(out, code).reshape(t.shape)
    return (res, b.reshape(t.shape), lmin.value, lmax.value) if bins else res


def grid_quantize(t, levels, bins=False):
    code = dtype_code(t)
    a = as_f64(t).ravel()
    out = np.empty_like(a)
    b = np.empty(a.size, np.int32) if bins else None
    mn, mx = C.c_double(), C.c_double()
    lib().nbo_grid_quantize(a.size, code, _dp(a), _dp(out), levels, C.byref(mn), C.byref(mx), _ip(b))
    res = from_f64(out, code).reshape(t.shape)
    return (res, b.reshape(t.shape), mn.value, mx.value) if bins else res


def quantize_distance_squared(t, mode, custom_levels=None, min_dist_sq=0.01):
    code = dtype_code(t)
    a = as_f64(t).ravel()
    out = np.empty_like(a)
    tout = C.c_int()
    lib().nbo_quantize_distance_squared(a.size, code, _dp(a), _dp(out), mode_code(mode),
                                        custom_levels or 0, min_dist_sq, C.byref(tout))
    return from_f64(out, tout.value).reshape(t.shape)


def quantize_force(t, mode, custom_levels=None):
    code = dtype_code(t)
    a = as_f64(t).ravel()
    out = np.empty_like(a)
    tout = C.c_int()
    mn, mx = C.c_double(), C.c_double()
    lib().nbo_quantize_force(a.size, code, _dp(a), _dp(out), mode_code(mode), custom_levels or 0,
                             C.byref(tout), C.byref(mn), C.byref(mx), None)
    return from_f64(out, tout.value).reshape(t.shape)


def accelerations_f64_fast(pos, mass, G=0.001, softening=0.1, j_range=None):
    n, d = pos.shape
    p, m = as_f64(pos), as_f64(mass)
    out = np.empty((n, d), np.float64)
    j0, j1 = (0, n) if j_range is None else j_range
    lib().nbo_accelerations_f64_fast(n, d, _dp(p), _dp(m), float(G), float(softening) ** 2, j0, j1, _dp(out))
    return out


def accelerations_f32_fast(pos, mass, G=0.001, softening=0.1, j_range=None):
    n, d = pos.shape
    p = np.ascontiguousarray(pos, np.float32)
    m = np.ascontiguousarray(mass, np.float32)
    out = np.empty((n, d), np.float32)
    j0, j1 = (0, n) if j_range is None else j_range
    fp = C.POINTER(C.c_float)
    lib().nbo_accelerations_f32_fast(n, d, p.ctypes.data_as(fp), m.ctypes.data_as(fp),
                                     np.float32(G), np.float32(float(softening) ** 2), j0, j1,
                                     out.ctypes.data_as(fp))
    return out


def accelerations_f32_fast_subset(pos, mass, i0, i1, G=0.001, softening=0.1):
    """FLOAT32-mode forces on targets [i0, i1) only (all sources)."""
    n, d = pos.shape
    p = np.ascontiguousarray(pos, np.float32)
    m = np.ascontiguousarray(mass, np.float32)
    out = np.empty((i1 - i0, d), np.float32)
    fp = C.POINTER(C.c_float)
    lib().nbo_accelerations_f32_fast_sub(n, d, p.ctypes.data_as(fp), m.ctypes.data_as(fp), np.float32(G),
                                         np.float32(float(softening) ** 2), i0, i1, out.ctypes.data_as(fp))
    return out


def potential_energy_f64_fast(pos, mass, G=0.001, softening=0.1):
    n, d = pos.shape
    p, m = as_f64(pos), as_f64(mass)
    return lib().nbo_potential_energy_f64_fast(n, d, _dp(p), _dp(m), float(G), float(softening) ** 2)


def num_threads():
    return lib().nbo_num_threads()


# ---------------------------------------------------------------- stateful mirror

class OracleSim:
    """Reference GalaxySimulation semantics on numpy arrays (simulation.py:12-196)."""

    def __init__(self, positions, velocities, masses, precision_mode="float64", G=0.001,
                 softening=0.1, dt=0.01, levels=0, codes=None):
        self.mode = mode_code(precision_mode)
        self.levels = levels
        self.G, self.softening, self.dt = G, softening, dt
        self.softening_sq = softening ** 2
        pc, vc, mc = codes or (dtype_code(positions), dtype_code(velocities), dtype_code(masses))
        self.codes = [pc, vc, mc, None]
        self._pos, self._vel, self._mass = as_f64(positions).copy(), as_f64(velocities).copy(), as_f64(masses).copy()
        self.n, self.d = self._pos.shape
        self._acc = np.empty_like(self._pos)
        self._force()
        self.tick = 0

    def _force(self):
        out = np.empty_like(self._pos)
        self.codes[3] = lib().nbo_accelerations(self.n, self.d, self.codes[0], _dp(self._pos),
                                                self.codes[2], _dp(self._mass), self.mode, self.levels,
                                                float(self.G), float(self.softening_sq), 0, self.n, 1,
                                                _dp(out), None, None, None, None)
        self._acc = out

    def _axpy(self, ia, a, ib, b, s):
        out = np.empty_like(a)
        code = lib().nbo_axpy(a.size, self.codes[ia], _dp(a), self.codes[ib], _dp(b), float(s), _dp(out))
        self.codes[ia] = code
        return out

    def step(self):
        self._vel = self._axpy(1, self._vel, 3, self._acc, self.dt / 2)
        self._pos = self._axpy(0, self._pos, 1, self._vel, self.dt)
        self._force()
        self._vel = self._axpy(1, self._vel, 3, self._acc, self.dt / 2)
        self.tick += 1

    def run(self, n):
        for _ in range(n):
            self.step()

    positions = property(lambda s: from_f64(s._pos, s.codes[0]))
    velocities = property(lambda s: from_f64(s._vel, s.codes[1]))
    masses = property(lambda s: from_f64(s._mass, s.codes[2]))
    accelerations = property(lambda s: from_f64(s._acc, s.codes[3]))

    def get_kinetic_energy(self):
        return lib().nbo_kinetic_energy(self.n, self.d, self.codes[1], _dp(self._vel), self.codes[2], _dp(self._mass))

    def get_potential_energy(self):
        return lib().nbo_potential_energy(self.n, self.d, self.codes[0], _dp(self._pos), self.codes[2],
                                          _dp(self._mass), float(self.G), float(self.softening_sq), 0, self.n)

    def get_total_energy(self):
        return self.get_kinetic_energy() + self.get_potential_energy()
