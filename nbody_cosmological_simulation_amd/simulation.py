"""N-body engine front-end -- same Python surface as the reference's simulation.py.

Reference: simulation.py:12-250 (`GalaxySimulation`, `run_comparison`).  Constructor
arguments, attribute names, method names, defaults, dtype behaviour and error behaviour
(NaN/Inf propagate silently) are the reference's; everything underneath
`_compute_accelerations` / `step` / `get_*_energy` is the HIP library behind the C-ABI of
include/nbody_amd.h.  No tensor arithmetic of the hot path runs in PyTorch: torch tensors
are only the containers callers read and write.

State ownership.  The authoritative state lives in the library's device buffers.  The
attributes `positions`, `velocities`, `masses`, `accelerations` are torch tensors
materialised on demand (on `self.device`, with the dtype the reference would show at that
moment) and cached until the next step.  A caller that edits such a tensor in place
(omega_point_test.py:738) or rebinds the attribute is detected through the tensor's version
counter / identity and the array is re-uploaded before the next native call.

Subclass rule: `__init__` and `step` dispatch through
`self._compute_accelerations()`; when a subclass overrides it (sensitivity_test.py:55-76 and
17 clones) its returned tensor becomes the force and only the two kicks and the drift run
natively; otherwise the whole step is one native call.
"""
import ctypes as C
import os
from typing import Callable

import torch

from . import _native as N
from . import runtime
from .quantization import PrecisionMode, mode_code, _TORCH_TO_NB, _NB_TO_TORCH

_ARRAYS = ("positions", "velocities", "masses", "accelerations")
_IDX = {"positions": 0, "velocities": 1, "masses": 2, "accelerations": 3}


class GalaxySimulation:
    """N-body gravitational simulation with configurable precision (reference simulation.py:12)."""

    def __init__(
        self,
        positions: torch.Tensor,
        velocities: torch.Tensor,
        masses: torch.Tensor,
        precision_mode: PrecisionMode = PrecisionMode.FLOAT64,
        G: float = 0.001,
        softening: float = 0.1,
        dt: float = 0.01,
        device: torch.device = None,
        custom_levels: int = None,
        profile: bool = False,
        shard: tuple = None,
    ):
        # reference simulation.py:54-66
        self.device = torch.device(device) if device is not None else positions.device
        self.precision_mode = precision_mode
        self.G = G
        self.softening = softening
        self.softening_sq = softening ** 2
        self.dt = dt
        self.num_stars = len(masses)
        self.custom_levels = custom_levels

        if positions.dim() != 2 or positions.shape[1] not in (2, 3):
            raise ValueError(f"positions must be (N, 2) or (N, 3), got {tuple(positions.shape)}")
        if positions.shape[0] != self.num_stars or velocities.shape != positions.shape:
            raise ValueError("positions, velocities and masses disagree on N")

        self._handle = C.c_void_p()
        self._serial = 0            # bumped whenever the device state changes (energy memo key)
        self._energy_memo = {}
        self._empty = self.num_stars == 0
        if self._empty:
            self._init_empty(positions, velocities, masses)
            return
        self._cfg_dim = int(positions.shape[1])
        self._cache = {}        # name -> [tensor, version, dirty]
        self._native_acc = None
        rank, world = runtime.rank_world()
        flags = N.NB_FLAG_PROFILE if profile else 0
        fake = os.environ.get("NBODY_SHARD_TIMING")       # "r/P": time one shard of P on a single GPU
        if fake and shard is None and runtime.force_comm():
            rank, world = (int(v) for v in fake.split("/"))
            flags |= N.NB_FLAG_SHARD_TIMING
        if shard is not None:
            # explicit (rank, world) without a communicator: this handle only produces the partial
            # sums of its source block (single-GPU shard tests / caller-side reduction)
            rank, world = int(shard[0]), int(shard[1])
            flags |= N.NB_FLAG_NO_COMM
        if torch.float64 in (positions.dtype, velocities.dtype, masses.dtype):
            flags |= N.NB_FLAG_F64_STORAGE     # torch promotes such a run to fp64 (mass product / first kick)
        if self.device.type == "cuda":
            dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        else:
            dev = runtime.default_hip_device()
        cfg = N.NbConfig(
            n=self.num_stars, dim=int(positions.shape[1]), mode=mode_code(precision_mode),
            levels=int(custom_levels or 0), G=float(G), softening_sq=float(self.softening_sq), dt=float(dt),
            device=int(dev), rank=rank, nranks=world,
            flags=flags)
        N.check(N.lib().nb_create(C.byref(self._handle), C.byref(cfg)))
        if shard is None and (world > 1 or runtime.force_comm()):
            runtime.attach_communicator(self._handle)      # one communicator per process, shared

        # simulation.py:63-65: clone -> device.  The clone is the upload itself.
        self._upload_initial(positions, velocities, masses)

        # simulation.py:69 (virtual call: subclasses may override _compute_accelerations)
        self.accelerations = self._compute_accelerations()
        self.tick = 0

    # ------------------------------------------------------------------ N = 0
    def _init_empty(self, positions, velocities, masses):
        """Zero stars: nothing to launch.  Upstream returns empty tensors with the usual dtype
        promotion, 0.0 / -0.0 energies, and the grid modes fail in `min()` of an empty tensor."""
        if self.precision_mode in (PrecisionMode.INT8_SIM, PrecisionMode.INT4_SIM, PrecisionMode.CUSTOM):
            raise RuntimeError("min(): cannot quantise an empty distance tensor (0 stars)")
        self._cfg_dim = int(positions.shape[1])
        self._cache = {}
        self._native_acc = None
        acc_dtype = torch.float64 if self.precision_mode == PrecisionMode.FLOAT64 else \
            torch.promote_types(torch.promote_types(torch.float32, masses.dtype), positions.dtype)
        self._empty_state = {
            "positions": positions.clone().to(self.device), "velocities": velocities.clone().to(self.device),
            "masses": masses.clone().to(self.device),
            "accelerations": torch.zeros((0, self._cfg_dim), dtype=acc_dtype, device=self.device)}
        self.tick = 0

    def _step_empty(self):
        st = self._empty_state
        st["velocities"] = st["velocities"].to(torch.promote_types(st["velocities"].dtype, st["accelerations"].dtype))
        st["positions"] = st["positions"].to(torch.promote_types(st["positions"].dtype, st["velocities"].dtype))
        self.tick += 1

    # ------------------------------------------------------------------ native plumbing
    def close(self):
        """Release the native handle (device buffers, streams) now instead of at garbage collection.  Never a
        collective: the RCCL communicator belongs to the process (runtime.shutdown() destroys it)."""
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                N.lib().nb_destroy(h)
            except Exception:
                pass
            h.value = None

    def __del__(self):
        self.close()

    def _upload_initial(self, positions, velocities, masses):
        """The three initial arrays: ONE native call (one stream synchronisation) when they share dtype and residence --
        what every script of the reference passes -- else array by array."""
        ts = [t.detach() for t in (positions, velocities, masses)]
        same = (len({t.dtype for t in ts}) == 1 and len({t.device.type for t in ts}) == 1 and ts[0].dtype in _TORCH_TO_NB
                and tuple(ts[1].shape) == tuple(ts[0].shape) == (self.num_stars, self._cfg_dim)
                and tuple(ts[2].shape) == (self.num_stars,))
        if not same:
            self._upload("positions", positions)
            self._upload("velocities", velocities)
            self._upload("masses", masses)
            return
        self._serial += 1
        ts = [t.contiguous() for t in ts]
        on_device = ts[0].device.type == "cuda"
        if on_device:
            for dev in {t.device for t in ts}:
                torch.cuda.current_stream(dev).synchronize()
        N.check(N.lib().nb_set_state(self._handle, C.c_void_p(ts[0].data_ptr()), C.c_void_p(ts[1].data_ptr()),
                                     C.c_void_p(ts[2].data_ptr()), _TORCH_TO_NB[ts[0].dtype], int(on_device)))

    def _upload(self, name, tensor):
        self._serial += 1
        t = tensor.detach()
        if t.dtype not in _TORCH_TO_NB:
            raise TypeError(f"{name}: unsupported dtype {t.dtype}")
        # the native side copies exactly N (x D) elements from this pointer: a rebound tensor of another shape
        # (sim.positions = sim.positions[:100], a wrong-D override result) must fail here, like the broadcast
        # error it would raise upstream, not read out of bounds
        want = (self.num_stars,) if name == "masses" else (self.num_stars, self._cfg_dim)
        if tuple(t.shape) != want:
            raise ValueError(f"{name} must have shape {want}, got {tuple(t.shape)}")
        t = t.contiguous()
        on_device = t.device.type == "cuda"
        if on_device:
            torch.cuda.current_stream(t.device).synchronize()
        ptr = C.c_void_p(t.data_ptr())
        code = _TORCH_TO_NB[t.dtype]
        L = N.lib()
        if name == "accelerations":
            N.check(L.nb_set_accelerations(self._handle, ptr, code, int(on_device)))
        else:
            args = {"positions": (ptr, None, None), "velocities": (None, ptr, None), "masses": (None, None, ptr)}[name]
            N.check(L.nb_set_state(self._handle, args[0], args[1], args[2], code, int(on_device)))

    def _download(self, name):
        dts = (C.c_int32 * 4)()
        N.check(N.lib().nb_state_dtypes(self._handle, dts))
        dtype = _NB_TO_TORCH[dts[_IDX[name]]]
        shape = (self.num_stars,) if name == "masses" else (self.num_stars, self._dim())
        out = torch.empty(shape, dtype=dtype, device=self.device)
        on_device = out.device.type == "cuda"
        ptr = C.c_void_p(out.data_ptr())
        args = [None, None, None, None]
        args[{"positions": 0, "velocities": 1, "accelerations": 2, "masses": 3}[name]] = ptr
        N.check(N.lib().nb_get_state(self._handle, args[0], args[1], args[2], args[3], int(on_device)))
        return out

    def _prefetch(self, names):
        """Download every array of `names` that is not cached yet in ONE native call (one stream synchronisation
        instead of one per array: get_state reads three)."""
        missing = [n for n in names if n not in self._cache]
        if len(missing) < 2 or self._empty:
            return
        dts = (C.c_int32 * 4)()
        N.check(N.lib().nb_state_dtypes(self._handle, dts))
        outs, args = {}, [None, None, None, None]
        for n in missing:
            shape = (self.num_stars,) if n == "masses" else (self.num_stars, self._dim())
            outs[n] = torch.empty(shape, dtype=_NB_TO_TORCH[dts[_IDX[n]]], device=self.device)
            args[{"positions": 0, "velocities": 1, "accelerations": 2, "masses": 3}[n]] = C.c_void_p(outs[n].data_ptr())
        N.check(N.lib().nb_get_state(self._handle, args[0], args[1], args[2], args[3], int(self.device.type == "cuda")))
        for n, t in outs.items():
            self._cache[n] = [t, t._version, False]

    def _dim(self):
        return self._cfg_dim

    def _get(self, name):
        if self._empty:
            return self._empty_state[name]
        ent = self._cache.get(name)
        if ent is None:
            t = self._download(name)
            self._cache[name] = [t, t._version, False]
            return t
        return ent[0]

    def _set(self, name, value):
        if not isinstance(value, torch.Tensor):
            raise TypeError(f"{name} must be a torch.Tensor")
        if self._empty:
            self._empty_state[name] = value
            return
        if name == "accelerations" and value is self._native_acc:
            self._cache[name] = [value, value._version, False]     # produced by the library itself
        else:
            self._cache[name] = [value, value._version, True]

    positions = property(lambda s: s._get("positions"), lambda s, v: s._set("positions", v))
    velocities = property(lambda s: s._get("velocities"), lambda s, v: s._set("velocities", v))
    masses = property(lambda s: s._get("masses"), lambda s, v: s._set("masses", v))
    accelerations = property(lambda s: s._get("accelerations"), lambda s, v: s._set("accelerations", v))

    def _flush(self, names=_ARRAYS):
        """Push caller-side edits (in-place writes or rebinding) down to the device state."""
        for name in names:
            ent = self._cache.get(name)
            if ent is None:
                continue
            t, ver, dirty = ent
            if dirty or t._version != ver:
                self._upload(name, t)
                ent[1], ent[2] = t._version, False
        # attribute writes such as `sim.dt = 0.02` (simulation.py reads them at every use): pushed down when they change
        params = (float(self.G), float(self.softening_sq), float(self.dt))
        if params != getattr(self, "_params_sent", None):
            N.check(N.lib().nb_set_params(self._handle, *params))
            self._params_sent = params

    def _invalidate(self, *names):
        self._serial += 1
        for name in names:
            self._cache.pop(name, None)

    def _native_metrics_ready(self) -> bool:
        """metrics.collect_metrics: push pending edits down so the diagnostics can run on the device state."""
        if self._empty:
            return False
        self._flush(("positions", "velocities", "masses"))
        return True

    def _overridden(self):
        return type(self)._compute_accelerations is not GalaxySimulation._compute_accelerations

    # ------------------------------------------------------------------ hot path
    def _compute_accelerations(self) -> torch.Tensor:
        """All-pairs softened gravity with the precision hook of `self.precision_mode`
        (reference simulation.py:74-118) -- one native call, result returned as a tensor."""
        self._flush(("positions", "masses"))
        N.check(N.lib().nb_compute_accelerations(self._handle))
        self._invalidate("accelerations")
        acc = self._download("accelerations")
        self._native_acc = acc
        return acc

    compute_forces = _compute_accelerations   # north-star alias

    def step(self):
        """One kick-drift-kick leapfrog step (reference simulation.py:120-143)."""
        if self._empty:
            return self._step_empty()
        L = N.lib()
        if self._overridden():
            self._flush()
            N.check(L.nb_kick_drift(self._handle))                   # :132, :135
            self._invalidate("positions", "velocities")
            self.accelerations = self._compute_accelerations()        # :138 (subclass code)
            self._flush(("accelerations",))
            N.check(L.nb_kick(self._handle))                         # :141
            self._invalidate("velocities")
        else:
            self._flush()
            N.check(L.nb_step(self._handle, 1))
            self._invalidate("positions", "velocities", "accelerations")
        self.tick += 1

    def run(self, num_ticks: int, callback: Callable = None, callback_interval: int = 100):
        """Run `num_ticks` steps; `callback(self, self.tick)` every `callback_interval`
        (reference simulation.py:145-158)."""
        fused = (not self._overridden()) and type(self).step is GalaxySimulation.step and not self._empty
        if not fused:
            for t in range(num_ticks):
                self.step()
                if callback and (t + 1) % callback_interval == 0:
                    callback(self, self.tick)
            return
        # whole stretches between callbacks stay on the device: one native call each
        done = 0
        while done < num_ticks:
            if callback:
                nxt = min(num_ticks, (done // callback_interval + 1) * callback_interval)
            else:
                nxt = num_ticks
            self._flush()
            N.check(N.lib().nb_step(self._handle, nxt - done))
            self._invalidate("positions", "velocities", "accelerations")
            self.tick += nxt - done
            done = nxt
            if callback and done % callback_interval == 0:
                callback(self, self.tick)

    def get_state(self) -> dict:
        """Current state as clones (reference simulation.py:160-168)."""
        if not self._empty:
            self._prefetch(("positions", "velocities", "masses"))
        return {
            "positions": self.positions.clone(),
            "velocities": self.velocities.clone(),
            "masses": self.masses.clone(),
            "tick": self.tick,
            "precision_mode": self.precision_mode.value,
        }

    def get_kinetic_energy(self) -> float:
        """sum(0.5 * m * v^2) (reference simulation.py:170-174)."""
        if self._empty:
            return 0.0
        self._flush(("velocities", "masses"))
        key = ("ke", self._serial, float(self.G))
        if key not in self._energy_memo:       # main.py asks for the same energy up to 3x per callback
            ke = C.c_double()
            N.check(N.lib().nb_energy(self._handle, C.byref(ke), None))
            self._energy_memo = {k: v for k, v in self._energy_memo.items() if k[1] == self._serial}
            self._energy_memo[key] = ke.value
        return self._energy_memo[key]

    def get_potential_energy(self) -> float:
        """-G * sum_{i<j} m_i m_j / sqrt(r_ij^2 + eps^2) (reference simulation.py:176-192)."""
        if self._empty:
            return -0.0
        self._flush(("positions", "masses"))
        key = ("pe", self._serial, float(self.G), float(self.softening_sq))
        if key not in self._energy_memo:       # O(N^2): memoised until the state or G / softening change
            pe = C.c_double()
            N.check(N.lib().nb_energy(self._handle, None, C.byref(pe)))
            self._energy_memo = {k: v for k, v in self._energy_memo.items() if k[1] == self._serial}
            self._energy_memo[key] = pe.value
        return self._energy_memo[key]

    def get_total_energy(self) -> float:
        """Total mechanical energy (reference simulation.py:194-196).  When neither part is memoised both come from ONE
        native call (one stream synchronisation instead of two)."""
        if not self._empty:
            self._flush(("positions", "velocities", "masses"))
            kk = ("ke", self._serial, float(self.G))
            kp = ("pe", self._serial, float(self.G), float(self.softening_sq))
            if kk not in self._energy_memo and kp not in self._energy_memo:
                ke, pe = C.c_double(), C.c_double()
                N.check(N.lib().nb_energy(self._handle, C.byref(ke), C.byref(pe)))
                self._energy_memo = {k: v for k, v in self._energy_memo.items() if k[1] == self._serial}
                self._energy_memo[kk] = ke.value
                self._energy_memo[kp] = pe.value
        return self.get_kinetic_energy() + self.get_potential_energy()

    # ------------------------------------------------------------------ extras (not in the reference)
    def synchronize(self):
        N.check(N.lib().nb_synchronize(self._handle))

    def spin_up(self, evaluations: int):
        """Re-evaluate the forces `evaluations` times without touching the state (the result is
        bit-identical every time): lets the GPU reach its sustained clock before a timed region."""
        self._flush(("positions", "masses"))
        for _ in range(max(0, int(evaluations))):
            N.check(N.lib().nb_compute_accelerations(self._handle))

    def kernel_time(self):
        """(total_ms, launches) of the force kernel since the last call (needs profile=True)."""
        ms, n = C.c_double(), C.c_int32()
        N.check(N.lib().nb_kernel_time(self._handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def allreduce_time(self, which: str = "rccl", iters: int = 200):
        """Average microseconds of one all-reduce of this simulation's force vector on the node it runs on
        ("rccl" or "direct"; collective, measurement only).  None when that path is not available."""
        us = C.c_double(0.0)
        rc = N.lib().nb_comm_allreduce_time(self._handle, 1 if which == "direct" else 0, int(iters), C.byref(us))
        return us.value if rc == 0 else None

    def force_kernel_name(self) -> str:
        """Kernel the last force evaluation launched (matches the rocprofv3 kernel-trace rows)."""
        return N.lib().nb_force_kernel_name(self._handle).decode()

    def quant_debug(self, bins: bool = False):
        """Grid internals of the last force evaluation (INT8/INT4/CUSTOM modes)."""
        import numpy as np
        info = (C.c_double * 8)()
        n, d = self.num_stars, self._dim()
        d2 = np.empty((n, n), np.int16) if bins else None
        has_fq = self.precision_mode in (PrecisionMode.INT8_SIM, PrecisionMode.INT4_SIM)
        fb = np.empty((n, d), np.int16) if (bins and has_fq) else None
        N.check(N.lib().nb_quant_debug(self._handle, info,
                                       None if d2 is None else d2.ctypes.data_as(C.c_void_p),
                                       None if fb is None else fb.ctypes.data_as(C.c_void_p)))
        return dict(lmin=info[0], lmax=info[1], fmin=info[2], fmax=info[3], r2max=info[4], d2bins=d2, fbins=fb,
                    fast_path=bool(info[5]), fast_maxdev=info[6], fast_maxrel=info[7])

    def quant_bin_sums(self, which: str = "last"):
        """Quant-bin assignments read out of the PRODUCTION pair loop (include/nbody_amd.h nb_quant_bin_sums): per
        particle p the exact integers sum_q k(p, q) and sum_q k(p, q) * ((q mod 65521) + 1) over its row of the N x N
        bin matrix of quantization.py:119-121.  which: "last" (the path the last evaluation / step took), "tiled"
        (what _compute_accelerations launches) or "small" (the one-launch small-system step kernel)."""
        import numpy as np
        self._flush(("positions", "masses"))
        n = self.num_stars
        s1, s2 = np.empty(n, np.int64), np.empty(n, np.int64)
        info = (C.c_double * 8)()
        N.check(N.lib().nb_quant_bin_sums(self._handle, {"last": 0, "tiled": 1, "small": 2}[which],
                                          s1.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p), info))
        return dict(sum_k=s1, sum_kw=s2, path={1: "sym", 2: "onesided", 3: "small"}[int(info[0])], shape=int(info[1]),
                    uniform_kernel=bool(info[2]), fast_path=bool(info[3]), pairs_table_free=int(info[4]),
                    pairs_table=int(info[5]), levels=int(info[6]))

    def quant_bins_rows(self, i0: int, i1: int):
        """Distance-bin indices of target rows [i0, i1) of the last force evaluation (grid modes), (i1-i0, N) int16."""
        import numpy as np
        out = np.empty((i1 - i0, self.num_stars), np.int16)
        N.check(N.lib().nb_quant_bins_rows(self._handle, int(i0), int(i1), out.ctypes.data_as(C.c_void_p)))
        return out



def run_comparison(
    positions: torch.Tensor,
    velocities: torch.Tensor,
    masses: torch.Tensor,
    modes: list,
    num_ticks: int = 1000,
    callback: Callable = None,
    callback_interval: int = 100,
    **sim_kwargs
) -> dict:
    """Same initial conditions under several precision modes (reference simulation.py:199-250)."""
    results = {}
    for mode in modes:
        print(f"\nRunning simulation with {mode.value} precision...")
        sim = GalaxySimulation(positions.clone(), velocities.clone(), masses.clone(),
                               precision_mode=mode, **sim_kwargs)
        history = {
            "positions": [positions.clone().cpu()],
            "energies": [sim.get_total_energy()],
            "ticks": [0],
        }

        def record_callback(s, tick, history=history):
            history["positions"].append(s.positions.clone().cpu())
            history["energies"].append(s.get_total_energy())
            history["ticks"].append(tick)
            if callback:
                callback(s, tick)

        sim.run(num_ticks, callback=record_callback, callback_interval=callback_interval)
        results[mode.value] = {
            "final_state": sim.get_state(),
            "history": history,
            "simulation": sim,
        }
    return results
