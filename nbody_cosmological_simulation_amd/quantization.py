"""Precision-mode hooks -- same surface as the reference's quantization.py.

Reference: quantization.py:10-189.  The enum, the string helpers and the function
signatures/defaults are identical; the tensor-in/tensor-out functions run as HIP kernels
through the C-ABI (include/nbody_amd.h: nb_quantize_distance_squared, nb_quantize_force,
nb_grid_quantize, nb_grid_quantize_safe).  Inside GalaxySimulation the same hooks are fused
into the pair loop (csrc/nb_force.hip); these standalone versions exist for the subclasses
that override `_compute_accelerations` (sensitivity_test.py:55-76 and its clones).

No CPU fallback: without the HIP library / a GPU these functions raise.
"""
import ctypes as C
from enum import Enum

import torch

from . import _native as N


class PrecisionMode(Enum):
    """Available precision modes (reference quantization.py:10-18)."""
    FLOAT64 = "float64"
    FLOAT32 = "float32"
    BFLOAT16 = "bfloat16"
    FLOAT16 = "float16"
    INT8_SIM = "int8_sim"
    INT4_SIM = "int4_sim"
    CUSTOM = "custom"


_TORCH_TO_NB = {torch.float16: N.NB_F16, torch.bfloat16: N.NB_BF16,
                torch.float32: N.NB_F32, torch.float64: N.NB_F64}
_NB_TO_TORCH = {v: k for k, v in _TORCH_TO_NB.items()}


def mode_code(mode) -> int:
    return N.MODE_CODES[mode.value if isinstance(mode, Enum) else str(mode)]


def _hip_device_for(t: torch.Tensor) -> int:
    from .runtime import default_hip_device
    return t.device.index if t.device.type == "cuda" and t.device.index is not None else default_hip_device()


def _run_hook(t: torch.Tensor, out_dtype, call):
    """Run a tensor-level hook.  `call(dev, in_ptr, out_ptr, count, nb_dtype, on_device)`."""
    if t.dtype not in (torch.float32, torch.float64):
        raise NotImplementedError(f"precision hooks take float32/float64 tensors (got {t.dtype})")
    src = t.contiguous()
    on_device = src.device.type == "cuda"
    out = torch.empty(src.shape, dtype=out_dtype, device=src.device)
    if src.numel() == 0:
        return out
    # device tensors: the kernels are queued on torch's current stream (no synchronisation, no allocation: the
    # library keeps a per-device scratch); host tensors: staged through that scratch, the call waits for the copy back
    with hook_stream(src):
        N.check(call(_hip_device_for(src), C.c_void_p(src.data_ptr()), C.c_void_p(out.data_ptr()),
                     src.numel(), _TORCH_TO_NB[src.dtype], int(on_device)))
    return out


class hook_stream:
    """Tell the library which stream the calling thread's device work runs on (nb_set_hook_stream) for the duration
    of a handle-less call: torch's current stream of the tensor's device, or the default for host tensors."""

    def __init__(self, tensor):
        self.dev = _hip_device_for(tensor)
        self.cuda = tensor.device.type == "cuda"
        self.tensor = tensor

    def __enter__(self):
        if self.cuda:
            st = torch.cuda.current_stream(self.tensor.device).cuda_stream
            N.check(N.lib().nb_set_hook_stream(self.dev, C.c_void_p(st), 1))
        return self

    def __exit__(self, *exc):
        if self.cuda:
            N.lib().nb_set_hook_stream(self.dev, None, 0)
        return False


def _grid_quantize(tensor: torch.Tensor, levels: int) -> torch.Tensor:
    """Linear grid over the tensor's global min/max (reference quantization.py:74-88)."""
    L = N.lib()
    return _run_hook(tensor, tensor.dtype,
                     lambda dev, i, o, n, dt, od: L.nb_grid_quantize(dev, i, o, n, dt, int(levels), od))


def _grid_quantize_safe(tensor: torch.Tensor, levels: int, min_val: float = 0.01) -> torch.Tensor:
    """Log-space grid above a floor (reference quantization.py:91-127)."""
    L = N.lib()
    return _run_hook(tensor, tensor.dtype,
                     lambda dev, i, o, n, dt, od: L.nb_grid_quantize_safe(dev, i, o, n, dt, int(levels),
                                                                           float(min_val), od))


def quantize_distance_squared(dist_sq: torch.Tensor, mode: PrecisionMode, custom_levels: int = None,
                              min_dist_sq: float = 0.01) -> torch.Tensor:
    """Precision degradation of r^2 (reference quantization.py:21-71)."""
    if not isinstance(mode, PrecisionMode):
        return dist_sq                                   # reference falls through to `return dist_sq`
    L = N.lib()
    code = mode_code(mode)
    if mode == PrecisionMode.FLOAT64:
        odt = torch.float64
    elif mode in (PrecisionMode.FLOAT32, PrecisionMode.BFLOAT16, PrecisionMode.FLOAT16):
        odt = torch.float32
    else:
        odt = dist_sq.dtype
    got = C.c_int32(-1)
    out = _run_hook(dist_sq, odt,
                    lambda dev, i, o, n, dt, od: L.nb_quantize_distance_squared(
                        dev, i, o, n, dt, code, int(custom_levels or 0), float(min_dist_sq), od, C.byref(got)))
    return out


def quantize_force(force: torch.Tensor, mode: PrecisionMode, custom_levels: int = None) -> torch.Tensor:
    """Force quantisation (reference quantization.py:130-157)."""
    if not isinstance(mode, PrecisionMode):
        return force
    if mode in (PrecisionMode.FLOAT64, PrecisionMode.FLOAT32):
        return force                                     # identity returns the SAME tensor upstream
    L = N.lib()
    code = mode_code(mode)
    odt = torch.float32 if mode in (PrecisionMode.BFLOAT16, PrecisionMode.FLOAT16) else force.dtype
    got = C.c_int32(-1)
    return _run_hook(force, odt,
                     lambda dev, i, o, n, dt, od: L.nb_quantize_force(dev, i, o, n, dt, code,
                                                                       int(custom_levels or 0), od, C.byref(got)))


def get_mode_from_string(mode_str: str) -> PrecisionMode:
    """String -> PrecisionMode with the reference's aliases; unknown -> FLOAT64 (quantization.py:160-175)."""
    mode_map = {
        "float64": PrecisionMode.FLOAT64,
        "float32": PrecisionMode.FLOAT32,
        "bfloat16": PrecisionMode.BFLOAT16, "bf16": PrecisionMode.BFLOAT16,
        "float16": PrecisionMode.FLOAT16, "fp16": PrecisionMode.FLOAT16,
        "int8": PrecisionMode.INT8_SIM, "int8_sim": PrecisionMode.INT8_SIM,
        "int4": PrecisionMode.INT4_SIM, "int4_sim": PrecisionMode.INT4_SIM,
        "custom": PrecisionMode.CUSTOM,
    }
    return mode_map.get(mode_str.lower(), PrecisionMode.FLOAT64)


def describe_mode(mode: PrecisionMode) -> str:
    """Human-readable description (quantization.py:178-189)."""
    descriptions = {
        PrecisionMode.FLOAT64: "64-bit float (baseline)",
        PrecisionMode.FLOAT32: "32-bit float (standard GPU)",
        PrecisionMode.BFLOAT16: "Brain Float 16 (AI precision, fast on RTX)",
        PrecisionMode.FLOAT16: "16-bit float (half precision)",
        PrecisionMode.INT8_SIM: "Simulated 8-bit (256 levels)",
        PrecisionMode.INT4_SIM: "Simulated 4-bit (16 levels)",
        PrecisionMode.CUSTOM: "Custom quantization levels",
    }
    return descriptions.get(mode, "Unknown mode")
