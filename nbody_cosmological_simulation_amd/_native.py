"""ctypes binding of libnbody_amd.so (C-ABI declared in include/nbody_amd.h).

This is the only place the package touches the native library.  There is no CPU
fallback: if the shared object is missing, or no HIP device is present, every
constructor raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NBODY_LIB: load an alternative build of the same C-ABI (A/B experiments: `make exp EXPFLAGS=-D...`)
LIB_PATH = os.environ.get("NBODY_LIB") or os.path.join(_HERE, "libnbody_amd.so")

NB_F16, NB_BF16, NB_F32, NB_F64 = 0, 1, 2, 3
NB_FLAG_PROFILE = 1
NB_FLAG_CUSTOM_FORCEQ = 2
NB_FLAG_NO_COMM = 4
NB_FLAG_SHARD_TIMING = 8
NB_FLAG_F64_STORAGE = 16

MODE_CODES = {
    "float64": 0, "float32": 1, "bfloat16": 2, "float16": 3,
    "int8_sim": 4, "int4_sim": 5, "custom": 6,
}

# every symbol include/nbody_amd.h declares (checked by tests/test_cabi_symbols.py)
EXPORTS = [
    "nb_create", "nb_destroy", "nb_cache_trim", "nb_set_params", "nb_set_state", "nb_get_state", "nb_state_dtypes",
    "nb_set_accelerations", "nb_compute_accelerations", "nb_step", "nb_kick_drift", "nb_kick",
    "nb_energy", "nb_quant_debug", "nb_quant_bins_rows", "nb_quant_bin_sums", "nb_comm_info", "nb_quantize_distance_squared", "nb_quantize_force",
    "nb_grid_quantize", "nb_grid_quantize_safe", "nb_comm_unique_id", "nb_comm_init", "nb_comm_ready",
    "nb_comm_shutdown", "nb_comm_quiesce", "nb_comm_p2p_export", "nb_comm_p2p_import", "nb_comm_p2p_selftest",
    "nb_comm_p2p_enable", "nb_comm_p2p_state", "nb_comm_p2p_allreduce", "nb_comm_allreduce_time", "nb_comm_p2p_virtual_test", "nb_plan_debug", "nb_set_hook_stream", "nb_metrics", "nb_metrics_tensors",
    "nb_kernel_time", "nb_force_kernel_name", "nb_synchronize", "nb_device_count", "nb_abi_version", "nb_last_error",
]


class NbConfig(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("dim", C.c_int32), ("mode", C.c_int32), ("levels", C.c_int32),
        ("G", C.c_double), ("softening_sq", C.c_double), ("dt", C.c_double),
        ("device", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32), ("flags", C.c_int32),
    ]


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libnbody_amd error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libnbody_amd.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pi32 = C.POINTER(C.c_int32)
    pdbl = C.POINTER(C.c_double)
    sig = {
        "nb_create": ([C.POINTER(vp), C.POINTER(NbConfig)], C.c_int),
        "nb_destroy": ([vp], C.c_int),
        "nb_set_params": ([vp, dbl, dbl, dbl], C.c_int),
        "nb_set_state": ([vp, vp, vp, vp, C.c_int, C.c_int], C.c_int),
        "nb_get_state": ([vp, vp, vp, vp, vp, C.c_int], C.c_int),
        "nb_state_dtypes": ([vp, pi32], C.c_int),
        "nb_set_accelerations": ([vp, vp, C.c_int, C.c_int], C.c_int),
        "nb_compute_accelerations": ([vp], C.c_int),
        "nb_step": ([vp, i32], C.c_int),
        "nb_kick_drift": ([vp], C.c_int),
        "nb_kick": ([vp], C.c_int),
        "nb_energy": ([vp, pdbl, pdbl], C.c_int),
        "nb_quant_debug": ([vp, pdbl, vp, vp], C.c_int),
        "nb_quant_bins_rows": ([vp, i32, i32, vp], C.c_int),
        "nb_quant_bin_sums": ([vp, i32, vp, vp, pdbl], C.c_int),
        "nb_comm_info": ([pi32], C.c_int),
        "nb_cache_trim": ([C.POINTER(C.c_int64)], C.c_int),
        "nb_quantize_distance_squared": ([C.c_int, vp, vp, i64, C.c_int, C.c_int, C.c_int, dbl, C.c_int, pi32], C.c_int),
        "nb_quantize_force": ([C.c_int, vp, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, pi32], C.c_int),
        "nb_grid_quantize": ([C.c_int, vp, vp, i64, C.c_int, C.c_int, C.c_int], C.c_int),
        "nb_grid_quantize_safe": ([C.c_int, vp, vp, i64, C.c_int, C.c_int, dbl, C.c_int], C.c_int),
        "nb_comm_unique_id": ([vp, pi32], C.c_int),
        "nb_comm_init": ([vp, vp, i32], C.c_int),
        "nb_set_hook_stream": ([C.c_int, vp, C.c_int], C.c_int),
        "nb_metrics": ([vp, i32, vp, dbl, dbl, i32, pdbl, C.POINTER(C.c_int64), pdbl], C.c_int),
        "nb_metrics_tensors": ([C.c_int, vp, vp, vp, i32, i32, C.c_int, C.c_int, dbl, i32, vp, dbl, dbl, i32, pdbl,
                                C.POINTER(C.c_int64), pdbl], C.c_int),
        "nb_comm_ready": ([], C.c_int),
        "nb_comm_shutdown": ([], C.c_int),
        "nb_comm_quiesce": ([], C.c_int),
        "nb_comm_p2p_export": ([C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.POINTER(C.c_int32)], C.c_int),
        "nb_comm_p2p_import": ([C.c_void_p, C.c_int32], C.c_int),
        "nb_comm_p2p_selftest": ([C.c_int32, C.c_double], C.c_int),
        "nb_comm_p2p_enable": ([C.c_int32], C.c_int),
        "nb_comm_p2p_state": ([], C.c_int),
        "nb_comm_p2p_allreduce": ([C.c_void_p, C.c_int64, C.c_int32, C.c_double], C.c_int),
        "nb_comm_allreduce_time": ([C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)], C.c_int),
        "nb_comm_p2p_virtual_test": ([C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                      C.POINTER(C.c_int32), C.POINTER(C.c_double)], C.c_int),
        "nb_plan_debug": ([C.POINTER(NbConfig), i32, i32, i32, pi32, pi32, i64, pi32, pi32, pi32, pi32, pi32], C.c_int),
        "nb_kernel_time": ([vp, pdbl, pi32], C.c_int),
        "nb_force_kernel_name": ([vp], C.c_char_p),
        "nb_synchronize": ([vp], C.c_int),
        "nb_device_count": ([pi32], C.c_int),
        "nb_abi_version": ([], C.c_int),
        "nb_last_error": ([], C.c_char_p),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = res
    _lib = L
    return L


def last_error() -> str:
    return lib().nb_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != 0:
        raise NativeError(rc, lib().nb_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int32(0)
    rc = lib().nb_device_count(C.byref(n))
    return n.value if rc == 0 else 0
