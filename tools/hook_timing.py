"""Tensor-level hooks (quantization.py:21-157 through the C-ABI) on N x N tensors resident on the GPU: time per call and
the HBM traffic it implies (read + write of the tensor = 8 bytes per element, the grid modes read it twice)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_cosmological_simulation_amd as nb
from nbody_cosmological_simulation_amd import quantization as Q

for n in ([int(a) for a in sys.argv[1:]] or [1024, 4096, 8192]):
    t = (torch.rand(n, n, device="cuda") * 100 + 0.01).float()
    for name, fn in (("quantize_distance_squared INT8", lambda: Q.quantize_distance_squared(t, nb.PrecisionMode.INT8_SIM)),
                     ("quantize_distance_squared FLOAT16", lambda: Q.quantize_distance_squared(t, nb.PrecisionMode.FLOAT16)),
                     ("_grid_quantize_safe 64", lambda: Q._grid_quantize_safe(t, 64)),
                     ("_grid_quantize 256", lambda: Q._grid_quantize(t, 256))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"N={n} {name}: {dt * 1e6:.1f} us per call, {t.numel() * 8 / dt / 1e9:.0f} GB/s (8 B per element)")
