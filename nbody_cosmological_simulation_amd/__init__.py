"""MI355X-native direct-summation N-body engine behind the reference's GalaxySimulation API.

Hot path: hand-written HIP kernels for gfx950 (csrc/) behind the C-ABI of include/nbody_amd.h,
called through ctypes (_native.py).  See DESIGN.md.
"""
from .quantization import (PrecisionMode, quantize_distance_squared, quantize_force, _grid_quantize,
                           _grid_quantize_safe, get_mode_from_string, describe_mode)
from .simulation import GalaxySimulation, run_comparison

__all__ = ["GalaxySimulation", "run_comparison", "PrecisionMode", "quantize_distance_squared",
           "quantize_force", "_grid_quantize", "_grid_quantize_safe", "get_mode_from_string",
           "describe_mode"]
