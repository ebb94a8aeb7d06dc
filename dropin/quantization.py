"""Drop-in for the reference's quantization.py."""
from nbody_cosmological_simulation_amd.quantization import (  # noqa: F401
    PrecisionMode, quantize_distance_squared, quantize_force, _grid_quantize, _grid_quantize_safe,
    get_mode_from_string, describe_mode)
