"""Drop-in for the reference's simulation.py: same names, MI355X engine underneath."""
from nbody_cosmological_simulation_amd.simulation import GalaxySimulation, run_comparison  # noqa: F401
from nbody_cosmological_simulation_amd.quantization import (PrecisionMode, quantize_distance_squared,  # noqa: F401
                                                            quantize_force)
