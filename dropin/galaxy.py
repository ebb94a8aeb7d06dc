"""Drop-in for the reference's galaxy.py."""
from nbody_cosmological_simulation_amd.galaxy import (  # noqa: F401
    create_disk_galaxy, create_test_galaxy, nfw_enclosed_mass, create_galaxy_with_halo)
