"""Synthetic initial conditions -- same surface as the reference's galaxy.py.

Reference: galaxy.py:10-211 (formulas restated in SURVEY.md Appendix B).  O(N) host-side
set-up work, not part of the accelerated path: plain torch on the CPU, moved to `device`
at the end.  The random draws happen in the reference's order (radii, angles, dispersion)
from torch's global generator, so `torch.manual_seed(s)` before a call reproduces the
reference's galaxy for the same seed; the optional `seed=` keyword uses a private
generator instead (bench.py / tests).
"""
import math

import torch


def _resolve_device(device):
    if device is None:
        return torch.device("cuda" if torch.cuda.is_available() else "cpu")
    return torch.device(device)


def _rng(seed):
    if seed is None:
        return None
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def _enclosed_mass_disk(radii, total_mass, galaxy_radius, core_mass_fraction, scale, max_r):
    """Bulge (r < 0.2 R_g, quadratic) + exponential-disk enclosed mass (galaxy.py:62-76)."""
    core_radius = galaxy_radius * 0.2
    inner = radii < core_radius
    bulge = core_mass_fraction * total_mass * (radii / core_radius) ** 2
    disk = (1 - core_mass_fraction) * total_mass * (
        1 - (1 + radii / scale) * torch.exp(-radii / scale)) / (1 - 2 * math.exp(-max_r / scale))
    return torch.where(inner, bulge, core_mass_fraction * total_mass + disk)


def _disk(num_stars, galaxy_radius, core_mass_fraction, gen):
    """Body of create_disk_galaxy on the CPU; draws radii, angles, dispersion in that order."""
    G = 0.001                                   # hard-coded upstream (galaxy.py:59)
    scale = galaxy_radius / 3.0
    max_r = galaxy_radius * 2.0

    u = torch.rand(num_stars, generator=gen)
    radii = -scale * torch.log(1 - u * (1 - math.exp(-max_r / scale)))
    radii = torch.clamp(radii, min=0.1, max=max_r)
    angles = torch.rand(num_stars, generator=gen) * 2 * math.pi

    cos_a, sin_a = torch.cos(angles), torch.sin(angles)
    positions = torch.stack((radii * cos_a, radii * sin_a), dim=1)
    masses = torch.ones(num_stars)
    total_mass = num_stars * 1.0

    enclosed = _enclosed_mass_disk(radii, total_mass, galaxy_radius, core_mass_fraction, scale, max_r)
    v_circ = torch.sqrt(G * enclosed / radii.clamp(min=0.1))
    dispersion = 0.1 * v_circ.mean()
    velocities = torch.stack((-v_circ * sin_a, v_circ * cos_a), dim=1)
    velocities = velocities + torch.randn(num_stars, 2, generator=gen) * dispersion
    return positions, velocities, masses


def create_disk_galaxy(num_stars: int = 5000, galaxy_radius: float = 10.0, core_mass_fraction: float = 0.3,
                       device: torch.device = None, seed: int = None):
    """Exponential disk + central bulge on near-circular orbits (reference galaxy.py:10-92).

    Returns (positions (N,2), velocities (N,2), masses (N,)) in float32 on `device`.
    """
    device = _resolve_device(device)
    pos, vel, mass = _disk(num_stars, galaxy_radius, core_mass_fraction, _rng(seed))
    return pos.to(device), vel.to(device), mass.to(device)


def create_test_galaxy(num_stars: int = 1000, device: torch.device = None, seed: int = None):
    """Uniform disk with approximate circular velocities (reference galaxy.py:95-124)."""
    device = _resolve_device(device)
    gen = _rng(seed)
    G = 0.001
    radii = torch.sqrt(torch.rand(num_stars, generator=gen)) * 10.0 + 0.5
    angles = torch.rand(num_stars, generator=gen) * 2 * math.pi
    cos_a, sin_a = torch.cos(angles), torch.sin(angles)
    positions = torch.stack((radii * cos_a, radii * sin_a), dim=1)
    masses = torch.ones(num_stars)
    v_circ = torch.sqrt(G * num_stars * 0.5 / radii)
    velocities = torch.stack((-v_circ * sin_a, v_circ * cos_a), dim=1)
    return positions.to(device), velocities.to(device), masses.to(device)


def nfw_enclosed_mass(r: torch.Tensor, M_total: float, r_s: float) -> torch.Tensor:
    """Analytic NFW enclosed mass normalised at 10 r_s (reference galaxy.py:127-139)."""
    x = r / r_s
    f_x = torch.log(1 + x) - x / (1 + x)
    f_norm = math.log(1 + 10) - 10 / 11
    return M_total * f_x / f_norm


def create_galaxy_with_halo(num_stars: int = 5000, galaxy_radius: float = 10.0, halo_radius: float = 30.0,
                            dm_mass_ratio: float = 5.0, device: torch.device = None, seed: int = None):
    """Disk galaxy whose velocities include an analytic NFW halo (reference galaxy.py:142-211).

    The halo only shapes the initial velocities; it never enters the force loop.
    """
    device = _resolve_device(device)
    gen = _rng(seed)
    pos, vel, mass = _disk(num_stars, galaxy_radius, 0.3, gen)
    G = 0.001
    visible = mass.sum().item()
    dm_total = visible * dm_mass_ratio
    radii = torch.sqrt((pos ** 2).sum(dim=-1))
    angles = torch.atan2(pos[:, 1], pos[:, 0])

    order = torch.argsort(radii)
    cumulative = torch.cumsum(mass[order], dim=0)
    enclosed_visible = cumulative[torch.argsort(order)]
    enclosed_total = enclosed_visible + nfw_enclosed_mass(radii, dm_total, halo_radius)

    v_circ = torch.sqrt(G * enclosed_total / radii.clamp(min=0.1))
    vel = torch.stack((-v_circ * torch.sin(angles), v_circ * torch.cos(angles)), dim=1)
    dispersion = 0.05 * v_circ.mean()
    vel = vel + torch.randn(num_stars, 2, generator=gen) * dispersion
    return pos.to(device), vel.to(device), mass.to(device)
