"""Per-snapshot diagnostics -- same surface as the reference's metrics.py.

Reference: metrics.py:12-227 (formulas restated in SURVEY.md Appendix C).  These are O(N log N)
read-only consumers of `sim.positions / velocities / masses` evaluated a few times per run
(every 100 ticks in main.py:161-167); they are NOT part of the accelerated path and run as
plain host-side tensor code on whatever device the state tensors live on.  The O(N^2) energies
they call (`get_kinetic_energy`, `get_potential_energy`) are the native kernels.
"""
from dataclasses import dataclass, field

import numpy as np
import torch


@dataclass
class SimulationMetrics:
    """Container of metric time series (reference metrics.py:12-22)."""
    ticks: list = field(default_factory=list)
    total_energy: list = field(default_factory=list)
    kinetic_energy: list = field(default_factory=list)
    potential_energy: list = field(default_factory=list)
    galaxy_radius_90: list = field(default_factory=list)
    bound_fraction: list = field(default_factory=list)
    velocity_dispersion: list = field(default_factory=list)
    rotation_curves: list = field(default_factory=list)


def _radii(positions):
    return torch.sqrt((positions ** 2).sum(dim=-1))


def compute_rotation_curve(positions: torch.Tensor, velocities: torch.Tensor, num_bins: int = 20,
                           max_radius: float = None) -> dict:
    """Mean tangential speed in `num_bins` radial bins (reference metrics.py:25-78).

    Bin i holds edge_i <= r < edge_{i+1}; empty bins give NaN; columns 0,1 define the plane."""
    radii = _radii(positions)
    if max_radius is None:
        max_radius = radii.max().item()
    v_tan = torch.abs(positions[:, 0] * velocities[:, 1] - positions[:, 1] * velocities[:, 0]) / radii.clamp(min=0.1)
    edges = torch.linspace(0, max_radius, num_bins + 1, device=positions.device)
    centres = (edges[:-1] + edges[1:]) / 2
    # membership of bin i is edge_i <= r < edge_{i+1} exactly as upstream's masks (r == max_radius falls out of
    # the last bin); all bins in one pass and one device-to-host transfer instead of two per bin.  Sums go
    # through a one-hot matrix product: deterministic (no atomics), accumulated in fp64.
    member = (radii.unsqueeze(1) >= edges[:-1].unsqueeze(0)) & (radii.unsqueeze(1) < edges[1:].unsqueeze(0))
    onehot = member.to(torch.float64)
    counts_t = onehot.sum(dim=0)
    sums = onehot.t() @ v_tan.to(torch.float64)
    means_t = (sums / counts_t).to(v_tan.dtype)           # 0 / 0 = NaN for empty bins, like upstream
    stacked = torch.stack([means_t.to(torch.float64), counts_t]).cpu().numpy()
    return {"radii": centres.cpu().numpy(), "velocities": stacked[0].astype(np.float64),
            "num_stars_per_bin": [int(c) for c in stacked[1]]}


def compute_galaxy_radius(positions: torch.Tensor, percentile: float = 90) -> float:
    """Radius containing `percentile` % of the stars (reference metrics.py:81-95)."""
    radii = _radii(positions)
    idx = int(len(radii) * percentile / 100)
    return torch.sort(radii)[0][min(idx, len(radii) - 1)].item()


def compute_bound_fraction(positions: torch.Tensor, velocities: torch.Tensor, masses: torch.Tensor,
                           G: float = 0.001) -> float:
    """Fraction of stars slower than the local escape speed (reference metrics.py:98-145)."""
    total_mass = masses.sum()
    com = (positions * masses.unsqueeze(-1)).sum(dim=0) / total_mass
    r_com = torch.sqrt(((positions - com) ** 2).sum(dim=-1))
    order = torch.argsort(r_com)
    enclosed = torch.cumsum(masses[order], dim=0)[torch.argsort(order)]
    v_esc = torch.sqrt(2 * G * enclosed / r_com.clamp(min=0.1))
    v_mag = torch.sqrt((velocities ** 2).sum(dim=-1))
    return (v_mag < v_esc).float().mean().item()


def compute_velocity_dispersion(velocities: torch.Tensor) -> float:
    """Unbiased standard deviation of |v| (reference metrics.py:148-156)."""
    return torch.sqrt((velocities ** 2).sum(dim=-1)).std().item()


def collect_metrics(simulation, tick: int, metrics: SimulationMetrics):
    """Append every diagnostic for the current state (reference metrics.py:159-179)."""
    pos, vel, masses = simulation.positions, simulation.velocities, simulation.masses
    metrics.ticks.append(tick)
    metrics.kinetic_energy.append(simulation.get_kinetic_energy())
    metrics.potential_energy.append(simulation.get_potential_energy())
    metrics.total_energy.append(simulation.get_total_energy())
    metrics.galaxy_radius_90.append(compute_galaxy_radius(pos, 90))
    metrics.bound_fraction.append(compute_bound_fraction(pos, vel, masses, simulation.G))
    metrics.velocity_dispersion.append(compute_velocity_dispersion(vel))
    metrics.rotation_curves.append(compute_rotation_curve(pos, vel))


def compare_rotation_curves(curve1: dict, curve2: dict, label1: str = "Baseline", label2: str = "Quantized") -> dict:
    """Difference statistics of two rotation curves (reference metrics.py:182-227)."""
    v1, v2 = np.array(curve1["velocities"]), np.array(curve2["velocities"])
    valid = ~(np.isnan(v1) | np.isnan(v2))
    if valid.sum() == 0:
        return {"error": "No valid comparison points"}
    v1v, v2v, rv = v1[valid], v2[valid], curve1["radii"][valid]
    outer = rv > np.median(rv)
    if outer.sum() > 2:
        slope1 = np.polyfit(rv[outer], v1v[outer], 1)[0]
        slope2 = np.polyfit(rv[outer], v2v[outer], 1)[0]
    else:
        slope1 = slope2 = 0
    return {
        "mean_velocity_diff": (v2v - v1v).mean(),
        "outer_slope_baseline": slope1,
        "outer_slope_quantized": slope2,
        "flatness_increase": slope2 - slope1,
        "num_valid_bins": valid.sum(),
    }


def summarize(metrics_dict: dict) -> dict:
    """Numbers behind the reference's text summary (visualization.py:281-313): energy drift %,
    radius change %, final radius, final bound fraction, dispersion change % per mode."""
    out = {}
    for mode, m in metrics_dict.items():
        row = {}
        if m.total_energy:
            e0, e1 = m.total_energy[0], m.total_energy[-1]
            row["energy_drift_pct"] = (e1 - e0) / abs(e0) * 100 if abs(e0) > 1e-10 else 0
        if m.galaxy_radius_90:
            r0, r1 = m.galaxy_radius_90[0], m.galaxy_radius_90[-1]
            row["radius_change_pct"] = (r1 - r0) / r0 * 100 if r0 > 0 else 0
            row["final_radius"] = r1
        if m.bound_fraction:
            row["final_bound_fraction"] = m.bound_fraction[-1]
        if m.velocity_dispersion:
            d0, d1 = m.velocity_dispersion[0], m.velocity_dispersion[-1]
            row["dispersion_change_pct"] = (d1 - d0) / d0 * 100 if d0 > 0 else 0
        out[mode] = row
    return out
