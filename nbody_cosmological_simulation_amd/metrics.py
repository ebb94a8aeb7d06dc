"""Per-snapshot diagnostics -- same surface as the reference's metrics.py.

Reference: metrics.py:12-227 (formulas restated in SURVEY.md Appendix C).  The four diagnostics run as HIP kernels
behind the C-ABI (`nb_metrics` on a simulation's device-resident state, `nb_metrics_tensors` on caller tensors;
csrc/nb_metrics.hip): radii, tangential speeds, the rotation-curve bins, the r-percentile order statistic, the
enclosed masses behind the escape test and the dispersion never leave the device.  The two rankings take the
reference's own route -- sort, then prefix sum: a stable device radix sort of the radii (rocPRIM, csrc/nb_sort.hip) and
a fixed-order fp64 scan of the masses in that order.  The only host-side arithmetic kept is what the reference also does
on the host: the 21 float32 bin edges come from `torch.linspace` itself, so bin membership is decided against
bit-identical edges.

Deviations from the reference, stated: (1) no torch fallback -- CPU tensors are staged through the GPU, and without a HIP
device (or the library) these functions raise; (2) float16 / bfloat16 tensors are evaluated in float32 (the reference
evaluates them in their own dtype; no script of the reference passes such tensors here, the engine-state path
`collect_metrics` is unaffected); (3) the single-scalar helpers compute_galaxy_radius / compute_bound_fraction /
compute_velocity_dispersion run the whole pipeline (both sorts, the scan) and return one of its results -- use
`native_metrics` / `collect_metrics` to get all of them from one evaluation.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _native as N
from .quantization import _TORCH_TO_NB, _hip_device_for, hook_stream


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


def _edges(max_radius: float, num_bins: int):
    """The reference's bin edges: torch.linspace(0, max_radius, num_bins + 1) (float32), as a host array."""
    return np.ascontiguousarray(torch.linspace(0, max_radius, num_bins + 1).numpy(), dtype=np.float32)


def _prep(t: torch.Tensor, dtype):
    t = t.detach()
    if t.dtype != dtype:
        t = t.to(dtype)                 # mixed / half-typed tensors: evaluated in float32 (fp64 if any is fp64)
    return t.contiguous()


def native_metrics(positions, velocities, masses=None, G: float = 0.001, num_bins: int = 20,
                   max_radius: float = None, percentile: float = 90, simulation=None) -> dict:
    """All four diagnostics in one native evaluation.  `simulation`: use that engine's device-resident state
    (`nb_metrics`) instead of the tensors (`nb_metrics_tensors`)."""
    L = N.lib()
    mean = (C.c_double * max(num_bins, 1))()
    count = (C.c_int64 * max(num_bins, 1))()
    sc = (C.c_double * 5)()

    def call(edges, mr, radius_only):
        ep = None if edges is None else edges.ctypes.data_as(C.c_void_p)
        if simulation is not None:
            N.check(L.nb_metrics(simulation._handle, num_bins, ep, mr, float(percentile), int(radius_only), mean, count, sc))
            return
        n, d = pos.shape
        with hook_stream(pos):
            N.check(L.nb_metrics_tensors(_hip_device_for(pos), C.c_void_p(pos.data_ptr()), C.c_void_p(vel.data_ptr()),
                                         C.c_void_p(mas.data_ptr()), n, d, _TORCH_TO_NB[pos.dtype],
                                         int(pos.device.type == "cuda"), float(G), num_bins, ep, mr, float(percentile),
                                         int(radius_only), mean, count, sc))

    if simulation is None:
        dt = torch.float64 if torch.float64 in (positions.dtype, velocities.dtype) else torch.float32
        pos, vel = _prep(positions, dt), _prep(velocities, dt).to(positions.device)
        mas = _prep(masses if masses is not None else torch.ones(pos.shape[0], dtype=dt, device=pos.device), dt).to(pos.device)
        if pos.shape[0] == 0:
            raise RuntimeError("metrics of an empty galaxy are undefined (max() of an empty tensor upstream)")
    if max_radius is None:
        call(None, -1.0, True)                    # radii.max() from the device
        max_radius = sc[0]
    edges = _edges(max_radius, num_bins)
    call(edges, float(max_radius), False)
    return {"edges": edges, "mean": np.array(mean[:num_bins], np.float64), "count": [int(c) for c in count[:num_bins]],
            "max_radius": float(max_radius), "r_percentile": sc[1], "bound_fraction": sc[2], "dispersion": sc[3]}


def compute_rotation_curve(positions: torch.Tensor, velocities: torch.Tensor, num_bins: int = 20,
                           max_radius: float = None) -> dict:
    """Mean tangential speed in `num_bins` radial bins (reference metrics.py:25-78).

    Bin i holds edge_i <= r < edge_{i+1} (the farthest star falls out of the last bin); empty bins give NaN; a star
    with a non-finite radius belongs to no bin, a non-finite speed only affects its own bin; columns 0,1 define the
    plane."""
    m = native_metrics(positions, velocities, None, num_bins=num_bins, max_radius=max_radius)
    e = torch.from_numpy(m["edges"])
    return {"radii": ((e[:-1] + e[1:]) / 2).numpy(), "velocities": m["mean"], "num_stars_per_bin": m["count"]}


def compute_galaxy_radius(positions: torch.Tensor, percentile: float = 90) -> float:
    """Radius containing `percentile` % of the stars (reference metrics.py:81-95)."""
    return native_metrics(positions, torch.zeros_like(positions), None, num_bins=0, max_radius=1.0,
                          percentile=percentile)["r_percentile"]


def compute_bound_fraction(positions: torch.Tensor, velocities: torch.Tensor, masses: torch.Tensor,
                           G: float = 0.001) -> float:
    """Fraction of stars slower than the local escape speed (reference metrics.py:98-145)."""
    return native_metrics(positions, velocities, masses, G=G, num_bins=0, max_radius=1.0)["bound_fraction"]


def compute_velocity_dispersion(velocities: torch.Tensor) -> float:
    """Unbiased standard deviation of |v| (reference metrics.py:148-156)."""
    return native_metrics(torch.zeros_like(velocities), velocities, None, num_bins=0, max_radius=1.0)["dispersion"]


def collect_metrics(simulation, tick: int, metrics: SimulationMetrics):
    """Append every diagnostic for the current state (reference metrics.py:159-179): energies and all four
    diagnostics from the engine's device-resident state, one native evaluation, no state download."""
    metrics.ticks.append(tick)
    total = simulation.get_total_energy()          # both parts from one native evaluation; the getters below hit its memo
    metrics.kinetic_energy.append(simulation.get_kinetic_energy())
    metrics.potential_energy.append(simulation.get_potential_energy())
    metrics.total_energy.append(total)
    if hasattr(simulation, "_native_metrics_ready") and simulation._native_metrics_ready():
        m = native_metrics(None, None, None, simulation=simulation)
    else:                       # another object with the same attributes (or an empty galaxy): through the tensors
        m = native_metrics(simulation.positions, simulation.velocities, simulation.masses, G=simulation.G)
    e = torch.from_numpy(m["edges"])
    metrics.galaxy_radius_90.append(m["r_percentile"])
    metrics.bound_fraction.append(m["bound_fraction"])
    metrics.velocity_dispersion.append(m["dispersion"])
    metrics.rotation_curves.append({"radii": ((e[:-1] + e[1:]) / 2).numpy(), "velocities": m["mean"],
                                    "num_stars_per_bin": m["count"]})


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
