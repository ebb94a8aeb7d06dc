"""CPU restatement of the reference's galaxy diagnostics (metrics.py:25-156) in numpy.

TEST INFRASTRUCTURE ONLY (see nbody_oracle.c header): the checker for the native `nb_metrics` kernels; pinned
against the reference's own outputs in tests/golden/g6_galaxy_metrics.npz and g12_metric_flow.npz
(tests/test_oracle_golden.py).  Arithmetic is done in the dtype of the inputs, op by op like the torch code.
"""
import numpy as np


def radii(positions):
    return np.sqrt((positions ** 2).sum(axis=-1, dtype=positions.dtype))               # metrics.py:52


def linspace_f32(max_radius, num_bins):
    """torch.linspace(0, max_radius, num_bins + 1) (float32, ATen's symmetric formulation)."""
    steps = num_bins + 1
    end = np.float32(max_radius)
    step = end / np.float32(steps - 1)
    i = np.arange(steps)
    lo = (step * i.astype(np.float32)).astype(np.float32)
    hi = (end - step * (steps - 1 - i).astype(np.float32)).astype(np.float32)
    return np.where(i < steps // 2, lo, hi).astype(np.float32)


def rotation_curve(positions, velocities, num_bins=20, max_radius=None, edges=None):
    """metrics.py:25-78: per-bin masked means, NaN for empty bins."""
    r = radii(positions)
    if max_radius is None:
        max_radius = float(r.max()) if not np.isnan(r).any() else float("nan")
    dt = positions.dtype
    vt = np.abs(positions[:, 0] * velocities[:, 1] - positions[:, 1] * velocities[:, 0]) / np.maximum(r, dt.type(0.1))
    vt = np.where(np.isnan(r), np.nan, vt).astype(dt)
    e = linspace_f32(max_radius, num_bins) if edges is None else np.asarray(edges, np.float32)
    means, counts = [], []
    for b in range(num_bins):
        mask = (r >= e[b]) & (r < e[b + 1])                                             # :65
        counts.append(int(mask.sum()))
        with np.errstate(invalid="ignore"):
            means.append(float(vt[mask].astype(np.float64).mean().astype(dt)) if mask.any() else float("nan"))
    centres = ((e[:-1] + e[1:]) / np.float32(2)).astype(np.float32)
    return {"radii": centres, "velocities": np.array(means), "num_stars_per_bin": counts, "edges": e}


def galaxy_radius(positions, percentile=90):
    """metrics.py:81-95"""
    r = radii(positions)
    idx = int(len(r) * percentile / 100)
    return float(np.sort(r)[min(idx, len(r) - 1)])


def bound_fraction(positions, velocities, masses, G=0.001):
    """metrics.py:98-145 (ties in r_com broken by index)"""
    dt = positions.dtype
    total = masses.sum(dtype=np.float64).astype(dt)
    com = ((positions * masses[:, None]).sum(axis=0, dtype=np.float64).astype(dt) / total).astype(dt)
    rc = np.sqrt(((positions - com) ** 2).sum(axis=-1, dtype=dt))
    order = np.argsort(rc, kind="stable")
    enclosed = np.empty_like(masses)
    enclosed[order] = np.cumsum(masses[order].astype(np.float64)).astype(dt)
    vesc = np.sqrt(dt.type(2 * G) * enclosed / np.maximum(rc, dt.type(0.1)))
    vmag = np.sqrt((velocities ** 2).sum(axis=-1, dtype=dt))
    return float(np.float32((vmag < vesc).sum()) / np.float32(len(masses)))


def velocity_dispersion(velocities):
    """metrics.py:148-156 (unbiased)"""
    vmag = np.sqrt((velocities ** 2).sum(axis=-1, dtype=velocities.dtype))
    return float(vmag.astype(np.float64).std(ddof=1).astype(velocities.dtype))
