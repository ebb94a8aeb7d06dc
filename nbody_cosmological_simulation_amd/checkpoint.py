"""Snapshot / restart and state hashing.

The reference has no real checkpointing: `get_state()` returns clones (simulation.py:160-168) and
`hash_tensor_state` (reproducibility.py:227-232) prints a SHA-256 prefix of the position and
velocity bytes that is never compared.  Here both become usable for the long multi-GPU runs
(BASELINE config 5: 10 000 ticks): a snapshot is a plain `.npz` (numpy, no pickle) holding the
state in its CURRENT logical dtypes plus the constructor arguments; restoring it rebuilds the
simulation and carries on bit-for-bit, because the engine's sums are order-deterministic.
"""
import hashlib

import numpy as np
import torch

from .quantization import PrecisionMode
from .simulation import GalaxySimulation


def state_hash(sim_or_positions, velocities: torch.Tensor = None) -> str:
    """SHA-256[:16] of positions||velocities bytes -- same recipe as reproducibility.py:227-232."""
    if velocities is None:
        positions, velocities = sim_or_positions.positions, sim_or_positions.velocities
    else:
        positions = sim_or_positions
    # bfloat16 has no numpy dtype (upstream's .numpy() raises there): hash its raw 16-bit patterns
    data = _np(positions)[0].tobytes() + _np(velocities)[0].tobytes()
    return hashlib.sha256(data).hexdigest()[:16]


def _np(t: torch.Tensor):
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:            # numpy has no bfloat16: keep the raw 16-bit patterns
        return t.view(torch.int16).numpy(), "bfloat16"
    return t.numpy(), str(t.dtype).replace("torch.", "")


def save_snapshot(sim: GalaxySimulation, path: str) -> str:
    """Write the full restartable state to `path` (.npz); returns the state hash."""
    arrays, dtypes = {}, {}
    for name in ("positions", "velocities", "masses", "accelerations"):
        arrays[name], dtypes[name] = _np(getattr(sim, name))
    h = state_hash(sim)
    np.savez(path, **arrays,
             dtypes=np.array([dtypes[k] for k in ("positions", "velocities", "masses", "accelerations")]),
             precision_mode=np.array(sim.precision_mode.value), G=sim.G, softening=sim.softening, dt=sim.dt,
             tick=sim.tick, custom_levels=-1 if sim.custom_levels is None else sim.custom_levels,
             state_hash=np.array(h))
    return h


def _tensor(arr, dtype_name):
    if dtype_name == "bfloat16":
        return torch.from_numpy(arr.copy()).view(torch.bfloat16)
    return torch.from_numpy(arr.copy())


def load_snapshot(path: str, device=None, cls=GalaxySimulation, **overrides) -> GalaxySimulation:
    """Rebuild a simulation from `save_snapshot` output and continue exactly where it stopped.

    The stored accelerations are restored instead of being recomputed, so a FLOAT64-mode run that
    had already been promoted to fp64 restarts on the fp64 path with the very same force values."""
    z = np.load(path, allow_pickle=False)
    names = [str(s) for s in z["dtypes"]]
    pos, vel, mass, acc = (_tensor(z[k], n) for k, n in zip(("positions", "velocities", "masses", "accelerations"), names))
    levels = int(z["custom_levels"])
    kw = dict(precision_mode=PrecisionMode(str(z["precision_mode"])), G=float(z["G"]), softening=float(z["softening"]),
              dt=float(z["dt"]), device=device, custom_levels=None if levels < 0 else levels)
    kw.update(overrides)
    sim = cls(pos, vel, mass, **kw)
    sim.accelerations = acc.to(sim.device)      # marks the array dirty -> uploaded before the next step
    sim.tick = int(z["tick"])
    return sim
