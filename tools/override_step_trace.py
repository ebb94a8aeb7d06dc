#!/usr/bin/env python3
"""A sensitivity_test.py-style subclass (tensor-level _grid_quantize_safe inside an overridden
_compute_accelerations, torch ops around it) stepped at N = 4096: the workload of the reference's 18 override
subclasses.  Run under `rocprofv3 --hip-trace --stats` to count hipMalloc / hipFree / synchronisations per step
(the hooks must not allocate or block per call)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                  # noqa: E402
import nbody_cosmological_simulation_amd as nb                # noqa: E402
from nbody_cosmological_simulation_amd import galaxy           # noqa: E402
from nbody_cosmological_simulation_amd.quantization import _grid_quantize_safe   # noqa: E402


class QuantSim(nb.GalaxySimulation):
    def __init__(self, *a, quant_levels=64, **kw):
        self.quant_levels = quant_levels
        super().__init__(*a, **kw)

    def _compute_accelerations(self):
        pos = self.positions
        diff = pos.unsqueeze(0) - pos.unsqueeze(1)
        dist_sq = (diff ** 2).sum(dim=-1) + self.softening_sq
        dist_sq = _grid_quantize_safe(dist_sq, self.quant_levels, min_val=0.01)
        ff = self.G / dist_sq ** 1.5
        ff = ff * self.masses.unsqueeze(0)
        ff = ff * (1 - torch.eye(self.num_stars, device=pos.device))
        return (ff.unsqueeze(-1) * diff).sum(dim=1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    pos, vel, mass = galaxy.create_disk_galaxy(n, seed=3, device="cpu")
    sim = QuantSim(pos.cuda(), vel.cuda(), mass.cuda(), precision_mode=nb.PrecisionMode.FLOAT32, quant_levels=64)
    for _ in range(5):
        sim.step()
    torch.cuda.synchronize()
    print("WARMUP-DONE", flush=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step()
    torch.cuda.synchronize()
    print(f"N={n}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per override step ({steps} steps)")


if __name__ == "__main__":
    main()
