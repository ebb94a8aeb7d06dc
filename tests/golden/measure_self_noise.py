#!/usr/bin/env python3
"""Noise floor of particle-level parity for config 1's system (N = 1024, FLOAT64 mode), measured by RUNNING the
reference (build container only, like make_golden.py):

  (a) the reference against itself with the sources summed in reverse order and x*sqrt(x) for the 1.5 power
      (same mathematics, same dtype state machine, different rounding);
  (b) the reference against the CPU oracle (oracle/nbody_oracle.c).

Relative to max|x| / max|v| at ticks 200 / 1000 / 2000.  Output committed as profiles/r02_self_noise_n1024.txt;
tests/test_gpu_parity.py::test_2000_tick_horizon_fp64_vs_oracle quotes it.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
import simulation as ref_sim                      # noqa: E402
from quantization import PrecisionMode            # noqa: E402
from oracle import oracle as O                    # noqa: E402

torch.set_num_threads(8)
g = np.load(os.path.join(ROOT, "tests", "golden", "g2_config1_n1024.npz"))
pos, vel, mass = (torch.from_numpy(g[k]) for k in ("pos", "vel", "mass"))


class Reversed(ref_sim.GalaxySimulation):
    def _compute_accelerations(self):
        p = self.positions
        d = p.unsqueeze(0) - p.unsqueeze(1)
        r2 = ((d ** 2).sum(-1) + self.softening_sq).double()
        w = self.G / (r2 * torch.sqrt(r2))
        w = w * self.masses.unsqueeze(0)
        w = w * (1 - torch.eye(self.num_stars))
        return (w.unsqueeze(-1) * d).flip(1).sum(1)


def main():
    kw = dict(precision_mode=PrecisionMode.FLOAT64, device=torch.device("cpu"))
    a = ref_sim.GalaxySimulation(pos.clone(), vel.clone(), mass.clone(), **kw)
    b = Reversed(pos.clone(), vel.clone(), mass.clone(), **kw)
    ora = O.OracleSim(g["pos"], g["vel"], g["mass"], "float64")
    ora.step()
    p, v, m, acc = (np.ascontiguousarray(x, np.float64).copy() for x in
                    (ora.positions, ora.velocities, ora.masses, ora.accelerations))
    a.step()
    b.step()
    done = 1
    for t in (200, 1000, 2000):
        O.lib().nbo_step_f64_fast(1024, 2, O._dp(p), O._dp(v), O._dp(m), O._dp(acc), 0.001, 0.1 ** 2, 0.01, t - done)
        for _ in range(t - done):
            a.step()
            b.step()
        done = t
        sx, sv = a.positions.abs().max().item(), a.velocities.abs().max().item()
        print(f"tick {t}: reference vs reversed-sum reference: pos {(a.positions - b.positions).abs().max().item() / sx:.3e} "
              f"vel {(a.velocities - b.velocities).abs().max().item() / sv:.3e} | reference vs oracle: "
              f"pos {np.abs(a.positions.numpy() - p).max() / sx:.3e} vel {np.abs(a.velocities.numpy() - v).max() / sv:.3e}",
              flush=True)


if __name__ == "__main__":
    main()
