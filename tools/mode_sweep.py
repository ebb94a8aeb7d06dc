#!/usr/bin/env python3
"""ms per force launch of every precision mode at one size (HIP events on the kernel's dispatch), with the grid
modes' table-free-path diagnostics.    python tools/mode_sweep.py [--n 65536] [--dim 2] [--steps 300]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--modes", default="float64,float32,bfloat16,float16,int8,int4,custom")
    ap.add_argument("--softening", type=float, default=0.1)
    ap.add_argument("--unequal", action="store_true", help="unequal masses (general-mass kernels)")
    args = ap.parse_args()
    import torch
    import nbody_cosmological_simulation_amd as nb
    from nbody_cosmological_simulation_amd import galaxy
    dev = torch.device("cuda", 0)
    pos, vel, mass = galaxy.create_disk_galaxy(args.n, seed=42, device="cpu")
    if args.dim == 3:
        g = torch.Generator().manual_seed(1)
        pos = torch.cat([pos, 0.3 * torch.randn(args.n, 1, generator=g)], 1)
        vel = torch.cat([vel, torch.zeros(args.n, 1)], 1)
    if args.unequal:
        g = torch.Generator().manual_seed(2)
        mass = mass * (0.5 + torch.rand(args.n, generator=g))
    flop = (5 * args.dim + 4) * float(args.n) ** 2
    for name in args.modes.split(","):
        mode = nb.get_mode_from_string(name)
        sim = nb.GalaxySimulation(pos.to(dev), vel.to(dev), mass.to(dev), precision_mode=mode, softening=args.softening,
                                  device=dev, profile=True)
        sim.run(40)
        sim.kernel_time()
        sim.run(args.steps)
        ms, launches = sim.kernel_time()
        row = {"mode": mode.value, "n": args.n, "dim": args.dim, "kernel": sim.force_kernel_name(),
               "ms_per_launch": round(ms / max(launches, 1), 5), "launches": launches}
        peak = 78.6 if mode.value == "float64" else 157.3
        row["tflops"] = round(flop / (row["ms_per_launch"] * 1e-3) / 1e12, 2)
        row["frac_of_peak"] = round(row["tflops"] / peak, 4)
        if mode.value in ("int8_sim", "int4_sim", "custom"):
            d = sim.quant_debug()
            row.update(fast_path=d["fast_path"], fast_maxdev=d["fast_maxdev"], fast_maxrel=d["fast_maxrel"])
        print(json.dumps(row), flush=True)
        sim.close()


if __name__ == "__main__":
    main()
