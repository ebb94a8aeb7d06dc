#!/usr/bin/env python3
"""Single-GPU stand-in for the multi-GPU scaling curve.

One GPU computes exactly rank r's share of P ranks (NBODY_SHARD_TIMING=r/P) with a 1-rank RCCL communicator,
so everything of a multi-GPU step except the xGMI latency of the all-reduce is measured: the rank's force
launch(es), prefix reductions, the RCCL calls, kicks / drift / repack.  Sweeps the planner's knobs.

    python tools/shard_timing.py [--n 65536] [--mode float64] [--steps 400]
"""
import argparse
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--mode", default="float64")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--chunks", default="1", help="(ignored: the round-2 chunk pipeline was removed; kept so old command lines still parse)")
    ap.add_argument("--tail", default="4,8")
    ap.add_argument("--which", default="0", help="rank(s) whose share is timed: '0' or 'all'")
    args = ap.parse_args()
    import torch
    import nbody_cosmological_simulation_amd as nb
    from nbody_cosmological_simulation_amd import galaxy, runtime

    os.environ["NBODY_FORCE_COMM"] = "1"
    runtime.init_distributed(device=0)
    dev = torch.device("cuda", 0)
    pos, vel, mass = galaxy.create_disk_galaxy(args.n, seed=42, device="cpu")
    mode = nb.get_mode_from_string(args.mode)
    rows = []
    for P in [int(v) for v in args.ranks.split(",")]:
        ranks = range(P) if args.which == "all" else [0]
        for chunks, tail in itertools.product(args.chunks.split(","), args.tail.split(",")):
            if P == 1 and (chunks != "1"):
                continue
            worst = 0.0
            for r in ranks:
                os.environ["NBODY_SHARD_TIMING"] = f"{r}/{P}"
                os.environ["NB_SYM_TAIL"] = tail
                sim = nb.GalaxySimulation(pos.to(dev), vel.to(dev), mass.to(dev), precision_mode=mode, device=dev)
                sim.run(60)
                sim.synchronize()
                t0 = time.perf_counter()
                sim.run(args.steps)
                sim.synchronize()
                ms = (time.perf_counter() - t0) / args.steps * 1e3
                worst = max(worst, ms)
                sim.close()
            row = {"n": args.n, "mode": args.mode, "P": P, "chunks": int(chunks), "tail_pieces": int(tail),
                   "ms_per_step_slowest_rank": round(worst, 5)}
            rows.append(row)
            print(json.dumps(row), flush=True)
    base = min(r["ms_per_step_slowest_rank"] for r in rows if r["P"] == 1) if any(r["P"] == 1 for r in rows) else None
    if base:
        for r in rows:
            print(f"P={r['P']} chunks={r['chunks']} tail={r['tail_pieces']}: {r['ms_per_step_slowest_rank']:.4f} ms/step "
                  f"-> x{base / r['ms_per_step_slowest_rank']:.2f} before xGMI latency")
    runtime.shutdown()


if __name__ == "__main__":
    main()
