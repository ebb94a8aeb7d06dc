"""`main.py`-shaped entry point (reference main.py:99-208 call sequence, without the plots).

    python -m nbody_cosmological_simulation_amd.cli --stars 1024 --ticks 200 --compare float64

Builds a disk galaxy, casts it to fp32 (main.py:131-133), and for every requested precision mode
constructs a GalaxySimulation, collects metrics at tick 0 and every 100 ticks through the
run() callback, and prints the summary table.  Plotting (visualization.py) is out of scope.
"""
import argparse
import json

import torch

from .galaxy import create_disk_galaxy
from .metrics import SimulationMetrics, collect_metrics, summarize
from .quantization import describe_mode, get_mode_from_string
from .simulation import GalaxySimulation


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Direct-sum galaxy simulation on MI355X (reference CLI flags)")
    p.add_argument("--stars", "-n", type=int, default=3000)
    p.add_argument("--ticks", "-t", type=int, default=1000)
    p.add_argument("--compare", "-c", type=str, default="float64,int4")
    p.add_argument("--quick", action="store_true")
    p.add_argument("--dt", type=float, default=0.01)
    p.add_argument("--G", type=float, default=0.001)
    p.add_argument("--seed", type=int, default=None, help="seed torch's generator first (the reference is unseeded)")
    p.add_argument("--json", type=str, default=None, help="also write the summary numbers to this file")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if args.quick:
        args.stars, args.ticks = 500, 500
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    modes = [get_mode_from_string(s.strip()) for s in args.compare.split(",")]
    print(f"Device: {device}")
    for m in modes:
        print(f"  - {m.value}: {describe_mode(m)}")
    if args.seed is not None:
        torch.manual_seed(args.seed)
    positions, velocities, masses = create_disk_galaxy(num_stars=args.stars, galaxy_radius=10.0, device=device)
    positions, velocities, masses = positions.float(), velocities.float(), masses.float()

    all_metrics = {}
    for mode in modes:
        sim = GalaxySimulation(positions.clone(), velocities.clone(), masses.clone(), precision_mode=mode,
                               G=args.G, dt=args.dt, device=device)
        metrics = SimulationMetrics()
        collect_metrics(sim, 0, metrics)

        def progress(s, tick, metrics=metrics):
            collect_metrics(s, tick, metrics)
            if tick % 200 == 0:
                print(f"  Tick {tick}: Energy={s.get_total_energy():.4f}")

        sim.run(num_ticks=args.ticks, callback=progress, callback_interval=100)
        all_metrics[mode.value] = metrics

    summary = summarize(all_metrics)
    print("\n" + "=" * 60 + "\nSIMULATION RESULTS SUMMARY\n" + "=" * 60)
    for mode, row in summary.items():
        print(f"\n{mode}:\n" + "-" * 40)
        if "energy_drift_pct" in row:
            print(f"  Energy drift: {row['energy_drift_pct']:+.2f}%")
        if "radius_change_pct" in row:
            print(f"  Radius change: {row['radius_change_pct']:+.2f}%")
            print(f"  Final radius: {row['final_radius']:.2f}")
        if "final_bound_fraction" in row:
            print(f"  Final bound fraction: {row['final_bound_fraction']:.1%}")
        if "dispersion_change_pct" in row:
            print(f"  Velocity dispersion change: {row['dispersion_change_pct']:+.2f}%")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(summary, f, indent=1)
    return all_metrics


if __name__ == "__main__":
    main()
