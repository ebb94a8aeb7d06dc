#!/usr/bin/env python3
"""Headline benchmark: particle-steps/s of the direct-sum leapfrog at N = 65 536, fp64.

    python bench.py --gpus 1 --steps 2000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one GalaxySimulation.step() (kick-drift-force-kick) over the whole galaxy:
BASELINE.json config 2 -- N=65 536 synthetic disk galaxy, fp32 initial conditions, FLOAT64
mode, G=1e-3, softening=0.1, dt=0.01.  With N GPUs the SAME galaxy is stepped with the
source loop block-partitioned over the ranks and one RCCL all-reduce of the force vectors
per step (strong scaling).  The state is resident in HBM before the timed region starts;
the K timed steps are ONE native call (nb_step) bracketed by barrier + device sync.

Rank 0 prints one JSON line (contract in the task description) with two extra objects:
  roofline      dominant kernel (force_f64_kernel) algorithmic fp64 flop (14 per ordered pair,
                SURVEY.md section 8d) / its HIP-event duration measured inside this run, against
                the fp64 vector peak of MI355X.
  cpu_baseline  the CPU oracle (oracle/, "port" of the reference's algorithm, OpenMP over all
                host cores) timed on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X fp64 vector (= fp64 matrix) peak, AMD spec; SURVEY.md 8d
FP32_VECTOR_PEAK_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md
FLOP_PER_PAIR_2D = 14              # 5*D + 4, SURVEY.md section 8d
SPINUP_EVALS = 40                  # untimed force evaluations (incl. warm-up steps) before the timed region


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000, help="timed steps (default: the 2 000 ticks of BASELINE config 2)")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--n", "--particles", dest="n", type=int, default=65536,
                    help="particles (default: BASELINE config 2); use --particles under torch.distributed.run, whose\n"
                         "own parser takes --n for an abbreviation of --nnodes")
    ap.add_argument("--mode", default="float64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample length")
    return ap.parse_args()


def cpu_baseline(pos, vel, mass, target_s):
    """Oracle (C port of the reference algorithm, all host cores) on a bounded sample."""
    import numpy as np
    from oracle import oracle as O
    p = np.ascontiguousarray(pos.double().numpy())
    v = np.ascontiguousarray(vel.double().numpy())
    m = np.ascontiguousarray(mass.double().numpy())
    n, d = p.shape
    lib = O.lib()
    acc = np.empty_like(p)
    t0 = time.perf_counter()
    lib.nbo_accelerations_f64_fast(n, d, O._dp(p), O._dp(m), 0.001, 0.1 ** 2, 0, n, O._dp(acc))
    t_first = time.perf_counter() - t0
    steps = max(1, min(50, int(target_s / max(t_first, 1e-3))))
    t0 = time.perf_counter()
    lib.nbo_step_f64_fast(n, d, O._dp(p), O._dp(v), O._dp(m), O._dp(acc), 0.001, 0.1 ** 2, 0.01, steps)
    dt = time.perf_counter() - t0
    # the reference's own (materialised N x N x D torch) formulation cannot reach N = 65536; time it at
    # the largest comfortable size on the same cores for context (SURVEY.md section 8d CPU baseline (ii))
    ref_form = None
    try:
        import torch
        from oracle import torch_materialised as TM
        nr = 4096
        best = None
        for threads in (None, 16):          # torch's default thread count (what a user gets), and 16
            rt, used = TM.time_steps(pos[:nr].clone(), vel[:nr].clone(), mass[:nr].clone(), steps=3, warmup=1,
                                     threads=threads)
            row = {"value": nr * 3 / rt, "threads": used, "pair_interactions_per_s": float(nr) * nr * 3 / rt}
            if best is None or row["value"] > best["value"]:
                best = row
        ref_form = dict(best, unit="particle-steps/s", n=nr,
                        what="reference formulation (materialised torch broadcasts, simulation.py:74-143) restated "
                             "in oracle/torch_materialised.py, FLOAT64 mode, 3 steps, best of torch-default / 16 threads")
    except Exception as exc:            # never let the context measurement break the bench line
        ref_form = {"error": repr(exc)}
    return {
        "reference_formulation_torch_cpu": ref_form,
        "value": n * steps / dt, "unit": "particle-steps/s", "cores": O.num_threads(), "kind": "port",
        "sample": f"N={n} fp64 disk galaxy, {steps} leapfrog steps of the oracle's OpenMP fast path "
                  f"({dt:.1f} s; the reference's own PyTorch formulation cannot run at this N)",
        "pair_interactions_per_s": float(n) * n * steps / dt,
    }


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC summary (collected
    in separate --pmc passes as the MI355X guide prescribes), or None when not profiled yet."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        summary = json.load(open(path))
    except OSError:
        return None
    for name, counters in summary.items():
        if name.startswith(kernel_name) and "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
            return (2.0 * counters["FETCH_SIZE"]["avg"] + counters["WRITE_SIZE"]["avg"]) * 1024.0
    return None


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    launched = "RANK" in os.environ          # under torch.distributed.run, also with one rank
    if launched:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import nbody_cosmological_simulation_amd as nb
    from nbody_cosmological_simulation_amd import galaxy, runtime
    runtime.init_distributed(device=local_rank)

    n = args.n
    mode = nb.get_mode_from_string(args.mode)
    pos, vel, mass = galaxy.create_disk_galaxy(n, seed=42, device="cpu")       # fp32, like main.py:131-133
    dev = torch.device("cuda", local_rank)
    # HIP events on the force kernel's dispatch cost ~4 us per step (the launch can no longer overlap its
    # neighbours): measured at N=1, where the roofline is judged; multi-GPU runs bound the kernel by the step
    profile = os.environ.get("NB_BENCH_NOPROFILE") != "1" and (world == 1 or os.environ.get("NB_BENCH_PROFILE") == "1")
    sim = nb.GalaxySimulation(pos.to(dev), vel.to(dev), mass.to(dev), precision_mode=mode,
                              G=0.001, softening=0.1, dt=0.01, device=dev,
                              profile=profile)

    def barrier():
        if launched:
            dist.barrier()
        torch.cuda.synchronize()
        sim.synchronize()

    sim.run(args.warmup)
    # the chip needs ~30 launches (~40 ms) to settle at its sustained clock (profiles/r01_v8_clock_ramp.txt):
    # short warm-ups are topped up with force evaluations that leave the state untouched (untimed)
    spinup = max(0, SPINUP_EVALS - args.warmup) if n >= 16384 else 0
    sim.spin_up(spinup)
    e0 = sim.get_total_energy()
    sim.kernel_time()                 # reset the event accumulators
    barrier()
    t0 = time.perf_counter()
    sim.run(args.steps)               # one native call: K steps on the device
    sim.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms, launches = sim.kernel_time()
    kernel_name = sim.force_kernel_name()
    e1 = sim.get_total_energy()

    if rank == 0:
        is64 = mode == nb.PrecisionMode.FLOAT64
        peak = FP64_VECTOR_PEAK_TFLOPS if is64 else FP32_VECTOR_PEAK_TFLOPS
        pairs_per_launch = float(n) * n / world                 # this rank's source block
        avg_ms = kern_ms / max(launches, 1)
        timing = "hip events on the kernel's dispatch (hipExtLaunchKernelGGL), on the engine's own stream"
        if launches == 0:                 # no events (multi-GPU / NB_BENCH_NOPROFILE=1): bound the kernel by the step
            avg_ms = elapsed / args.steps * 1e3
            timing = "events off: kernel time bounded above by the whole step (includes O(N) kernels and the all-reduce)"
        achieved = FLOP_PER_PAIR_2D * pairs_per_launch / (avg_ms * 1e-3) / 1e12
        out = {
            "metric": f"particle-steps/sec (N={n} {'fp64' if is64 else args.mode} direct-sum leapfrog)",
            "value": n * args.steps / elapsed,
            "unit": "particle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_force_evals": spinup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if is64 else "f32",
            "data": "synthetic",
            "config": {"workload": f"N={n} exponential-disk galaxy (seed 42), fp32 ICs, {mode.value} mode, "
                                   f"G=1e-3 eps=0.1 dt=0.01, KDK leapfrog, all-pairs direct sum",
                       "parallelism": f"j-block x{world} + RCCL all-reduce" if world > 1 else "single GPU"},
            "pair_interactions_per_s": float(n) * n * args.steps / elapsed,
            "energy_drift_rel": (e1 - e0) / abs(e0),
            "roofline": {
                "bound": "mfma",
                "bound_note": "compute-bound: priced against the dense fp64 (fp32 modes: fp32) MFMA peak, which on MI355X "
                              "equals the vector-ALU peak the kernel actually issues on (the pair loop is VALU code: "
                              "contraction dimension D = 2, DESIGN.md section 3); HBM traffic is O(N) per step",
                "kernel": kernel_name,
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "avg_launch_ms": avg_ms, "launches": launches, "timing": timing,
                "flop_per_launch": FLOP_PER_PAIR_2D * pairs_per_launch,
                "traffic": pmc_traffic(kernel_name),
                "traffic_note": "HBM bytes per launch from rocprofv3 PMC passes committed under profiles/ "
                                "(2*FETCH_SIZE + WRITE_SIZE, gfx950 correction); dominated by the slab writes "
                                "that make the sums order-deterministic, DESIGN.md section 2",
            },
        }
        if world == 1 and not args.no_cpu_baseline and is64:
            out["cpu_baseline"] = cpu_baseline(pos, vel, mass, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    sim.close()                       # communicator down on every rank before the process group goes
    if launched:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
