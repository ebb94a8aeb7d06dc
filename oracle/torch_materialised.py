"""The reference's OWN formulation of the step, restated with torch tensor ops on the CPU.

TEST / BASELINE INFRASTRUCTURE ONLY (see nbody_oracle.c header): used by bench.py to time "the
reference's PyTorch-CPU path" on the GPU box's host cores -- the reference's source cannot travel
there -- and by tests/test_oracle_golden.py as a second, independent check of the C oracle.

Restates simulation.py:74-143 (fully materialised N x N x D broadcast, `** 1.5`, `G / .`,
`* masses`, `* (1 - eye)`, `.sum(dim=1)`, KDK with rebinding) and quantization.py:43-56 for the
cast modes.  Memory is O(N^2 D): usable up to N ~ 8192 in fp64, exactly like upstream.
"""
import time

import torch


def accelerations(pos, masses, G=0.001, softening_sq=0.1 ** 2, mode="float64"):
    """simulation.py:83-112 with the cast hooks of quantization.py:43-56."""
    n = pos.shape[0]
    sep = pos[None, :, :] - pos[:, None, :]                    # :83   sep[i, j] = x_j - x_i
    r2 = sep.pow(2).sum(-1) + softening_sq                     # :86
    if mode == "float64":                                      # quantization.py:43-45
        r2 = r2.double()
    elif mode == "float32":
        r2 = r2.float()
    elif mode == "bfloat16":
        r2 = r2.bfloat16().float()
    elif mode == "float16":
        r2 = r2.half().float()
    else:
        raise ValueError("grid modes are covered by the C oracle")
    w = G / r2.pow(1.5)                                        # :97-101
    w = w * masses[None, :]                                    # :105
    w = w * (1 - torch.eye(n, device=pos.device))              # :108
    return (w[:, :, None] * sep).sum(1)                        # :112


def step(state, G=0.001, softening_sq=0.1 ** 2, dt=0.01, mode="float64"):
    """simulation.py:132-141 on a dict(pos, vel, masses, acc); tensors are rebound, not mutated."""
    state["vel"] = state["vel"] + state["acc"] * (dt / 2)
    state["pos"] = state["pos"] + state["vel"] * dt
    state["acc"] = accelerations(state["pos"], state["masses"], G, softening_sq, mode)
    state["vel"] = state["vel"] + state["acc"] * (dt / 2)
    return state


def time_steps(pos, vel, masses, steps, warmup=2, mode="float64", threads=None):
    """perf_counter timing idiom of omega_point_test.py:305-319.  Returns (seconds, threads)."""
    if threads:
        torch.set_num_threads(threads)
    st = dict(pos=pos.clone(), vel=vel.clone(), masses=masses.clone())
    st["acc"] = accelerations(st["pos"], st["masses"], mode=mode)
    for _ in range(warmup):
        step(st, mode=mode)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(st, mode=mode)
    return time.perf_counter() - t0, torch.get_num_threads()
