"""Frame-parallel rendering over the GPUs of one node.

Rendering shards naturally: one frame = one ``batch`` = one target view + its source views, no exchange
step, weights replicated (1.74 MB).  Frame f goes to rank ``f mod G``; the only collectives are the
timing barrier and a MAX/SUM reduction of (elapsed, frames) for reporting — RCCL ("nccl") on GPUs, gloo
in the CPU tests.
"""
from __future__ import annotations

import time
from typing import Callable, List, Optional, Tuple

import torch


def frames_of_rank(n_frames: int, rank: int, world: int) -> List[int]:
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_frames, world))


def render_sharded(render: Callable[[int], dict], n_frames: int, rank: int, world: int,
                   sync: Optional[Callable[[], None]] = None, group=None) -> Tuple[dict, float, float]:
    """Render this rank's frames; return ({frame: output}, aggregate_fps, max_elapsed_s).

    ``render(f)`` renders frame f; ``sync`` waits for the device (torch.cuda.synchronize on GPUs).
    """
    import torch.distributed as dist
    use_dist = world > 1 and dist.is_available() and dist.is_initialized()
    sync = sync or (lambda: None)
    sync()
    if use_dist:
        dist.barrier(group=group)
    t0 = time.perf_counter()
    outs = {f: render(f) for f in frames_of_rank(n_frames, rank, world)}
    sync()
    if use_dist:
        dist.barrier(group=group)
    elapsed = time.perf_counter() - t0
    total = torch.tensor([float(len(outs))], dtype=torch.float64)
    tmax = torch.tensor([elapsed], dtype=torch.float64)
    if use_dist:
        backend = dist.get_backend(group)
        if backend == "nccl":
            total, tmax = total.cuda(), tmax.cuda()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    return outs, float(total.item()) / float(tmax.item()), float(tmax.item())
