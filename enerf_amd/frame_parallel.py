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


def rank_bindings(rank: int, world: int, local_rank: int, device: torch.device, group=None) -> dict:
    """Which device every rank of the job is bound to, and how many ranks the communicator really holds.

    Returns ``{"bindings": [{rank, local_rank, device, visible, id}, ...], "ranks_seen": n, "backend": name}`` on every rank.
    ``hw`` identifies the physical device (UUID / PCI bus id when the runtime reports one), ``id`` is the (visible list, index)
    pair; two ranks on the same device would time-share one GPU and report a meaningless aggregate, so that is an error
    on ALL ranks (raised after the exchange, so nobody is left inside a collective).  ``ranks_seen`` is not
    ``get_world_size()`` restated: it is the SUM over the communicator of a one per rank, reduced on the device the
    rank renders on — what the driver can hold against ``n_gpus``."""
    import os
    import torch.distributed as dist
    visible = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")))
    if device.type == "cuda":
        p = torch.cuda.get_device_properties(device)
        hw = None
        for attr in ("uuid", "pci_bus_id"):
            v = getattr(p, attr, None)
            if v is not None and str(v) not in ("", "None"):
                hw = f"{attr}:{v}" + (f"/{getattr(p, 'pci_device_id', '')}" if attr == "pci_bus_id" else "")
                break
        mine = {"rank": rank, "local_rank": local_rank, "device": f"cuda:{device.index}", "visible": visible,
                "id": f"visible[{visible}]#{device.index}", "hw": hw, "name": p.name}
    else:                                               # CPU lane-emulator ranks (launcher tests): one "device" per process
        mine = {"rank": rank, "local_rank": local_rank, "device": "cpu", "visible": visible, "id": f"cpu-rank{rank}", "hw": None, "name": "cpu"}
    use_dist = world > 1 and dist.is_available() and dist.is_initialized()
    if not use_dist:
        return {"bindings": [mine], "ranks_seen": 1, "backend": None}
    allb = [None] * world
    dist.all_gather_object(allb, mine, group=group)
    one = torch.ones(1, dtype=torch.float64, device=device if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(one, op=dist.ReduceOp.SUM, group=group)
    hws = [b["hw"] for b in allb]
    # the runtime's hardware ids decide when they are usable (all present and not one bogus constant); otherwise the
    # (visible list, index) pair: never a false refusal on a correctly launched job
    ids = hws if all(h is not None for h in hws) and len(set(hws)) > 1 else [b["id"] for b in allb]
    dup = sorted({i for i in ids if ids.count(i) > 1})
    if dup:
        raise RuntimeError(f"ranks share a device: {[(b['rank'], b['id'], b['hw']) for b, i in zip(allb, ids) if i in dup]} — one process per GPU")
    return {"bindings": allb, "ranks_seen": int(round(float(one.item()))), "backend": dist.get_backend(group)}
