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


def _cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def pin_rank_to_cores(local_rank: int, local_world: int, device: torch.device) -> dict:
    """Give this rank a core set no other rank of the node uses (the frame loop is host-enqueue sensitive —
    0.15 ms of launches per 0.8 ms frame — and N unpinned ranks migrate over each other's cores and cross NUMA nodes).

    The rank's share is cut from the cores of its GPU's NUMA node when sysfs reports one (the PCI device's ``numa_node`` ->
    ``nodeN/cpulist``, divided among the ranks whose GPUs sit on the same node, in local-rank order), otherwise from an even
    split of the process's current affinity mask.  Returns ``{"cores": "a-b", "n": count, "numa_node": node | None, "how": ...}``;
    never raises (an unpinned rank is reported as such)."""
    import os
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return {"cores": None, "n": 0, "numa_node": None, "how": "sched_getaffinity unavailable"}
    if local_world <= 1 or not allowed:
        return {"cores": None, "n": len(allowed), "numa_node": None, "how": "single rank: not pinned"}
    node, node_cores = None, None
    if device.type == "cuda":
        try:
            p = torch.cuda.get_device_properties(device)
            bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as fh:
                node = int(fh.read().strip())
            if node >= 0:
                with open(f"/sys/devices/system/node/node{node}/cpulist") as fh:
                    node_cores = [c for c in _cpulist(fh.read()) if c in set(allowed)]
        except (OSError, ValueError, AttributeError, TypeError):
            node, node_cores = None, None
    if node_cores and len(node_cores) >= local_world:
        # ranks on the same node cannot see each other's choice without a collective: every rank takes slot local_rank of
        # local_world equal slots of ITS node's cores (disjoint within a node; nodes are disjoint by construction)
        per = max(1, len(node_cores) // local_world)
        mine = node_cores[local_rank * per:(local_rank + 1) * per]
        how = f"NUMA node {node} of the GPU, slot {local_rank}/{local_world}"
    else:
        per = max(1, len(allowed) // local_world)
        mine = allowed[local_rank * per:(local_rank + 1) * per] or allowed
        node = None if not node_cores else node
        how = f"even split of the affinity mask, slot {local_rank}/{local_world}"
    try:
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), 8)))
    except OSError as e:
        return {"cores": None, "n": len(allowed), "numa_node": node, "how": f"sched_setaffinity failed: {e}"}
    return {"cores": f"{mine[0]}-{mine[-1]}" if mine == list(range(mine[0], mine[-1] + 1)) else ",".join(map(str, mine)),
            "n": len(mine), "numa_node": node, "how": how}


def rank_bindings(rank: int, world: int, local_rank: int, device: torch.device, group=None, pin: bool = False) -> dict:
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
    # Reporting by default.  ``pin=True`` (bench.py's N > 1 runs ask for it explicitly) ALSO narrows this process's affinity mask and
    # torch's thread count — a side effect every thread and child created later inherits (DataLoader workers, subprocesses) — so it
    # is the caller's decision, and the split is over the ranks of THIS node: LOCAL_WORLD_SIZE when the launcher sets it (torchrun
    # does), else the number of visible devices, never the global world size of a multi-node job.
    if pin:
        local_world = os.environ.get("LOCAL_WORLD_SIZE")
        if local_world is None:
            local_world = torch.cuda.device_count() if device.type == "cuda" and torch.cuda.device_count() > 0 else world
        mine["affinity"] = pin_rank_to_cores(local_rank, max(1, min(int(local_world), world)), device)
    else:
        try:
            n_allowed = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            n_allowed = 0
        mine["affinity"] = {"cores": None, "n": n_allowed, "numa_node": None, "how": "not pinned (rank_bindings(pin=False))"}
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
