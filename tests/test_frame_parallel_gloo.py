"""world_size-2 gloo run of the frame-parallel renderer (the N>1 path), on the CPU lane emulator."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from enerf_amd.frame_parallel import frames_of_rank

HERE = os.path.dirname(os.path.abspath(__file__))


def test_frame_assignment_partitions_all_frames():
    for n, g in [(7, 2), (8, 8), (3, 4), (0, 2)]:
        got = sorted(f for r in range(g) for f in frames_of_rank(n, r, g))
        assert got == list(range(n))


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ENERF_EMU_THREADS="2")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from enerf_amd.frame_parallel import render_sharded
    from enerf_amd.network import Network
    from enerf_amd.synth import make_batch
    from emu_lib import emu_lib
    from golden_cases import case_config, load_weights
    cfg = case_config("tiny_s4_mask")
    net = Network(cfg, lib=emu_lib()).eval()
    net.load_state_dict(load_weights(), strict=False)

    def render(f):
        b = {k: torch.from_numpy(v) for k, v in make_batch(32, 64, 3, cfg, seed=50 + f, textured=True).items()}
        return float(net(b)["rgb_level1"].double().sum())
    outs, fps, tmax = render_sharded(render, 5, rank, world)
    q.put((rank, outs, fps, tmax))
    dist.destroy_process_group()


def test_two_rank_gloo_frame_parallel_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in procs]
    [p.join(60) for p in procs]
    merged = {}
    for rank, outs, fps, tmax in res:
        assert sorted(outs) == frames_of_rank(5, rank, 2)
        merged.update(outs)
        assert fps == pytest.approx(5 / tmax)
    assert sorted(merged) == list(range(5))
    assert res[0][2] == res[1][2]                       # both ranks agree on the aggregate
    # same frames rendered in this process give the same numbers (no cross-rank interference)
    sys.path.insert(0, HERE)
    from enerf_amd.network import Network
    from enerf_amd.synth import make_batch
    from emu_lib import emu_lib
    from golden_cases import case_config, load_weights
    cfg = case_config("tiny_s4_mask")
    net = Network(cfg, lib=emu_lib()).eval()
    net.load_state_dict(load_weights(), strict=False)
    b = {k: torch.from_numpy(v) for k, v in make_batch(32, 64, 3, cfg, seed=53, textured=True).items()}
    assert float(net(b)["rgb_level1"].double().sum()) == pytest.approx(merged[3], rel=1e-6)   # torch thread count differs
