"""Dev probe: frames/s with 1..4 frames in flight on separate HIP streams (enerf_amd/pipeline.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _seeded_network
from enerf_amd.config import EnerfConfig
from enerf_amd.pipeline import FramePipeline
from enerf_amd.synth import make_batch

cfg = EnerfConfig.dtu_eval()
dev = torch.device("cuda:0")
net = _seeded_network(cfg, dev)
batch = {k: torch.from_numpy(v).to(dev) for k, v in make_batch(512, 640, 3, cfg, seed=0, textured=True).items()}
for depth in (1, 2, 3, 4):
    pipe = FramePipeline(net, depth)
    for _ in range(12):
        pipe.submit(batch)
    pipe.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        pipe.submit(batch)
    pipe.join(); torch.cuda.synchronize()
    print(f"depth={depth}: {300 / (time.perf_counter() - t0):.1f} frames/s")
