"""Dev probe: frames/s when B target views are rendered per forward (same API, batch dimension B)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import _seeded_network
from enerf_amd.config import EnerfConfig
from enerf_amd.synth import make_batch

cfg = EnerfConfig.dtu_eval()
dev = torch.device("cuda:0")
net = _seeded_network(cfg, dev)
for B in (1, 2, 4):
    b = make_batch(512, 640, 3, cfg, seed=0, textured=True, B=B)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
    for _ in range(10):
        net(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        net(batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"B={B}: {B * n / dt:.1f} frames/s, {1e3 * dt / n:.3f} ms/forward")
