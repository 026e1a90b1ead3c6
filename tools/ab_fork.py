"""A/B: enerf_forward with the FeatureNet side lane (default) vs single_stream=1: per-frame-sync latency, back-to-back
frames, six frames in flight.  python tools/ab_fork.py [dtu|lego|zju]"""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import _seeded_network
from enerf_amd.config import EnerfConfig
from enerf_amd.lib import Options, throughput_options
from enerf_amd.pipeline import FramePipeline
wl = sys.argv[1] if len(sys.argv) > 1 else "dtu"
dev = torch.device("cuda:0")
sys.argv = sys.argv[:1]
from bench import make_workload
cfg, b, human, _ = make_workload(wl, 0)
net = _seeded_network(cfg, dev, human=human)
net.eval()
batch = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
def lat(opt, n=300):
    net.options = opt
    with torch.no_grad():
        for _ in range(30): net(batch)
        torch.cuda.synchronize(); ts = []
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter(); net(batch); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): net(batch)
        torch.cuda.synchronize(); seq = n / (time.perf_counter() - t0)
    ts.sort()
    return 1e3 * sum(ts) / len(ts), 1e3 * ts[len(ts) // 2], seq
for name, opt in (("fork", None), ("single", Options(single_stream=1)), ("fork", None), ("single", Options(single_stream=1))):
    m, p50, seq = lat(opt)
    print(f"{wl} {name:7s} latency mean {m:.4f} ms p50 {p50:.4f} -> {1e3/m:.1f} FPS ; back-to-back {seq:.1f} FPS", flush=True)
if not human:
    for name, o in (("lanes", Options()), ("single_stream", throughput_options()), ("lanes", Options()), ("single_stream", throughput_options())):
        net.options = None
        pipe = FramePipeline(net, depth=6, options=o)
        with torch.no_grad():
            for _ in range(30): pipe.submit(batch)
            pipe.join(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(1000): pipe.submit(batch)
            pipe.join(); torch.cuda.synchronize()
        print(f"{wl} pipelined x6 {name}: {1000 / (time.perf_counter() - t0):.1f} FPS", flush=True)
        pipe.close()
