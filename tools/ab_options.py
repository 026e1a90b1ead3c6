"""A/B of enerf_options_t variants on one workload: per-frame-sync latency, back-to-back rate and (single-stream) stage times.
python tools/ab_options.py dtu "name=field:val,field:val" ...   e.g.  default= b4off=conv3d_b4:1"""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import _seeded_network
from enerf_amd.lib import Options
wl = sys.argv[1]
specs = sys.argv[2:] or ["default="]
sys.argv = sys.argv[:1]
from bench import make_workload, StageTimer
dev = torch.device("cuda:0")
cfg, b, human, _ = make_workload(wl, 0)
net = _seeded_network(cfg, dev, human=human).eval()
batch = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
def mk(spec, **extra):
    kw = dict(extra)
    for f in filter(None, spec.split(",")):
        k, v = f.split(":"); kw[k] = int(v)
    return Options(**kw) if kw else None
def run(opt, n=200):
    net.options = opt
    with torch.no_grad():
        for _ in range(20): net(batch)
        torch.cuda.synchronize(); ts = []
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter(); net(batch); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): net(batch)
        torch.cuda.synchronize(); seq = n / (time.perf_counter() - t0)
    return 1e3 * sum(ts) / len(ts), seq
for rep in range(2):
    for spec in specs:
        name, _, fields = spec.partition("=")
        m, seq = run(mk(fields))
        line = f"{wl} {name:10s} latency {m:.4f} ms -> {1e3/m:7.1f} FPS ; back-to-back {seq:7.1f} FPS"
        if rep == 1:
            timer = StageTimer(); net._timer = timer; net.options = mk(fields, single_stream=1)
            with torch.no_grad():
                for _ in range(30): net(batch)
            torch.cuda.synchronize(); net._timer = None
            st = timer.summary()
            line += " ; " + " ".join(f"{k}={v*1e3:.0f}" for k, v in st.items() if k.startswith(("cost_reg", "feature")))
        print(line, flush=True)
