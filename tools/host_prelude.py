"""How much host time sits in front of the frame's first kernel launch and behind its last kernel (per-frame-sync protocol)?
python tools/host_prelude.py   (through gpurun): python prelude of Network.forward, duration of the C call, sync wake-up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import _seeded_network
sys.argv = sys.argv[:1]
from bench import make_workload
dev = torch.device("cuda:0")
cfg, b, human, _ = make_workload("dtu", 0)
net = _seeded_network(cfg, dev, human=human).eval()
batch = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
lib = net.lib
orig = lib.forward
marks = {}
def fwd(a, sid):
    marks["enter_c"] = time.perf_counter()
    r = orig(a, sid)
    marks["leave_c"] = time.perf_counter()
    return r
lib.forward = fwd
with torch.no_grad():
    for _ in range(300):
        net(batch); torch.cuda.synchronize()
    pre, cc, post, tot, syn = [], [], [], [], []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net(batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pre.append(marks["enter_c"] - t0); cc.append(marks["leave_c"] - marks["enter_c"]); post.append(t1 - marks["leave_c"]); tot.append(t2 - t0); syn.append(t2 - t1)
    med = lambda v: sorted(v)[len(v) // 2] * 1e6
    print(f"python before the C call {med(pre):.1f} us | enerf_forward (all enqueues) {med(cc):.1f} us | python after {med(post):.1f} us | "
          f"synchronize() wait {med(syn):.1f} us | frame latency {med(tot):.1f} us")
    # GPU time of the frame by events on the stream (includes the launch gaps, not the host prelude / wake-up)
    g = []
    for _ in range(100):
        torch.cuda.synchronize(); ev0.record(); net(batch); ev1.record(); torch.cuda.synchronize(); g.append(ev0.elapsed_time(ev1) * 1e3)
    print(f"event-to-event GPU time of a frame {sorted(g)[50]:.1f} us")
