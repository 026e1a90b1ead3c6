import sys, os, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch.nn.functional as F
from __graft_entry__ import _seeded_network
from enerf_amd.config import EnerfConfig
from enerf_amd.synth import make_batch
dev = torch.device("cuda:0")
cfg = EnerfConfig()
net = _seeded_network(cfg, dev).train()
opt = torch.optim.Adam(net.parameters(), lr=5e-4)
b = make_batch(512, 640, 3, cfg, seed=0, textured=True)
rng = np.random.default_rng(0)
for i in range(2):
    b[f"rgb_{i}"] = rng.uniform(0, 1, size=(1, b[f"rays_{i}"].shape[1], 3)).astype(np.float32)
batch = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
def step():
    out = net(batch)
    loss = sum(w * F.mse_loss(batch[f"rgb_{i}"], out[f"rgb_level{i}"]) for i, w in enumerate((0.1, 1.0)))
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
tab = prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60)
print(tab)
# the torch-op glue by source line: which lines of enerf_amd/ launch the library (non-enerf) kernels
import collections
by_line, by_op = collections.Counter(), collections.Counter()
for ev in prof.events():
    t = getattr(ev, "self_device_time_total", 0)
    if t > 0 and ev.name.startswith("aten::"):
        by_op[ev.name] += t
        fr = next((f for f in (ev.stack or []) if "enerf_amd/" in f), "?")
        by_line[fr.split("enerf_amd/")[-1][:100]] += t
print("== aten ops by self device time (us)")
for k, v in by_op.most_common(25):
    print(f"{v:9.1f}  {k}")
print("== source lines by the self device time of the aten ops they launch (us)")
for k, v in by_line.most_common(40):
    print(f"{v:9.1f}  {k}")
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue per step {1e3*(t1-t0)/5:.1f} ms ; wall per step {1e3*(t2-t0)/5:.1f} ms")
