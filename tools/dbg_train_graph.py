"""Eager vs graphed training steps at bench size: per-step loss and time (debug aid)."""
import os, sys, time
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import _seeded_network
from enerf_amd.config import EnerfConfig
from enerf_amd.synth import make_batch
from enerf_amd.train_graph import GraphedTrainStep, mse_loss
dev = torch.device("cuda:0")
cfg = EnerfConfig()
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 640)
b = make_batch(H, W, 3, cfg, seed=0, textured=True)
rng = np.random.default_rng(0)
for i in range(2):
    b[f"rgb_{i}"] = rng.uniform(0, 1, size=(1, b[f"rays_{i}"].shape[1], 3)).astype(np.float32)
batch = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
loss_fn = lambda out, bt: sum(w * mse_loss(bt[f"rgb_{i}"], out[f"rgb_level{i}"]) for i, w in enumerate((0.1, 1.0)))
for mode in ("eager", "graph"):
    net = _seeded_network(cfg, dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=5e-4, capturable=True)
    if mode == "graph":
        g = GraphedTrainStep(net, opt, loss_fn, batch, warmup=3)
        step = lambda: g(batch)
    else:
        def step():
            out = net(batch); loss = loss_fn(out, batch); opt.zero_grad(); loss.backward()
            torch.nn.utils.clip_grad_value_(net.parameters(), 40); opt.step(); return loss
        for _ in range(4): step()      # same number of updates as warm-up + capture
    losses, times = [], []
    for _ in range(14):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        l = step(); torch.cuda.synchronize(); times.append(1e3 * (time.perf_counter() - t0)); losses.append(float(l))
    print(mode, "ms/step", [round(t, 1) for t in times])
    print(mode, "loss", [round(x, 5) for x in losses])
    bad = [n for n, p in net.named_parameters() if not torch.isfinite(p).all()]
    print(mode, "non-finite params:", bad[:5])

# ---- first-replay gradient comparison ----
nets = [_seeded_network(cfg, dev).train() for _ in range(2)]
opts = [torch.optim.SGD(n.parameters(), lr=0.0) for n in nets]          # lr 0: parameters never move, gradients comparable
g = GraphedTrainStep(nets[0], opts[0], loss_fn, batch, warmup=1, clip_value=None)
for rep in range(6):
    l0 = float(g(batch))
    out = nets[1](batch); l1 = loss_fn(out, batch); opts[1].zero_grad(); l1.backward()
    worst = []
    for (n, p0), (_, p1) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        if p0.grad is None or p1.grad is None:
            if (p0.grad is None) != (p1.grad is None): worst.append((float("inf"), n))
            continue
        d = float((p0.grad - p1.grad).abs().max()); s_ = float(p1.grad.abs().max()) + 1e-12
        worst.append((d / s_ if d == d else float('inf'), n))
    worst.sort(reverse=True)
    pd = dict(nets[0].named_parameters()); pe = dict(nets[1].named_parameters())
    for _, n in worst[:2]:
        print("   ", n, "graph", pd[n].grad.flatten()[:4].tolist(), "eager", pe[n].grad.flatten()[:4].tolist())
    print(f"replay {rep}: loss graph {l0:.6f} eager {float(l1):.6f}; worst relative grad mismatches:", [(round(a, 4), n) for a, n in worst[:6]])
