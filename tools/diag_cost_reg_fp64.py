"""GPU diagnostic: is the HIP cost-regularisation training stage faithful on the ACTUAL tensors of a training step?  Captures the
cost volume entering cost_reg_i and the gradients arriving at its outputs during one 128x160 (or 512x640) step on the GPU, replays
the stage in float64 on the CPU (the network's own modules, .double()) and compares every parameter gradient of the stage.
usage: python tools/diag_cost_reg_fp64.py [small|full]"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_training import FULL_TRAIN_CASE, TRAIN_CASES, _loss, _net, _train_batch  # noqa: E402
from enerf_amd import autograd as A  # noqa: E402
import torch_twins as T  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "small"
kw = FULL_TRAIN_CASE if which == "full" else TRAIN_CASES["train_small"]
dev = torch.device("cuda:0")
cfg, batch = _train_batch(**kw)
batch = {k: v.to(dev) for k, v in batch.items()}
net = _net(cfg).to(dev)
cap = {}
orig = A.cost_reg_train


def spy(lib, m, vol):
    i = 0 if not m.full else 1
    feat, prob = orig(lib, m, vol)
    cap[i] = {"vol": vol.detach().clone()}
    feat.register_hook(lambda g, i=i: cap[i].__setitem__("g_feat", g.detach().clone()))
    prob.register_hook(lambda g, i=i: cap[i].__setitem__("g_prob", g.detach().clone()))
    return feat, prob


A.cost_reg_train = spy
_loss(net(batch), batch).backward()
torch.cuda.synchronize()
for i in (0, 1):
    m = getattr(net, f"cost_reg_{i}")
    hip = {n: p.grad.detach().cpu().double() for n, p in m.named_parameters()}
    for dtype in (torch.float64, torch.float32):
        m2 = copy.deepcopy(m).cpu().to(dtype).train()
        for p in m2.parameters():
            p.grad = None
        vol = cap[i]["vol"].cpu().to(dtype)
        feat, prob = T.cost_reg_forward(m2, vol)
        torch.autograd.backward([feat, prob], [cap[i]["g_feat"].cpu().to(dtype), cap[i]["g_prob"].cpu().to(dtype)])
        if dtype == torch.float64:
            truth = {n: p.grad.double() for n, p in m2.named_parameters()}
        else:
            cpu32 = {n: p.grad.double() for n, p in m2.named_parameters()}
    rows = []
    for n, t in truth.items():
        sc = max(float(t.abs().max()), 1e-30)
        rows.append((float((hip[n] - t).abs().max()) / sc, float((cpu32[n] - t).abs().max()) / sc, n))
    rows.sort(reverse=True)
    print(f"cost_reg_{i}: HIP stage vs fp64 twin on the step's own tensors: median {np.median([r[0] for r in rows]):.2e} (torch CPU fp32 twin: "
          f"{np.median([r[1] for r in rows]):.2e}); worst " + "; ".join(f"{n} {a:.1e} (cpu32 {b:.1e})" for a, b, n in rows[:6]))
