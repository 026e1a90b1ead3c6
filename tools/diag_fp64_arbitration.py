"""GPU diagnostic: which HIP training stage moves the gradients away from the float64 reference step?  For the 128x160 (and
optionally the 512x640) fixture: the fp64 arbitration statistics (tests/test_training.py) with every HIP stage on, and with
one stage at a time routed through its torch-op twin (tests/torch_twins.py).   usage: python tools/diag_fp64_arbitration.py [small|full] ..."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_training import (FULL_TRAIN_CASE, GOLDEN, TRAIN_CASES, _distance_to_fp64, _loss, _net, _train_batch)  # noqa: E402

import torch_twins  # noqa: E402  (tests/torch_twins.py: the torch-op twins of the stages, test infrastructure)

# a stage name routes that stage through its torch twin (torch_twins.install); "all" = every stage: PyTorch-ROCm end to end
TOGGLES = ["depth_values", "rays", "gather", "mlp", "cost_reg", "feature_net", "feature_volume", "depth_regression", "composite"]


def run(case, sparse, off):
    kw = FULL_TRAIN_CASE if sparse else TRAIN_CASES[case]
    name = "train_full" if sparse else case
    g32, g64 = np.load(os.path.join(GOLDEN, name + ".npz")), np.load(os.path.join(GOLDEN, name + "_fp64.npz"))
    dev = torch.device("cuda:0")
    cfg, batch = _train_batch(**kw)
    batch = {k: v.to(dev) for k, v in batch.items()}
    net = _net(cfg).to(dev)
    if off:
        torch_twins.install(net, *([] if "all" in off else off))
    _loss(net(batch), batch).backward()
    dist = _distance_to_fp64([(n, p.grad) for n, p in net.named_parameters() if p.grad is not None], g32, g64, sparse=sparse)
    ours = np.array([d[0] for d in dist.values()]); ref = np.array([d[1] for d in dist.values()])
    floor = float(np.median(ref))
    ratios = sorted(((d[0] / max(d[1], floor), n, d) for n, d in dist.items()), reverse=True)
    print(f"{name:12s} off={','.join(off) or '-':40s} median ours {np.median(ours):.2e} ref {floor:.2e}  max ours {ours.max():.2e} ref {ref.max():.2e}  "
          f"violations(>3x) {sum(r[0] > 3 for r in ratios)}  worst: " + "; ".join(f"{n} {r:.1f}x ({d[0]:.1e}/{d[1]:.1e})" for r, n, d in ratios[:4]))


def only_on():
    """every stage on torch ops except ONE (which HIP stage alone moves which parameter?)"""
    for t in TOGGLES:
        run("train_small", False, [x for x in TOGGLES if x != t])


if __name__ == "__main__":
    if sys.argv[1:] == ["only_on"]:
        only_on()
        sys.exit(0)
    for which in (sys.argv[1:] or ["small"]):
        sparse = which == "full"
        run("train_small", sparse, [])
        run("train_small", sparse, [])                      # run-to-run (atomics order)
        for t in TOGGLES:
            run("train_small", sparse, [t])
        run("train_small", sparse, TOGGLES)                 # every stage on torch ops (MIOpen etc.); conv weight gradients still HIP
        run("train_small", sparse, ["all"])                 # every stage through its torch twin: PyTorch-ROCm end to end
