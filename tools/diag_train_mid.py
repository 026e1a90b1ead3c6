"""Dev tool: the mid-size (128x160) training step on the GPU against tests/golden/train_small.npz with the whole-FeatureNet HIP
function on and off (worst / median relative gradient errors), and the direct FeatureNetTrainFn-vs-modules check at three sizes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_training as TT          # noqa: E402
from enerf_amd.lib import get_lib   # noqa: E402

dev = torch.device("cuda:0")
for hw in ((32, 64), (128, 160), (512, 640)):
    try:
        TT._check_feature_net_train(get_lib(), dev, H=hw[0], W=hw[1], tol=2e-4)
        print("feature_net_train vs modules", hw, "ok at 2e-4")
    except AssertionError as e:
        print("feature_net_train vs modules", hw, "FAILED", str(e)[:200])
g2 = np.load(os.path.join(TT.GOLDEN, "train_small.npz"))
cfg2, batch2 = TT._train_batch(**TT.TRAIN_CASES["train_small"])
batch2 = {k: v.to(dev) for k, v in batch2.items()}
for hipfn in (True, False, True, False):
    net = TT._net(cfg2).to(dev)
    net.hip_feature_net_train = hipfn
    loss = TT._loss(net(batch2), batch2)
    loss.backward()
    errs = TT._grad_errors([(n, p.grad) for n, p in net.named_parameters() if p.grad is not None], g2)
    v = sorted(errs.items(), key=lambda kv: -kv[1])
    print("hip featnet", hipfn, "loss", float(loss.detach()), "worst", [(k, round(e, 4)) for k, e in v[:5]], "median",
          float(np.median(list(errs.values()))))
