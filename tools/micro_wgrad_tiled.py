"""Dev tool (GPU): the training step's big weight-gradient layers alone, at dtu_pretrain shapes, for rocprofv3 (kernel stats / --pmc).
    python tools/micro_wgrad_tiled.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enerf_amd.lib import get_lib

lib, dev, reps = get_lib(), torch.device("cuda:0"), int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = torch.Generator(device="cpu").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
cases2d = [("conv0.1 8<-8", 8, 8), ("conv0.0 8<-3", 8, 3), ("smooth0 8<-32", 8, 32), ("conv1.1 16<-16 (old kernel)", 16, 16)]
for name, ca, cb in cases2d:
    H, W = (256, 320) if ca == 16 else (512, 640)
    a, b = rnd(3, H, W, ca), rnd(3, H, W, cb)
    for _ in range(reps):
        lib.conv_wgrad_cl2d(a, b, 3, 1)
for name, ca, cb, D, H, W in [("conv0 L1 8<-16", 8, 16, 8, 256, 320), ("conv0 L0 8<-32", 8, 32, 64, 64, 80), ("heads L1 16<-8", 16, 8, 8, 256, 320),
                              ("heads L0 16<-8", 16, 8, 64, 64, 80)]:
    a, b = rnd(1, D, H, W, ca), rnd(1, D, H, W, cb)
    for _ in range(reps):
        lib.conv_wgrad_cl(a, b, 1)
torch.cuda.synchronize()
print("done")
