"""Dev tool (GPU): enerf_channel_sums on the training step's largest BatchNorm inputs against a plain streaming read of the
same bytes (torch sum), per call, kernels alone."""
import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enerf_amd.lib import get_lib
lib = get_lib(); dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for n, C in ((983040, 8), (245760, 32), (245760, 16), (655360, 8), (983040, 32), (61440, 32)):
    x = torch.randn(n, C, device=dev); y = torch.randn(n, C, device=dev); z = torch.randn(n, C, device=dev)
    ms, mh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mb = n * C * 4 / 1e6
    a = t(lambda: lib.channel_sums_raw(x, x)); b = t(lambda: lib.channel_sums_raw(x, y)); c = t(lambda: lib.channel_sums_raw(x, y, z, ms, mh))
    d = t(lambda: x.sum(0))
    print(f"n={n} C={C} ({mb:.1f} MB): same {a:.1f} us ({mb/a*1e-0:.2f} MB/us = TB/s x1e-0) | a,b {b:.1f} us ({2*mb/b:.2f}) | a,b,mask {c:.1f} us ({3*mb/c:.2f}) | torch sum(0) {d:.1f} us ({mb/d:.2f})")
