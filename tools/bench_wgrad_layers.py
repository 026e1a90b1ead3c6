"""Dev tool: time enerf_conv_wgrad for the cost-volume networks' layers at the config-5 training shapes (one MI355X)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enerf_amd.lib import get_lib

lib = get_lib()
dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# (name, A grid (D,H,W), Ca, B grid, Cb, stride): conv: A = dY (out grid), B = X (in grid); transposed: A = X coarse, B = dY fine
def layers(D, h, w, cin, full):
    L = [("conv0", (D, h, w), 8, (D, h, w), cin, 1), ("conv1", (D // 2, h // 2, w // 2), 16, (D, h, w), 8, 2),
         ("conv2", (D // 2, h // 2, w // 2), 16, (D // 2, h // 2, w // 2), 16, 1),
         ("conv3", (D // 4, h // 4, w // 4), 32, (D // 2, h // 2, w // 2), 16, 2),
         ("conv4", (D // 4, h // 4, w // 4), 32, (D // 4, h // 4, w // 4), 32, 1)]
    if full:
        L += [("conv5", (D // 8, h // 8, w // 8), 64, (D // 4, h // 4, w // 4), 32, 2),
              ("conv6", (D // 8, h // 8, w // 8), 64, (D // 8, h // 8, w // 8), 64, 1),
              ("conv7T", (D // 8, h // 8, w // 8), 64, (D // 4, h // 4, w // 4), 32, 2)]
    L += [("conv9T", (D // 4, h // 4, w // 4), 32, (D // 2, h // 2, w // 2), 16, 2),
          ("conv11T", (D // 2, h // 2, w // 2), 16, (D, h, w), 8, 2), ("heads", (D, h, w), 16, (D, h, w), 8, 1)]
    return L


tot = 0.0
for lvl, (D, h, w, cin, full) in enumerate(((64, 64, 80, 32, False), (8, 256, 320, 16, True))):
    for name, ga, Ca, gb, Cb, stride in layers(D, h, w, cin, full):
        a = torch.randn(1, *ga, Ca, device=dev)
        b = torch.randn(1, *gb, Cb, device=dev)
        t = timed(lambda: lib.conv_wgrad_cl(a, b, stride))
        npos = ga[0] * ga[1] * ga[2]
        floor = npos / 4 * 27 * ((Ca + 15) // 16) * ((Cb + 15) // 16) * 32 / 1024 / 2400
        gf = 2 * 27 * npos * Ca * Cb / 1e9
        tot += t
        print(f"L{lvl} {name:8s} A {ga} x{Ca:2d}  B x{Cb:2d} s{stride}: {t:8.1f} us   16x16x4 issue floor {floor:6.1f} us   {gf / t * 1e-3:6.2f} TF/s")
print("total", tot, "us")
