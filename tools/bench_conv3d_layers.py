"""GPU microbenchmark of the cost-regularisation middle layers (DTU shapes) under enerf_options_t variants.
    python tools/bench_conv3d_layers.py [reps]        (run through gpurun; prints one row per layer: us per variant)
Each layer is enqueued `reps` times back to back on one stream between two HIP events (L2-warm: relative numbers)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from enerf_amd.lib import Options, get_lib

lib = get_lib()
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
S1, S2, T2 = 0, 1, 2
LAYERS = [  # name, kind, cin, cout, input (D,h,w), residual
    ("L1.conv1", S2, 8, 16, (8, 256, 320), False), ("L1.conv2", S1, 16, 16, (4, 128, 160), False),
    ("L1.conv3", S2, 16, 32, (4, 128, 160), False), ("L1.conv4", S1, 32, 32, (2, 64, 80), False),
    ("L1.conv5", S2, 32, 64, (2, 64, 80), False), ("L1.conv6", S1, 64, 64, (1, 32, 40), False),
    ("L1.conv7", T2, 64, 32, (1, 32, 40), True), ("L1.conv9", T2, 32, 16, (2, 64, 80), True),
    ("L1.conv11", T2, 16, 8, (4, 128, 160), True),
    ("L0.conv1", S2, 8, 16, (48, 64, 80), False), ("L0.conv2", S1, 16, 16, (24, 32, 40), False),
    ("L0.conv3", S2, 16, 32, (24, 32, 40), False), ("L0.conv4", S1, 32, 32, (12, 16, 20), False),
    ("L0.conv9", T2, 32, 16, (12, 16, 20), True), ("L0.conv11", T2, 16, 8, (24, 32, 40), True),
]
VARIANTS = {"default": Options(), "small=1": Options(conv3d_small_variant=1), "t2=1": Options(conv3d_t2_variant=1),
            "t2=2": Options(conv3d_t2_variant=2), "global": Options(conv3d_global_only=1), "default2": Options()}
if os.environ.get("ENERF_LAYER_VARIANTS"):                # e.g. "default,small=1"
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ["ENERF_LAYER_VARIANTS"].split(",")}
g = torch.Generator().manual_seed(0)
_w = torch.randn(4096, 4096, device=dev)
for _ in range(40):                                     # clocks up before the first timed kernel
    _w = (_w @ _w).clamp_(-1, 1)
torch.cuda.synchronize()
print(f"{'layer':10s} " + " ".join(f"{k:>9s}" for k in VARIANTS))
for name, kind, cin, cout, (D, h, w), res in LAYERS:
    wshape = (cin, cout, 3, 3, 3) if kind == T2 else (cout, cin, 3, 3, 3)
    wt = (torch.randn(wshape, generator=g) * 0.1).to(dev)
    x = torch.randn((1, D, h, w, cin), generator=g).to(dev)
    packed = lib.conv3d_layer_pack(wt, cin, cout, kind)
    ref = None
    row = []
    for vn, opt in VARIANTS.items():
        applies = (vn.startswith("t2") and kind == T2) or (vn.startswith("small") and kind != T2 and cin >= 16) or vn in ("default", "global", "default2")
        if not applies:
            row.append("        -")
            continue
        out = lib.conv3d_layer(packed, cin, cout, kind, x, None, opt)
        r = torch.randn(out.shape, generator=g).to(dev) if res else None
        out = lib.conv3d_layer(packed, cin, cout, kind, x, r, opt)
        if ref is None:
            ref = (out - (r if r is not None else 0)).clone()
        else:
            err = float(((out - (r if r is not None else 0)) - ref).abs().max() / ref.abs().max())
            assert err < 1e-4, (name, vn, err)
        best = 1e9
        for _ in range(3):                              # best of three batches
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                lib.conv3d_layer(packed, cin, cout, kind, x, r, opt)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
        row.append(f"{best:9.2f}")
    print(f"{name:10s} " + " ".join(row), flush=True)
