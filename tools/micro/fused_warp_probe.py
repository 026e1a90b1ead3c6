"""Driver of tools/micro/fused_warp_probe.hip: DTU config-2 shapes, both cascade levels.
Prints, per level: the product's warp kernel, today's conv0 staging traffic as a box copy, and the staging phase of a fused
warp-in-conv0 kernel (haloed boxes warped into LDS).  Run on the GPU box: python tools/micro/fused_warp_probe.py"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from enerf_amd.config import EnerfConfig          # noqa: E402
from enerf_amd.lib import EnerfLib                # noqa: E402
from enerf_amd.synth import make_batch            # noqa: E402

SO = os.path.join(ROOT, "tools", "micro", "fwp.so")
if not os.path.exists(SO):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-fno-slp-vectorize",
                    "-I", os.path.join(ROOT, "enerf_amd/csrc"), "-I", os.path.join(ROOT, "include"), "-shared", "-fPIC",
                    os.path.join(ROOT, "tools/micro/fused_warp_probe.hip"), "-o", SO], check=True)
dll = ctypes.CDLL(SO)
dll.probe_launch.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 9 + [ctypes.c_void_p, ctypes.c_void_p]
lib = EnerfLib()
dev = torch.device("cuda:0")
cfg = EnerfConfig().with_cas(volume_planes=(48, 8), render_if=(False, True))
b = {k: torch.from_numpy(v).to(dev) for k, v in make_batch(512, 640, 3, cfg, seed=3, textured=True).items()}
H, W, S, B = 512, 640, 3, 1


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


prev = None
for level, (C, fscale, vscale, D) in enumerate(((32, 0.25, 0.125, 48), (16, 0.5, 0.5, 8))):
    Hs, Ws, h, w = int(H * fscale), int(W * fscale), int(H * vscale), int(W * vscale)
    feat = torch.randn(B, S, Hs, Ws, C, device=dev)
    proj, dv, nf = lib.level_prep(b["src_ixts"], b["src_exts"], b["tar_ixt"], b["tar_ext"], fscale, vscale, b["near_far"], prev,
                                  D, h, w, cfg.cas.depth_inv[level])
    vol = lib.build_feature_volume(feat, proj, dv, C)
    nb = B * ((D + 3) // 4) * ((h + 7) // 8) * ((w + 15) // 16)
    sink = torch.empty(nb * 256, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def probe(mode):
        rc = dll.probe_launch(feat.data_ptr(), proj.data_ptr(), dv.data_ptr(), vol.data_ptr(), B, S, C, Hs, Ws, D, h, w, mode,
                              sink.data_ptr(), st)
        assert rc == nb, rc
    # the warped box must be the product's volume: compare the token sums of mode 0 and mode 1
    probe(0); s0 = sink.clone(); probe(1); s1 = sink.clone()
    torch.cuda.synchronize()
    err = float((s0 - s1).abs().max() / s1.abs().max())
    t_warp = timed(lambda: lib.build_feature_volume(feat, proj, dv, C))
    t_copy, t_fused, t_halo = timed(lambda: probe(1)), timed(lambda: probe(0)), timed(lambda: probe(2))
    print(f"level {level}: C={C} D={D} {h}x{w}  boxes {nb} (1080 haloed voxels each = {1080 * nb / (D * h * w):.2f}x the volume)")
    print(f"  k_feature_volume (product, writes the volume)            {t_warp:7.1f} us")
    print(f"  box copy of the precomputed volume into LDS (today)      {t_copy:7.1f} us")
    print(f"  warp of the haloed boxes into LDS (fused staging phase)  {t_fused:7.1f} us   (box sums vs today's: rel {err:.1e})")
    print(f"  the same gathers with a 16 KB LDS ring (full occupancy)  {t_halo:7.1f} us")
    print(f"  fused staging - (warp kernel + box copy) = {t_fused - t_warp - t_copy:+.1f} us per frame")
    # a plausible previous level for level 1's depth range
    mid = float(b["near_far"].mean())
    prev = (torch.full((B, h, w), mid, device=dev), torch.full((B, h, w), 0.02 * mid, device=dev), nf)
