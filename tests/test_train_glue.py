"""ABI v7 (train_glue.hip): the parts of the training step that were eager PyTorch ops until round 3, each against its torch
twin (tests/torch_twins.py) — the very expressions enerf_amd/train_path.py ran before round 4 (which restate utils.py:98-151, 390-441; feature_net.py:11,14).
CPU: the kernel sources on the lane emulator.  GPU (-m gpu): the product library."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import torch_twins as TP
from enerf_amd.config import EnerfConfig


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


def _check_s2k5_dgrad(lib, dev):
    g = torch.Generator().manual_seed(0)
    for cin, cout, N, H, W in ((8, 16, 2, 24, 40), (16, 32, 3, 16, 36), (8, 16, 1, 64, 96)):
        w = (torch.randn(cout, cin, 5, 5, generator=g) * 0.1).to(dev)
        dz = torch.randn(N, H // 2, W // 2, cout, generator=g).to(dev)
        add = torch.randn(N, H, W, cin, generator=g).to(dev)
        ref = torch.nn.grad.conv2d_input((N, cin, H, W), w.cpu(), dz.cpu().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
        got = lib.conv2d_s2k5_dgrad(w, dz)
        assert got.shape == (N, H, W, cin)
        assert _rel(got, ref) < 2e-5, (cin, cout, _rel(got, ref))
        got = lib.conv2d_s2k5_dgrad(w, dz, add=add)
        assert _rel(got, ref + add.cpu()) < 2e-5


def _check_resize_adjoint(lib, dev):
    g = torch.Generator().manual_seed(1)
    for (hc, wc), k in (((8, 10), 2), ((6, 9), 4), ((5, 7), 1), ((7, 3), 3)):
        x = torch.randn(3, 2, hc, wc, generator=g, requires_grad=True)
        y = F.interpolate(x, None, scale_factor=k, mode="bilinear", align_corners=True, recompute_scale_factor=True) if k != 1 else x * 1.0
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        add = torch.randn(3, 2, hc, wc, generator=g)
        got = lib.resize_ac_adjoint(gy.to(dev).contiguous(), hc, wc)
        assert _rel(got, x.grad) < 1e-5, (hc, wc, k, _rel(got, x.grad))
        got = lib.resize_ac_adjoint(gy.to(dev).contiguous(), hc, wc, add=add.to(dev))
        assert _rel(got, x.grad + add) < 1e-5


def _level_inputs(g, B, hp, wp, clamp_share=0.4):
    """Previous-level maps in disparity space with a share of the pixels on each clamp (utils.py:122-127)."""
    nf0 = 1.0 / (425.0 + 20 * torch.rand(B, hp, wp, generator=g))
    nf1 = 1.0 / (905.0 - 20 * torch.rand(B, hp, wp, generator=g))
    depth = nf1 + (nf0 - nf1) * torch.rand(B, hp, wp, generator=g)
    std = (nf0 - nf1) * (0.02 + clamp_share * torch.rand(B, hp, wp, generator=g))
    return depth, std, torch.stack([nf0, nf1], 1)


def _check_depth_values_bwd(lib, dev):
    g = torch.Generator().manual_seed(2)
    for depth_inv_level in (False, True):
        cas = EnerfConfig().with_cas(depth_inv=(True, depth_inv_level), volume_planes=(8, 8)).cas
        B, H, W = 2, 64, 96
        hp, wp = int(H * cas.volume_scale[0]), int(W * cas.volume_scale[0])
        depth, std, nf = _level_inputs(g, B, hp, wp)
        depth.requires_grad_(True); std.requires_grad_(True)
        batch = {"near_far": torch.tensor([[425.0, 905.0]] * B), "src_inps": torch.zeros(B, 3, 3, H, W)}
        dv, out_nf = TP.depth_values(cas, batch, 1, 8, depth, std, nf)
        gdv = torch.randn(dv.shape, generator=g)
        dv.backward(gdv)
        h, w = dv.shape[-2:]
        # forward twin: the inference kernel
        dv_k, nf_k = lib.get_depth_values(batch["near_far"].to(dev), (depth.detach().to(dev), std.detach().to(dev), nf.to(dev)), B, 8, h, w,
                                          depth_inv_level)
        assert _rel(dv_k, dv) < 1e-6 and _rel(nf_k, out_nf) < 1e-6
        gd, gs = lib.get_depth_values_bwd(depth.detach().to(dev), std.detach().to(dev), nf.to(dev), gdv.to(dev), depth_inv_level)
        clamped = float(((depth + std > nf[:, 0]) | (depth - std < nf[:, 1])).float().mean())
        assert clamped > 0.1                                            # the case exercises the masked branches
        assert _rel(gd, depth.grad) < 2e-5 and _rel(gs, std.grad) < 2e-5, (depth_inv_level, _rel(gd, depth.grad), _rel(gs, std.grad))


def _check_ray_samples(lib, dev):
    g = torch.Generator().manual_seed(3)
    for level, Ns, same in ((0, 8, False), (1, 2, False), (1, 1, False), (0, 4, True)):
        cas = EnerfConfig().cas
        depth_inv = cas.depth_inv[level]
        B, h, w = 2, 12, 20
        k = 1 if same else 2
        Hr, Wr = h * k, w * k
        if depth_inv:
            depth, std, nf = _level_inputs(g, B, h, w)
        else:
            nf0 = 425.0 + 50 * torch.rand(B, h, w, generator=g)
            nf1 = nf0 + 100 + 50 * torch.rand(B, h, w, generator=g)
            depth = nf0 + (nf1 - nf0) * torch.rand(B, h, w, generator=g)
            std = (nf1 - nf0) * (0.02 + 0.4 * torch.rand(B, h, w, generator=g))
            nf = torch.stack([nf0, nf1], 1)
        depth.requires_grad_(True); std.requires_grad_(True)
        N = 301
        uu = torch.randint(0, Wr, (B, N), generator=g).float()
        vv = torch.randint(0, Hr, (B, N), generator=g).float()
        rays8 = torch.cat([torch.randn(B, N, 3, generator=g) * 10, torch.randn(B, N, 3, generator=g), uu[..., None], vv[..., None]], -1)
        cas_l = EnerfConfig().with_cas(render_scale=(float(k) * cas.volume_scale[0], float(k) * cas.volume_scale[1])).cas
        rays12 = TP.build_rays(cas_l, depth, std, rays8, nf, level)
        xyz, uvd, z = TP.sample_along_depth(cas_l, rays12, Ns, level)
        gx, gd = torch.randn(xyz.shape, generator=g), torch.randn(z.shape, generator=g)
        (xyz * gx).sum().add((uvd[..., 2] * gd).sum()).backward()
        a = lambda t: t.detach().to(dev).contiguous()
        zk, xk, dnk, uvk, r12 = lib.ray_samples_fwd(a(rays8), a(depth), a(std), a(nf), Ns, Hr, Wr, depth_inv, want_rays12=True)
        assert _rel(r12, rays12) < 1e-6 and _rel(zk, z) < 1e-6 and _rel(xk, xyz) < 2e-6 and _rel(dnk, uvd[..., 2]) < 1e-5
        assert torch.equal(uvk.cpu(), uvd[..., :2].contiguous())
        gdk, gsk = lib.ray_samples_bwd(a(rays8), a(depth), a(std), a(nf), a(gx), a(gd), Ns, Hr, Wr, depth_inv)
        assert _rel(gdk, depth.grad) < 5e-5 and _rel(gsk, std.grad) < 5e-5, (level, Ns, same, _rel(gdk, depth.grad), _rel(gsk, std.grad))


def _check_camera_tables_and_layout(lib, dev):
    from torch_twins import gather_cameras as gather_cameras_torch
    from enerf_amd.synth import make_batch
    cfg = EnerfConfig()
    b = {k: torch.from_numpy(v) for k, v in make_batch(32, 64, 3, cfg, seed=5, B=2).items()}
    for rs in (0.25, 1.0):
        cam_ref, tcen_ref = gather_cameras_torch(b, rs)
        cam, tcen = lib.camera_tables(b["src_ixts"].to(dev), b["src_exts"].to(dev), b["tar_ext"].to(dev), rs)
        assert _rel(cam, cam_ref) < 1e-6 and _rel(tcen, tcen_ref) < 1e-6
    g = torch.Generator().manual_seed(4)
    w = torch.randn(6, 5, 3, 3, 3, generator=g)
    assert torch.equal(lib.weights_flip_transpose(w.to(dev)).cpu(), w.flip(2, 3, 4).transpose(0, 1).contiguous())
    w2 = torch.randn(7, 4, 5, 5, generator=g)
    assert torch.equal(lib.weights_flip_transpose(w2.to(dev)).cpu(), w2.flip(2, 3).transpose(0, 1).contiguous())
    a, c = torch.randn(8, 8, 27, generator=g), torch.randn(1, 8, 27, generator=g)
    out = lib.concat2_pad(a.to(dev), c.to(dev), 16 * 8 * 27).cpu().view(16, 8, 27)
    assert torch.equal(out, torch.cat([a, c, torch.zeros(7, 8, 27)], 0))
    feat = torch.randn(3, 16, 24, 8, generator=g)
    src = torch.rand(3, 3, 64, 96, generator=g) * 2 - 1
    for Hr, Wr in ((16, 24), ):
        tex = lib.pack_texels_train(feat.to(dev), src.to(dev), Hr, Wr).cpu()
        rgb = TP._resize_ac(src * 0.5 + 0.5, Hr / 64, True).permute(0, 2, 3, 1)
        assert torch.equal(tex[..., :8], feat) and _rel(tex[..., 8:], rgb) < 1e-5       # (FMA contraction on the GPU: 3e-6)
    feat1 = torch.randn(2, 64, 96, 8, generator=g)
    tex = lib.pack_texels_train(feat1.to(dev), src[:2].to(dev), 64, 96).cpu()
    assert torch.equal(tex[..., 8:], (src[:2] * 0.5 + 0.5).permute(0, 2, 3, 1)) and torch.equal(tex[..., :8], feat1)
    assert torch.equal(lib.slice_channels(tex.to(dev), 0, 8).cpu(), feat1)
    assert torch.equal(lib.slice_channels(tex.to(dev), 8, 3).cpu(), tex[..., 8:].contiguous())
    srcs = [torch.randn(n, generator=g) for n in (10, 33, 7)]
    which = torch.randint(0, 3, (200,), generator=g).int()
    idx = torch.stack([torch.randint(-1, srcs[int(q)].numel(), (1,), generator=g)[0] for q in which]).int()
    ref = torch.stack([srcs[int(q)][int(i)] if i >= 0 else torch.tensor(0.0) for q, i in zip(which, idx)])
    got = lib.gather_images([t.to(dev) for t in srcs], which.to(dev), idx.to(dev)).cpu()
    assert torch.equal(got, ref)
    x, y = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    assert torch.equal(lib.add(x.to(dev), y.to(dev)).cpu(), x + y)


def _check_gather_bwd_tiled(lib, dev):
    """enerf_gather_bwd with the raster hints (a block per 2-D ray tile, texel / volume scatter accumulated in LDS patches) against
    the same entry without them (every contribution a global atomic; pinned to torch's grid_sample backward in
    tests/test_training.py): partial tiles in both directions, the one-patch-per-sample-index form of the coarse level, taps that
    do not fit a patch (random depths: the projections of a tile cover the whole source image), and a WRONG hint (permuted
    points) — all the same sums."""
    from torch_twins import gather_cameras as gather_cameras_torch
    from enerf_amd.synth import make_batch
    g = torch.Generator().manual_seed(9)
    for Hr, Wr, Ns, Fc, wild, permute in ((20, 44, 2, 11, False, False), (12, 40, 4, 35, False, False), (9, 35, 2, 11, True, False),
                                          (16, 32, 1, 11, False, True)):
        cfg = EnerfConfig().with_cas(render_scale=(1.0, 1.0))
        b = {k: torch.from_numpy(v) for k, v in make_batch(Hr, Wr, 3, cfg, seed=3 + Ns, B=2).items()}
        rays = b["rays_1"]                                               # (B, Hr*Wr, 8+): row-major full-image rays
        B, N = rays.shape[0], rays.shape[1]
        assert N == Hr * Wr
        near, far = float(b["near_far"].min()), float(b["near_far"].max())
        if wild:
            t = near + (far - near) * torch.rand(B, N, Ns, generator=g) * 3.0 - (far - near)
        else:                                                            # a smooth depth map + the samples of a ray close together
            yy, xx = torch.meshgrid(torch.linspace(0, 1, Hr), torch.linspace(0, 1, Wr), indexing="ij")
            base = near + (far - near) * (0.3 + 0.4 * torch.sin(3 * xx + 2 * yy).abs()).reshape(1, N, 1)
            t = base + (far - near) * 0.02 * torch.arange(Ns).reshape(1, 1, Ns) + (far - near) * 0.005 * torch.rand(B, N, Ns, generator=g)
        xyz = (rays[:, :, None, :3] + rays[:, :, None, 3:6] * t[..., None]).reshape(B, N * Ns, 3).contiguous()
        dn = (torch.rand(B, N * Ns, generator=g) * 1.2 - 0.1).contiguous()
        uv = rays[:, :, None, 6:8].expand(B, N, Ns, 2).reshape(B, N * Ns, 2).contiguous()
        if permute:
            perm = torch.randperm(N * Ns, generator=g)
            xyz, dn, uv = xyz[:, perm].contiguous(), dn[:, perm].contiguous(), uv[:, perm].contiguous()
        tex = torch.randn(B, 3, Hr, Wr, Fc, generator=g)
        vol = torch.randn(B, 8, max(Hr // 2, 2), max(Wr // 2, 2), 8, generator=g)
        gx, gv = torch.randn(B, N * Ns, 3, Fc + 4, generator=g), torch.randn(B, N * Ns, 8, generator=g)
        cam, tcen = gather_cameras_torch(b, 1.0)
        T = lambda *ts: [t_.to(dev).contiguous() for t_ in ts]
        args = T(xyz, dn, uv, tex, vol, cam, tcen, gx, gv)
        ref = lib.gather_bwd(*args)
        got = lib.gather_bwd(*args, n_samples=Ns, ray_w=Wr)
        for name, r, o in zip(("tex", "vol", "xyz", "dn"), ref, got):
            assert float(r.abs().max()) > 0
            assert _rel(o, r) < 2e-5, (Hr, Wr, Ns, Fc, wild, permute, name, _rel(o, r))


def _check_round4_kernel_pairs(lib, dev):
    """Round-4 kernel pairs that must agree with the form they replace.
    * the warp kernel with two depth planes per wave (even D) and the one-plane kernel (odd D): the same planes, bit for bit;
    * enerf_channel_sums with scratch (partial rows + finish launch) and without (fp64 atomics): the same sums up to the fp64
      summation order, for every channel width, ragged position counts, all three input forms;
    * the wave-per-64-points gather forward on point counts that are not a multiple of 64 and straddle batch elements."""
    from torch_twins import gather_cameras as gather_cameras_torch
    from enerf_amd.synth import make_batch
    g = torch.Generator().manual_seed(21)
    # ---- warp: D = 6 (two planes per wave) vs its first five planes as a D = 5 volume (one plane per wave) ----
    cfg = EnerfConfig()
    b = {k: torch.from_numpy(v) for k, v in make_batch(32, 64, 3, cfg, seed=2, B=2).items()}
    for Cc, (h, w) in ((32, (9, 14)), (16, (13, 22)), (8, (8, 16))):
        feat = torch.randn(2, 3, 16, 32, Cc, generator=g)
        P = lib.get_proj_mats(b["src_ixts"].to(dev), b["src_exts"].to(dev), b["tar_ixt"].to(dev), b["tar_ext"].to(dev), 16 / 32, h / 32).cpu()
        dv6 = 425.0 + 480.0 * torch.rand(2, 6, h, w, generator=g)
        v6 = lib.build_feature_volume(feat.to(dev), P.to(dev), dv6.to(dev), Cc).cpu()
        v5 = lib.build_feature_volume(feat.to(dev), P.to(dev), dv6[:, :5].contiguous().to(dev), Cc).cpu()
        assert torch.equal(v6[:, :5], v5), Cc
    # ---- BatchNorm statistics: scratch form vs atomic form ----
    for n, Cc in ((1000, 8), (4099, 16), (777, 32), (1, 64), (30000, 4)):
        a_, b_, z_ = (torch.randn(n, Cc, generator=g).to(dev) for _ in range(3))
        ms, mh = torch.rand(Cc, generator=g).to(dev) + 0.5, torch.randn(Cc, generator=g).to(dev) * 0.1
        for args in ((a_, a_, None, None, None), (a_, b_, None, None, None), (a_, b_, z_, ms, mh)):
            new = lib.channel_sums_raw(*args).cpu()
            old = torch.empty((2, Cc), dtype=torch.float64, device=dev)
            p_ = lambda t: None if t is None else t.data_ptr()
            lib._check(lib.dll.enerf_channel_sums(p_(args[0]), p_(args[1]), p_(args[2]), p_(args[3]), p_(args[4]), n, Cc, old.data_ptr(),
                                                  lib.stream_of(a_)), "channel_sums")
            old = old.cpu()
            m = (z_.cpu().double() * ms.cpu().double() + mh.cpu().double() > 0) if args[2] is not None else torch.ones(n, Cc, dtype=torch.bool)
            # (the kernels evaluate the mask in fp32: build the reference from the same comparison)
            if args[2] is not None:
                m = (z_.cpu() * ms.cpu() + mh.cpu()) > 0
            ref = torch.stack([(args[0].cpu().double() * m).sum(0), (args[0].cpu().double() * m * args[1].cpu().double()).sum(0)])
            scale = float(ref.abs().max()) + 1e-30
            assert float((new - ref).abs().max()) <= 1e-12 * scale * n and float((old - ref).abs().max()) <= 1e-12 * scale * n, (n, Cc)
    # ---- gather forward: B * P = 2 * 77 points (three waves, the middle one straddles the batch elements, the last is ragged) ----
    Hr, Wr, Fc = 12, 20, 11
    cfg1 = EnerfConfig().with_cas(render_scale=(1.0, 1.0))
    bb = {k: torch.from_numpy(v) for k, v in make_batch(Hr, Wr, 3, cfg1, seed=5, B=2).items()}
    rays = bb["rays_1"][:, :77]
    t = 500.0 + 300.0 * torch.rand(2, 77, 1, generator=g)
    xyz = (rays[..., :3] + rays[..., 3:6] * t).contiguous()
    dn, uv = torch.rand(2, 77, generator=g), rays[..., 6:8].contiguous()
    tex, vol = torch.randn(2, 3, Hr, Wr, Fc, generator=g), torch.randn(2, 4, 6, 10, 8, generator=g)
    cam, tcen = gather_cameras_torch(bb, 1.0)
    T = lambda *ts: [t_.to(dev).contiguous() for t_ in ts]
    x, vox = lib.gather_fwd(*T(xyz, dn, uv, tex, vol, cam, tcen))
    # per batch element on its own (one wave each, no straddling): the same rows
    for e in range(2):
        xe, ve = lib.gather_fwd(*T(xyz[e:e + 1], dn[e:e + 1], uv[e:e + 1], tex[e:e + 1], vol[e:e + 1], cam[e:e + 1], tcen[e:e + 1]))
        assert torch.equal(x[e].cpu(), xe[0].cpu()) and torch.equal(vox[e].cpu(), ve[0].cpu()), e
    nd = torch.stack([uv[..., 0] / (Wr - 1), uv[..., 1] / (Hr - 1), dn], -1).reshape(2, 1, 1, 77, 3) * 2.0 - 1.0
    v_ref = F.grid_sample(vol.permute(0, 4, 1, 2, 3), nd, align_corners=True)[:, :, 0, 0].permute(0, 2, 1)
    assert _rel(vox, v_ref) < 1e-5


CHECKS = [_check_round4_kernel_pairs, _check_gather_bwd_tiled, _check_s2k5_dgrad, _check_resize_adjoint, _check_depth_values_bwd, _check_ray_samples, _check_camera_tables_and_layout]


@pytest.mark.parametrize("check", CHECKS, ids=lambda f: f.__name__[7:])
def test_train_glue_emulated(check):
    from emu_lib import emu_lib
    check(emu_lib(), torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("check", CHECKS, ids=lambda f: f.__name__[7:])
def test_train_glue_gpu(check):
    from enerf_amd.lib import get_lib
    check(get_lib(), torch.device("cuda:0"))
