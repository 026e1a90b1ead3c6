"""SURVEY.md §8f rows 3 (device-side ray generation, bbox mask, view selection) and 4 (uint8 presentation pack, evaluator
statistics).  PINNED to the reference: ``tests/golden/io_rays.npz`` holds outputs of the unmodified
``lib/datasets/enerf_utils.build_rays`` (test and train splits) and ``lib/utils/net_utils.gen_rays_bbox``
(oracle/make_golden_io.py); the pack is compared with the three torch calls of gui_human.py:88-91 themselves; the
evaluator statistics with the numpy / skimage expressions of lib/evaluators/enerf.py:67-71,88-103 (skimage's psnr is
``10*log10(1/mean((a-b)**2))`` in float64; skimage itself is not installable here).  Runs on the CPU lane emulator here
and on the GPU with -m gpu."""
import os

import numpy as np
import pytest
import torch

from enerf_amd.config import EnerfConfig
from enerf_amd.synth import make_batch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io_rays.npz")


def _check_rays(lib, dev):
    g = np.load(GOLD)
    H, W = int(g["in/H"]), int(g["in/W"])
    ext = torch.from_numpy(g["in/tar_ext"])[None].to(dev)
    ixt = torch.from_numpy(g["in/tar_ixt"])[None].to(dev)
    for level, scale in enumerate(g["meta/scales"]):
        Hr, Wr = int(H * scale), int(W * scale)
        ref = g[f"test/rays_{level}"]                                   # enerf_utils.py:61-71, float64 -> float32
        rays = lib.gen_rays(ext, ixt, Hr, Wr, float(scale))
        assert rays.shape == (1,) + ref.shape
        np.testing.assert_allclose(rays[0].cpu().numpy(), ref, rtol=2e-6, atol=2e-6)
        # training branch (enerf_utils.py:33-56): the pixel list is the reference RNG's, the rays are built on device
        tr = g[f"train/rays_{level}"]
        xy = torch.from_numpy(np.ascontiguousarray(tr[:, 6:8]).astype(np.int32))[None].to(dev)
        got = lib.gen_rays_at(ext, ixt, xy, float(scale))
        np.testing.assert_allclose(got[0].cpu().numpy(), tr, rtol=2e-6, atol=2e-6)
    # gen_rays_bbox (net_utils.py:13-28): bit-exact int mask
    rays1 = torch.from_numpy(g["test/rays_1"]).to(dev)
    m = lib.rays_bbox_mask(rays1, torch.from_numpy(g["bbox/bounds"]).to(dev))
    assert np.array_equal(m.cpu().numpy(), g["bbox/mask"])


def _check_views(lib, dev):
    rng = np.random.default_rng(5)
    V, H, W, k = 21, 24, 40, 4
    cams = rng.normal(0, 2.0, (V, 3)).astype(np.float32)
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 3] = rng.normal(0, 2.0, 3)
    near = np.argsort(np.linalg.norm(cams - c2w[:3, 3][None], axis=-1))[:k]       # enerf_interactive.py:207-210
    idx = lib.select_views(torch.from_numpy(cams).to(dev), torch.from_numpy(c2w).to(dev), k)
    assert np.array_equal(idx.cpu().numpy(), near.astype(np.int32))
    inps = rng.uniform(-1, 1, (V, H, W, 3)).astype(np.float32)
    exts = rng.normal(size=(V, 4, 4)).astype(np.float32)
    ixts = rng.normal(size=(V, 3, 3)).astype(np.float32)
    si, se, sk = lib.gather_views(torch.from_numpy(inps).to(dev), torch.from_numpy(exts).to(dev),
                                  torch.from_numpy(ixts).to(dev), idx)
    assert np.array_equal(si.cpu().numpy(), inps[near].transpose(0, 3, 1, 2))       # :215 permute(0,3,1,2)
    assert np.array_equal(se.cpu().numpy(), exts[near]) and np.array_equal(sk.cpu().numpy(), ixts[near])


def _check_pack_and_stats(lib, dev):
    rng = np.random.default_rng(0)
    rgb = rng.uniform(0, 1, size=(40 * 56, 3)).astype(np.float32)
    rgb[:7] = np.array([0.0, 1.0, 0.5, 1 / 255, 254.999 / 255, 0.999999, 2 / 255], np.float32)[:, None]
    got = lib.pack_rgb8(torch.from_numpy(rgb).to(dev), 40, 56, flip=True).cpu()
    img = torch.from_numpy(rgb.copy()).reshape(40, 56, 3)
    img *= 255                                                          # gui_human.py:88-91, verbatim
    img = img.to(torch.uint8)
    img = torch.flip(img, (0,))
    assert got.dtype == torch.uint8 and torch.equal(got, img)
    # out-of-range policy (the reference's cast is undefined there): saturate
    oob = torch.tensor([[-0.2, 1.7, float("nan")]], dtype=torch.float32).to(dev)
    assert lib.pack_rgb8(oob, 1, 1, flip=False).cpu().reshape(-1).tolist() == [0, 255, 0]
    # evaluator statistics
    h, w = 50, 100
    P = h * w
    pred, gt = rng.uniform(0, 1, (P, 3)).astype(np.float32), rng.uniform(0, 1, (P, 3)).astype(np.float32)
    msk = (rng.uniform(size=P) > 0.3).astype(np.uint8) * rng.integers(1, 3, P).astype(np.uint8)   # values 0, 1, 2
    pd = rng.uniform(400, 900, P).astype(np.float32)
    gd = (pd + rng.normal(0, 6, P)).astype(np.float32)
    gd[rng.uniform(size=P) < 0.2] = 0.0
    T = lambda a: torch.from_numpy(a).to(dev)
    for mask_t in (T(msk), T(msk.astype(np.int32))):
        st = lib.eval_stats(T(pred), T(gt), mask_t, T(pd), T(gd))
        m = msk >= 1                                                    # evaluators/enerf.py:48
        mse = np.mean((pred[m].astype(np.float64) - gt[m].astype(np.float64)) ** 2)
        assert st["psnr"] == pytest.approx(10 * np.log10(1.0 / mse), rel=1e-9)
        v = gd != 0.                                                    # :94
        err = np.abs(pd[v] - gd[v])                                     # float32, as in :96-98
        assert st["abs"] == pytest.approx(float(err.astype(np.float64).mean()), rel=1e-9)
        assert st["acc_2"] == (err < 2).mean() and st["acc_10"] == (err < 10).mean()
    # eval_center crop (:50-54)
    ch, cw = int(h * 0.1), int(w * 0.1)
    st = lib.eval_stats(T(pred), T(gt), T(msk), image_hw=(h, w), crop=(ch, cw))
    P3 = lambda a: a.reshape(h, w, -1)[ch:-ch, cw:-cw]
    mc = P3(msk)[..., 0] >= 1
    mse = np.mean((P3(pred)[mc].astype(np.float64) - P3(gt)[mc].astype(np.float64)) ** 2)
    assert st["psnr"] == pytest.approx(10 * np.log10(1.0 / mse), rel=1e-9)
    st2 = lib.eval_stats(T(pred), T(gt))
    assert st2["psnr"] == pytest.approx(10 * np.log10(1.0 / np.mean((pred.astype(np.float64) - gt) ** 2)), rel=1e-9)


def _check(lib, dev):
    _check_rays(lib, dev)
    _check_views(lib, dev)
    _check_pack_and_stats(lib, dev)


def test_io_rows_emulated():
    from emu_lib import emu_lib
    _check(emu_lib(), torch.device("cpu"))


def test_device_evaluator_emulated():
    """enerf_amd.evaluator.DeviceEvaluator: the evaluate()/summarize() surface of lib/evaluators/enerf.py on a rendered frame."""
    from emu_lib import emu_lib
    from enerf_amd.evaluator import DeviceEvaluator
    from enerf_amd.network import Network
    from golden_cases import case_batch, case_config, load_weights
    cfg, batch = case_config("tiny_s3"), case_batch("tiny_s3")
    net = Network(cfg, lib=emu_lib()).eval()
    net.load_state_dict(load_weights(), strict=False)
    out = net(batch)
    rng = np.random.default_rng(1)
    for i, (h, w) in enumerate(((8, 16), (32, 64))):
        batch[f"rgb_{i}"] = torch.from_numpy(rng.uniform(0, 1, (1, h * w, 3)).astype(np.float32))
        batch[f"msk_{i}"] = torch.from_numpy((rng.uniform(size=(1, h * w)) > 0.2).astype(np.uint8))
    batch["tar_dpt"] = out["depth_level1"].reshape(1, 32, 64) + 1.5
    ev = DeviceEvaluator(cfg, eval_depth=True, lib=emu_lib())
    ev.evaluate(out, batch)
    s = ev.summarize()
    m = batch["msk_1"][0].numpy() >= 1
    mse = np.mean((out["rgb_level1"][0].numpy()[m].astype(np.float64) - batch["rgb_1"][0].numpy()[m]) ** 2)
    assert s["psnr"] == pytest.approx(10 * np.log10(1 / mse), rel=1e-9)
    assert s["abs"] == pytest.approx(1.5, rel=1e-5) and s["acc_2"] == 1.0 and "psnr_level0" in s
    # MVS-depth statistics (evaluators/enerf.py:91-103): depth_mvs_level1 (16x32) against tar_dpt resized to its resolution with
    # cv2.INTER_NEAREST = rows/cols min(floor(d * src/dst), src-1) -> here every second row / column; some pixels without gt
    gt = batch["tar_dpt"][0].clone()
    gt[::4, ::2] = 0.0                                                 # gt == 0: excluded (:94-95)
    batch["tar_dpt"] = gt[None]
    ev.evaluate(out, batch)
    s = ev.summarize()
    mvs = out["depth_mvs_level1"][0].numpy()
    assert mvs.shape == (16, 32)
    g = gt.numpy()[(np.floor(np.arange(16) * (1.0 / (16 / 32)))).astype(int)][:, (np.floor(np.arange(32) * (1.0 / (32 / 64)))).astype(int)]
    mk = g != 0.0
    e = np.abs(mvs[mk] - g[mk])
    assert mk.sum() == 16 * 32 // 2 + 0 or mk.sum() > 0
    assert s["mvs_abs"] == pytest.approx(float(e.mean()), rel=1e-6)
    assert s["mvs_acc_2"] == pytest.approx(float((e < 2).mean()), rel=1e-9)
    assert s["mvs_acc_10"] == pytest.approx(float((e < 10).mean()), rel=1e-9)
    from enerf_amd.evaluator import nearest_resize_index
    assert nearest_resize_index(7, 3, "cpu").tolist() == [0, 2, 4] and nearest_resize_index(5, 5, "cpu").tolist() == [0, 1, 2, 3, 4]
    assert nearest_resize_index(3, 7, "cpu").tolist() == [0, 0, 0, 1, 1, 2, 2]


_needs_gpu = pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")


@pytest.mark.gpu
@_needs_gpu
def test_io_rows_gpu():
    from enerf_amd.lib import get_lib
    _check(get_lib(), torch.device("cuda:0"))


@pytest.mark.gpu
@_needs_gpu
def test_device_rays_render_identically():
    """A frame rendered from device-generated rays (no rays_i in the batch: enerf_forward generates them) equals the frame
    rendered from the host-built rays."""
    from enerf_amd.network import Network
    from golden_cases import load_weights
    dev = torch.device("cuda:0")
    cfg = EnerfConfig.dtu_eval().with_cas(volume_planes=(16, 8))
    b = make_batch(128, 160, 3, cfg, seed=2, textured=True)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
    net = Network(cfg)
    net.load_state_dict(load_weights(), strict=False)
    net = net.to(dev).eval()
    ref = net(batch)["rgb_level1"].clone()
    del batch["rays_1"]
    out = net(batch)["rgb_level1"]
    assert float((out - ref).abs().max()) < 1e-5
