"""SURVEY.md §8f rows 3 (device-side ray generation) and 4 (uint8 presentation pack, evaluator statistics):
checked against numpy restatements of lib/datasets/enerf_utils.py:61-71, gui_human.py:88-91 and
lib/evaluators/enerf.py:67-71,88-103.  Runs on the CPU lane emulator here and on the GPU with -m gpu."""
import numpy as np
import pytest
import torch

from enerf_amd.config import EnerfConfig
from enerf_amd.synth import full_image_rays, make_batch


def _check(lib, dev):
    cfg = EnerfConfig()
    b = make_batch(64, 96, 3, cfg, seed=9, B=2)
    for scale in (1.0, 0.25):
        Hr, Wr = int(64 * scale), int(96 * scale)
        rays = lib.gen_rays(torch.from_numpy(b["tar_ext"]).to(dev), torch.from_numpy(b["tar_ixt"]).to(dev), Hr, Wr, scale)
        ref = np.stack([full_image_rays(b["tar_ext"][i].astype(np.float64), b["tar_ixt"][i].astype(np.float64), 64, 96, scale)
                        for i in range(2)])
        assert rays.shape == ref.shape
        np.testing.assert_allclose(rays.cpu().numpy(), ref, rtol=2e-6, atol=2e-6)
    # presentation pack
    rng = np.random.default_rng(0)
    rgb = rng.uniform(-0.05, 1.05, size=(40 * 56, 3)).astype(np.float32)
    got = lib.pack_rgb8(torch.from_numpy(rgb).to(dev), 40, 56, flip=True).cpu().numpy()
    ref8 = np.clip(rgb.reshape(40, 56, 3) * 255.0, 0, 255).astype(np.uint8)[::-1]
    assert got.dtype == np.uint8 and np.array_equal(got, ref8)
    # evaluator statistics
    P = 5000
    pred, gt = rng.uniform(0, 1, (P, 3)).astype(np.float32), rng.uniform(0, 1, (P, 3)).astype(np.float32)
    mask = (rng.uniform(size=P) > 0.3).astype(np.int32)
    pd = rng.uniform(400, 900, P).astype(np.float32)
    gd = (pd + rng.normal(0, 6, P)).astype(np.float32)
    gd[rng.uniform(size=P) < 0.2] = 0.0
    st = lib.eval_stats(torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev), torch.from_numpy(mask).to(dev),
                        torch.from_numpy(pd).to(dev), torch.from_numpy(gd).to(dev))
    m = mask == 1
    mse = np.mean((pred[m].astype(np.float64) - gt[m].astype(np.float64)) ** 2)
    assert st["psnr"] == pytest.approx(10 * np.log10(1.0 / mse), rel=1e-9)
    v = gd != 0
    err = np.abs(pd[v].astype(np.float64) - gd[v].astype(np.float64))
    assert st["abs"] == pytest.approx(err.mean(), rel=1e-9)
    assert st["acc_2"] == pytest.approx((err < 2).mean(), rel=1e-12) and st["acc_10"] == pytest.approx((err < 10).mean(), rel=1e-12)
    st2 = lib.eval_stats(torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev))
    assert st2["psnr"] == pytest.approx(10 * np.log10(1.0 / np.mean((pred.astype(np.float64) - gt) ** 2)), rel=1e-9)


def test_io_rows_emulated():
    from emu_lib import emu_lib
    _check(emu_lib(), torch.device("cpu"))


_needs_gpu = pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")


@pytest.mark.gpu
@_needs_gpu
def test_io_rows_gpu():
    from enerf_amd.lib import get_lib
    _check(get_lib(), torch.device("cuda:0"))


@pytest.mark.gpu
@_needs_gpu
def test_device_rays_render_identically():
    """A frame rendered from device-generated rays equals the frame rendered from the host-built rays."""
    from enerf_amd.lib import get_lib
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
    batch["rays_1"] = get_lib().gen_rays(batch["tar_ext"], batch["tar_ixt"], 128, 160, 1.0)
    out = net(batch)["rgb_level1"]
    assert float((out - ref).abs().max()) < 1e-5
