"""CPU checks of the *actual kernel sources* (compiled against the lane emulator) and of the host
logic (Network, ctypes binding, C-ABI argument checks) against reference goldens and the oracle."""
import numpy as np
import pytest
import torch

from enerf_amd.config import EnerfConfig
from enerf_amd.lib import EnerfError
from enerf_amd.network import Network, NetworkHuman
from enerf_amd.synth import make_batch
from oracle import enerf_oracle as O
from emu_lib import emu_lib
from golden_cases import CASES, case_batch, case_config, load_golden, load_weights


def _net(cfg, human=False):
    net = (NetworkHuman if human else Network)(cfg, lib=emu_lib()).eval()
    net.load_state_dict(load_weights(), strict=False)
    return net


def _close(a, ref, tol, name=""):
    a, ref = np.asarray(a), np.asarray(ref)
    scale = max(np.abs(ref).max(), 1e-12)
    err = np.abs(a - ref).max() / scale
    assert err < tol, f"{name}: max err / max|ref| = {err:.3e} (tol {tol})"


@pytest.mark.parametrize("name", list(CASES))
def test_emulated_kernels_match_reference_goldens(name):
    cfg, batch, gold = case_config(name), case_batch(name), load_golden(name)
    out = _net(cfg, CASES[name]["human"])(batch)
    assert sorted(out) == sorted(k[4:] for k in gold if k.startswith("out/"))
    for k, v in out.items():
        assert v.shape == gold["out/" + k].shape, k
        _close(v.numpy(), gold["out/" + k], 2e-5, k)


def test_render_precision_variants_match_reference_goldens():
    """enerf_options_t.render_precision: default = exact fp32 MFMAs; 2 = bf16x3 (every operand as two bf16 pieces on the bf16
    matrix cores, ~1e-5), 3 = bf16x6 (three pieces: fp32-level accuracy): all against the reference's outputs; bf16x6 must be
    as close as the exact kernel (within fp32 re-association noise), bf16x3 within 2e-5."""
    from enerf_amd.lib import Options
    for name in ("tiny_s3", "tiny_s4_mask"):
        cfg, batch, gold = case_config(name), case_batch(name), load_golden(name)
        net = _net(cfg, CASES[name]["human"])
        err = {}
        for tag, opt in (("fp32", None), ("bf16x3", Options(render_precision=2)), ("bf16x6", Options(render_precision=3))):
            net.options = opt
            out = net(batch)
            for k, v in out.items():
                _close(v.numpy(), gold["out/" + k], 2e-5, k)
            err[tag] = float(np.abs(out["rgb_level1"].numpy() - gold["out/rgb_level1"]).max()) / float(np.abs(gold["out/rgb_level1"]).max())
        print(name, err)
        assert err["bf16x6"] <= max(3.0 * err["fp32"], 2e-6) and err["fp32"] <= err["bf16x3"] < 2e-5, err
        # the frame's cached static fields are keyed on the option VALUES: an Options mutated in place takes effect, a fresh but
        # equal one is the same frame
        net.options = Options(render_precision=2)
        a = net(batch)["rgb_level1"].clone()
        net.options.render_precision = 0
        b = net(batch)["rgb_level1"].clone()
        net.options = None
        c = net(batch)["rgb_level1"].clone()
        net.options = Options(render_precision=2)
        d = net(batch)["rgb_level1"].clone()
        assert torch.equal(b, c) and torch.equal(a, d) and not torch.equal(a, b)


def test_state_dict_names_match_reference():
    sd = load_weights()
    net = Network(EnerfConfig())
    mine = {k for k in net.state_dict() if not k.endswith("num_batches_tracked")}
    assert mine == set(sd), sorted(mine ^ set(sd))[:10]
    for k, v in net.state_dict().items():
        if k in sd:
            assert tuple(v.shape) == tuple(sd[k].shape), k
    net.load_state_dict(sd, strict=False)


def test_batch2_ragged_rays_and_white_bkgd():
    """B=2, a ray list whose length is not a multiple of the 16-ray tile, white_bkgd quirk, vs oracle."""
    cfg = EnerfConfig(white_bkgd=True).with_cas(volume_planes=(8, 8))
    b = make_batch(32, 64, 3, cfg, seed=11, B=2, textured=True)
    keep = np.random.default_rng(0).permutation(32 * 64)[:1003]
    b["rays_1"] = np.ascontiguousarray(b["rays_1"][:, keep])
    b["rays_0"] = np.ascontiguousarray(b["rays_0"][:, :77])
    batch = {k: torch.from_numpy(v) for k, v in b.items()}
    out = _net(cfg)(batch)
    with torch.no_grad():
        ref = O.forward(cfg, load_weights(), batch)
    for k in ref:
        _close(out[k].numpy(), ref[k].numpy(), 5e-5, k)
    # the same two-element batch through the LDS-staged / batched-4x4 convolution kernels (the default routes volumes this
    # small to the global-load kernels): box decomposition with B = 2
    from enerf_amd.lib import Options
    net = _net(cfg)
    net.options = Options(conv3d_lds_min_voxels=1)
    out = net(batch)
    for k in ref:
        _close(out[k].numpy(), ref[k].numpy(), 5e-5, k)
    # (that default routes conv0 / the heads through the asynchronously staged kernels — global_load_lds, channel-quad-planar
    # cost volume and conv11 output — and the transposed layers through the every-class kernels); the round-2 kernels, B = 2
    net.options = Options(conv3d_lds_min_voxels=1, conv3d_b4=3, conv3d_t2_variant=1)
    out = net(batch)
    for k in ref:
        _close(out[k].numpy(), ref[k].numpy(), 5e-5, k)


def test_render_rays_surface_accepts_reference_volume_layout():
    """render_rays(rays, level=, batch=, im_feat=, feature_volume=(B,8,D,h,w), nerf_model=) like network.py:24."""
    name = "tiny_s3"
    cfg, batch, g = case_config(name), case_batch(name), load_golden(name)
    net = _net(cfg)
    feats = {"level_2": torch.from_numpy(g["mid/feat_l2"]).reshape(1, 3, 8, 32, 64)}
    out = net.render_rays(torch.from_numpy(g["mid/rays12_1"]), level=1, batch=batch, im_feat=feats["level_2"],
                          feature_volume=torch.from_numpy(g["mid/feat3d_1"]), nerf_model=net.nerf_1)
    _close(out["rgb"].numpy(), g["out/rgb_level1"], 1e-5, "rgb")
    _close(out["depth"].numpy(), g["out/depth_level1"], 1e-5, "depth")
    _close(out["weights"].numpy(), g["out/weights_level1"], 1e-5, "weights")
    # chunked == unchunked (batchify_rays, network.py:45-55)
    net.cfg = EnerfConfig(cas=cfg.cas, chunk_size=500)
    out2 = net.batchify_rays(torch.from_numpy(g["mid/rays12_1"]), level=1, batch=batch, im_feat=feats["level_2"],
                             feature_volume=torch.from_numpy(g["mid/feat3d_1"]), nerf_model=net.nerf_1)
    assert torch.equal(out2["rgb"], out["rgb"])


def test_stage_kernels_against_reference_intermediates():
    """Every C-ABI stage fed with the reference's own upstream tensors."""
    name = "tiny_s3"
    cfg, batch, g = case_config(name), case_batch(name), load_golden(name)
    lib, cas = emu_lib(), cfg.cas
    T = lambda k: torch.from_numpy(g["mid/" + k]).contiguous()
    net = _net(cfg)
    prev = None
    for i in range(2):
        P = lib.get_proj_mats(batch["src_ixts"], batch["src_exts"], batch["tar_ixt"], batch["tar_ext"],
                              cas.im_feat_scale[i], cas.volume_scale[i])
        _close(P.numpy(), g[f"mid/proj_{i}"], 1e-6, f"proj_{i}")
        D = cas.volume_planes[i]
        h, w = g[f"mid/dv_{i}"].shape[-2:]
        dv, nf = lib.get_depth_values(batch["near_far"], prev, 1, D, h, w, cas.depth_inv[i])
        _close(dv.numpy(), g[f"mid/dv_{i}"], 1e-6, f"dv_{i}")
        _close(nf.numpy(), g[f"mid/nf_{i}"], 1e-6, f"nf_{i}")
        f = T(f"feat_l{i}")                                                # (S, C, Hs, Ws)
        S, C, Hs, Ws = f.shape
        fcl = lib.channels_last(f.reshape(S, C, Hs * Ws), S, C, Hs * Ws).view(1, S, Hs, Ws, C)
        vol = lib.build_feature_volume(fcl, T(f"proj_{i}"), T(f"dv_{i}"), C)
        ref_vol = T(f"vol_{i}")                                            # (1,C,D,h,w)
        _close(vol.permute(0, 4, 1, 2, 3).numpy(), ref_vol.numpy(), 1e-5, f"vol_{i}")
        vin = lib.channels_last(ref_vol.reshape(1, C, -1), 1, C, D * h * w).view(1, D, h, w, C)
        m = getattr(net, f"cost_reg_{i}")
        feat, prob = lib.cost_reg(net._packed_weights(f"cost_reg_{i}"), m.in_channels, m.full, vin)
        _close(feat.permute(0, 4, 1, 2, 3).numpy(), g[f"mid/feat3d_{i}"], 1e-5, f"feat3d_{i}")
        _close(prob.numpy(), g[f"mid/prob_{i}"], 1e-5, f"prob_{i}")
        d, s = lib.depth_regression(T(f"prob_{i}"), T(f"dv_{i}"), cas.depth_inv[i])
        _close(d.numpy(), g[f"mid/depth_{i}"], 1e-6, f"depth_{i}")
        _close(s.numpy(), g[f"mid/std_{i}"], 1e-5, f"std_{i}")
        Hr, Wr = int(32 * cas.render_scale[i]), int(64 * cas.render_scale[i])
        r = lib.build_rays(batch[f"rays_{i}"], T(f"depth_{i}"), T(f"std_{i}"), T(f"nf_{i}"), Hr, Wr, cas.depth_inv[i])
        _close(r.numpy(), g[f"mid/rays12_{i}"], 1e-6, f"rays12_{i}")
        prev = (T(f"depth_{i}"), T(f"std_{i}"), T(f"nf_{i}"))


def test_capi_rejects_bad_arguments():
    lib = emu_lib()
    with pytest.raises(EnerfError, match="unsupported"):
        lib.build_feature_volume(torch.zeros(1, 3, 8, 8, 12), torch.zeros(1, 3, 3, 4), torch.ones(1, 4, 4, 4), 12)
    with pytest.raises(EnerfError, match="divisible"):
        net = _net(case_config("tiny_s3"))
        lib.cost_reg(net._packed_weights("cost_reg_1"), 16, True, torch.zeros(1, 4, 8, 8, 16))
    with pytest.raises(EnerfError, match="contiguous float32"):
        lib.depth_regression(torch.zeros(1, 4, 4, 4, dtype=torch.float64), torch.ones(1, 4, 4, 4), True)


def test_frame_sizes_the_reference_cannot_run_fail_loudly():
    """With the reference's cascade (volume_scale 0.125 / 0.5) only image sizes that are multiples of 32 survive its own U-Net
    skip additions (cost_reg_net.py:41-44, 81-84: a 7-row volume comes back from conv11 with 8 rows): the reference raises
    a size-mismatch RuntimeError there (restated by the oracle), and the HIP path must raise too — never render garbage.
    (This is also why every reference-generated golden has a multiple-of-32 size: int(H*scale) never floors for them.)"""
    from enerf_amd.config import EnerfConfig
    from enerf_amd.synth import make_batch
    from oracle import enerf_oracle as O
    cfg = EnerfConfig().with_cas(volume_planes=(8, 8), render_if=(True, True))
    for H, W in ((60, 100), (48, 80)):
        batch = {k: torch.from_numpy(v) for k, v in make_batch(H, W, 3, cfg, seed=1, textured=True).items()}
        with pytest.raises(RuntimeError, match="must match the size"):
            O.forward(cfg, load_weights(), batch)
        with pytest.raises(EnerfError, match="divisible by 4"):
            with torch.no_grad():
                _net(cfg)(batch)


def test_source_view_counts_outside_2_to_4_are_refused_and_the_reference_never_ships_them():
    """The render kernel maps source views onto its four lane groups, so ``enerf_forward`` takes S in 2..4 (frame.hip make_plan)
    while the reference's code is generic in S (nerf.py:74-89, network.py:58-67).  Pinned here (VERDICT r05 #6b): (i) every view
    count a shipped reference config asks for — test_input_views, train_input_views, the samplers' input_views_num — is 2, 3 or 4
    (tests/golden/ref_view_counts.json, regenerated from /root/reference when it is present); (ii) S = 1 is not a frame the
    reference can render either: Agg's unbiased variance over one view is NaN (nerf.py:81, restated by the oracle); (iii) S = 1
    and S = 5 are refused loudly by the C entry, never rendered from a partial lane mapping."""
    import json
    import os
    from enerf_amd.config import EnerfConfig
    from enerf_amd.synth import make_batch
    from golden_cases import GOLDEN
    from oracle import enerf_oracle as O
    counts = json.load(open(os.path.join(GOLDEN, "ref_view_counts.json")))
    assert len(counts) >= 10 and {v for c in counts.values() for v in c} <= {2, 3, 4}
    if os.path.isdir("/root/reference/configs/enerf"):
        from oracle.scan_ref_view_counts import scan
        assert scan() == counts, "tests/golden/ref_view_counts.json is stale: python oracle/scan_ref_view_counts.py"
    cfg = EnerfConfig().with_cas(volume_planes=(8, 8), render_if=(False, True))
    with torch.no_grad():
        one = {k: torch.from_numpy(v) for k, v in make_batch(32, 64, 1, cfg, seed=1, textured=True).items()}
        assert torch.isnan(O.forward(cfg, load_weights(), one)["rgb_level1"]).all()          # the reference's own arithmetic: var of ONE view
        for S in (1, 5):
            batch = {k: torch.from_numpy(v) for k, v in make_batch(32, 64, S, cfg, seed=1, textured=True).items()}
            with pytest.raises(EnerfError, match="S in 2..4"):
                _net(cfg)(batch)


def test_empty_ray_list():
    name = "tiny_s3"
    cfg, batch, g = case_config(name), case_batch(name), load_golden(name)
    net = _net(cfg)
    out = net.render_rays(torch.zeros(1, 0, 12), level=1, batch=batch,
                          im_feat=torch.from_numpy(g["mid/feat_l2"]).reshape(1, 3, 8, 32, 64),
                          feature_volume=torch.from_numpy(g["mid/feat3d_1"]), nerf_model=net.nerf_1)
    assert out["rgb"].shape == (1, 0, 3) and out["weights"].shape == (1, 0, 2)


@pytest.mark.parametrize("force", ["v1", "v2", "pk8", "b4"])
def test_conv3d_variants_agree_with_reference(force):
    """Every conv3d code path (global-load V1 incl. the row-split small-layer form, LDS-staged V2, tap-packed, batched
    4x4x1 MFMA) on a case whose volumes are not multiples of the 8x16 LDS box (h,w = 16,24 / 8,12 / 4,6); variants are
    chosen through the explicit enerf_options_t, not the environment."""
    from enerf_amd.lib import Options
    if force == "v1":
        opt = Options(conv3d_global_only=1)
    elif force == "b4":
        opt = Options(conv3d_lds_min_voxels=1)                                          # batched-4x4 kernel on conv0 + heads
    else:                                                                               # tap-packed kernel: all layers / off
        opt = Options(conv3d_lds_min_voxels=1, conv3d_b4=1, conv3d_pk8=2 if force == "pk8" else 1)
    name = "small_s3_eval"
    cfg, batch, gold = case_config(name), case_batch(name), load_golden(name)
    net = _net(cfg)
    net.options = opt
    out = net(batch)
    for k, v in out.items():
        _close(v.numpy(), gold["out/" + k], 2e-5, k)
    # and the level-0 network (MinCostRegNet, Cin=32: two LDS channel passes)
    name = "tiny_s3"
    cfg, batch, gold = case_config(name), case_batch(name), load_golden(name)
    net = _net(cfg)
    net.options = opt
    out = net(batch)
    for k, v in out.items():
        _close(v.numpy(), gold["out/" + k], 2e-5, k)


def test_fused_build_rays_equals_separate_launch():
    """forward() folds build_rays into the render kernel's prologue (same device function): bit-identical to
    enerf_build_rays + enerf_render_rays through the render_rays surface."""
    name = "tiny_s3"
    cfg, batch, g = case_config(name), case_batch(name), load_golden(name)
    net, lib = _net(cfg), emu_lib()
    a = net(batch)
    T = lambda k: torch.from_numpy(g["mid/" + k]).contiguous()
    # level 1 of the same frame, stage by stage from the frame's own maps is covered by the stage test; here: the
    # binding rejects 8-float rays without the maps (and vice versa)
    with pytest.raises(Exception):
        lib.render_rays(torch.zeros(1, 4, 8), None, None, None, None, None, None, n_samples=2, depth_inv=False, F=11,
                        render_scale=1.0)
    r12 = lib.build_rays(batch["rays_1"], a["depth_mvs_level1"], a["std_level1"], T("nf_1"), 32, 64, False)
    assert r12.shape == (1, 32 * 64, 12)


def test_forward_generates_rays_on_device_when_absent():
    """Without ``rays_{i}`` in the batch the frame driver generates the full-image rays itself (enerf_gen_rays,
    lib/datasets/enerf_utils.py:61-71): same image as with the host-built rays."""
    name = "tiny_s3"
    cfg, batch = case_config(name), case_batch(name)
    net = _net(cfg)
    ref = net(batch)
    b2 = {k: v for k, v in batch.items() if not k.startswith("rays_")}
    out = net(b2)
    for k in ref:
        _close(out[k].numpy(), ref[k].numpy(), 1e-5, k)


def test_human_static_shapes_and_mask_dtypes():
    """network_human.py:90-107 on the device: index-list compaction, rgb scattered into zeros; static_shapes=True returns
    full-size depth/weights + the device-side count instead of the reference's data-dependent shapes."""
    name = "tiny_s4_mask"
    cfg, batch, gold = case_config(name), case_batch(name), load_golden(name)
    net = _net(cfg, human=True)
    ref = net(batch)
    m = int(batch["mask_at_box"].sum())
    assert ref["depth_level1"].shape == (1, m)
    for dt in (torch.uint8, torch.bool, torch.int64):
        b2 = dict(batch)
        b2["mask_at_box"] = batch["mask_at_box"].to(dt)
        out = net(b2)
        for k in ref:
            assert torch.equal(out[k], ref[k]), (k, dt)
    net.static_shapes = True
    out = net(batch)
    assert int(out["num_rays_level1"][0]) == m and out["depth_level1"].shape == (1, 32 * 64)
    assert torch.equal(out["depth_level1"][:, :m], ref["depth_level1"]) and torch.equal(out["rgb_level1"], ref["rgb_level1"])
    # mask with a single selected ray: the reference leaves rgb all zero (mask.sum() > 1 quirk)
    b3 = dict(batch)
    one = torch.zeros_like(batch["mask_at_box"]); one.view(-1)[777] = 1
    b3["mask_at_box"] = one
    net.static_shapes = False
    out = net(b3)
    assert out["depth_level1"].shape == (1, 1) and float(out["rgb_level1"].abs().max()) == 0.0
    with torch.no_grad():
        oref = O.forward(cfg, load_weights(), b3)
    _close(out["depth_level1"].numpy(), oref["depth_level1"].numpy(), 2e-5, "single-ray depth")
    # empty mask
    b3["mask_at_box"] = torch.zeros_like(batch["mask_at_box"])
    out = net(b3)
    assert out["depth_level1"].shape == (1, 0) and float(out["rgb_level1"].abs().max()) == 0.0


def test_mask_compact_matches_nonzero():
    lib = emu_lib()
    rng = np.random.default_rng(3)
    for n in (1, 5, 1024, 1025, 5000, 70001):
        m = torch.from_numpy((rng.uniform(size=n) < 0.37).astype(np.uint8))
        idx, cnt = lib.mask_compact(m)
        ref = torch.nonzero(m).reshape(-1).to(torch.int32)
        assert int(cnt[0]) == ref.numel() and torch.equal(idx[: ref.numel()], ref)


@pytest.mark.parametrize("smooth0_plain", [0, 1])
def test_hip_feature_net_matches_reference_feature_maps(smooth0_plain):
    """enerf_feature_net (conv2d.hip) against the reference FeatureNet's three outputs, plain and texel mode;
    with the tap-packed (8x28 tiles) and the plain (8x32) fused smooth0 kernel."""
    from enerf_amd.lib import Options
    opt = Options(featnet_smooth0_plain=smooth0_plain)
    name = "tiny_s3"
    cfg, batch, g = case_config(name), case_batch(name), load_golden(name)
    net, lib = _net(cfg), emu_lib()
    src = batch["src_inps"][0].contiguous()
    f0, f1, f2, _ = lib.feature_net(net._packed_weights("feature_net"), src, 8, options=opt)
    for a, k in ((f0, "feat_l0"), (f1, "feat_l1"), (f2, "feat_l2")):
        _close(a.permute(0, 3, 1, 2).numpy(), g["mid/" + k], 5e-6, k)
    _, _, t2, _ = lib.feature_net(net._packed_weights("feature_net"), src, 12, options=opt)
    assert torch.equal(t2[..., :8], f2)
    _close(t2[..., 8:11].permute(0, 3, 1, 2).numpy(), (src * 0.5 + 0.5).numpy(), 1e-7, "texel rgb")
    assert float(t2[..., 11].abs().max()) == 0.0
    # the three-stage form the two-stream host path uses (enerf_feature_net_stage) produces the same maps
    bufs = lib.feature_net_alloc(src, 12)
    for stage in (lib.FEAT_TRUNK, lib.FEAT_LEVEL1, lib.FEAT_LEVEL2):
        lib.feature_net_stage(net._packed_weights("feature_net"), src, bufs, stage, 12, opt)
    assert torch.equal(bufs[0], f0) and torch.equal(bufs[1], f1) and torch.equal(bufs[2], t2)
    with pytest.raises(Exception):
        lib.feature_net_stage(net._packed_weights("feature_net"), src, bufs, 7, 12)


@pytest.mark.parametrize("name", ["tiny_s3", "small_s3_eval"])
def test_torch_feature_backend_still_matches(name):
    """feature_backend='torch' (north_star's split: FeatureNet in PyTorch, adapters + HIP path after it)."""
    cfg, batch, gold = case_config(name), case_batch(name), load_golden(name)
    net = Network(cfg, lib=emu_lib(), feature_backend="torch").eval()
    net.load_state_dict(load_weights(), strict=False)
    out = net(batch)
    for k, v in out.items():
        _close(v.numpy(), gold["out/" + k], 2e-5, k)



def test_unfused_lat0_smooth0_path():
    """enerf_options_t.featnet_unfused keeps one launch per FeatureNet layer (separate conv0.0 / conv0.1, toplayer,
    lat0 / smooth0) — same features."""
    from enerf_amd.lib import Options
    name = "tiny_s3"
    cfg, batch, g = case_config(name), case_batch(name), load_golden(name)
    net, lib = _net(cfg), emu_lib()
    _, _, f2, _ = lib.feature_net(net._packed_weights("feature_net"), batch["src_inps"][0].contiguous(), 8,
                                  options=Options(featnet_unfused=1))
    _close(f2.permute(0, 3, 1, 2).numpy(), g["mid/feat_l2"], 5e-6, "feat_l2 unfused")
