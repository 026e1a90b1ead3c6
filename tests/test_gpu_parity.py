"""Parity of the HIP path on a real MI355X (through the C ABI) against reference goldens and the oracle.

Tolerances: the reference's own fp32 result moves by 5.8e-5 max-abs in rgb between thread counts
(SURVEY.md §8c); north_star asks for <= 1e-3 PSNR / depth deviation.  We require max|err|/max|ref| <= 1e-4
on every output and PSNR(ours, reference) >= 70 dB.
"""
import numpy as np
import pytest
import torch

from enerf_amd.config import EnerfConfig
from enerf_amd.synth import make_batch
from oracle import enerf_oracle as O
from golden_cases import CASES, case_batch, case_config, check_sparse_golden, load_golden, load_weights

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU (run with -m gpu on the MI355X box)")]
REL_TOL = 1e-4


def _dev():
    return torch.device("cuda:0")


def _net(cfg, human=False, feature_backend="hip"):
    from enerf_amd.network import Network, NetworkHuman
    net = (NetworkHuman if human else Network)(cfg, feature_backend=feature_backend)   # enerf_amd/libenerf_hip.so
    net.load_state_dict(load_weights(), strict=False)
    return net.to(_dev()).eval()


def _to(batch):
    return {k: v.to(_dev()) for k, v in batch.items()}


def _rel(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-12)


def test_product_library_is_loaded_not_a_fallback():
    from enerf_amd.lib import LIB_PATH, get_lib
    assert get_lib().path == LIB_PATH
    maps = open("/proc/self/maps").read()
    assert "libenerf_hip.so" in maps


@pytest.mark.parametrize("backend", ["hip", "torch"])
@pytest.mark.parametrize("name", list(CASES))
def test_goldens(name, backend):
    cfg, gold = case_config(name), load_golden(name)
    out = _net(cfg, CASES[name]["human"], backend)(_to(case_batch(name)))
    torch.cuda.synchronize()
    assert sorted(out) == sorted(k[4:] for k in gold if k.startswith("out/"))
    for k, v in out.items():
        assert v.shape == gold["out/" + k].shape, k
        assert _rel(v.cpu().numpy(), gold["out/" + k]) < REL_TOL, (k, _rel(v.cpu().numpy(), gold["out/" + k]))


def test_stages_vs_reference_intermediates():
    from enerf_amd.lib import get_lib
    name = "tiny_s3"
    cfg, batch, g = case_config(name), _to(case_batch(name)), load_golden(name)
    lib, cas, dev = get_lib(), cfg.cas, _dev()
    T = lambda k: torch.from_numpy(g["mid/" + k]).contiguous().to(dev)
    net = _net(cfg)
    prev = None
    for i in range(2):
        P = lib.get_proj_mats(batch["src_ixts"], batch["src_exts"], batch["tar_ixt"], batch["tar_ext"],
                              cas.im_feat_scale[i], cas.volume_scale[i])
        assert _rel(P.cpu(), g[f"mid/proj_{i}"]) < 1e-6
        D = cas.volume_planes[i]
        h, w = g[f"mid/dv_{i}"].shape[-2:]
        dv, nf = lib.get_depth_values(batch["near_far"], prev, 1, D, h, w, cas.depth_inv[i])
        assert _rel(dv.cpu(), g[f"mid/dv_{i}"]) < 1e-6 and _rel(nf.cpu(), g[f"mid/nf_{i}"]) < 1e-6
        f = T(f"feat_l{i}")
        S, C, Hs, Ws = f.shape
        fcl = lib.channels_last(f.reshape(S, C, Hs * Ws), S, C, Hs * Ws).view(1, S, Hs, Ws, C)
        vol = lib.build_feature_volume(fcl, T(f"proj_{i}"), T(f"dv_{i}"), C)
        assert _rel(vol.permute(0, 4, 1, 2, 3).cpu(), g[f"mid/vol_{i}"]) < 2e-5
        vin = lib.channels_last(T(f"vol_{i}").reshape(1, C, -1), 1, C, D * h * w).view(1, D, h, w, C)
        m = getattr(net, f"cost_reg_{i}")
        feat, prob = lib.cost_reg(net._packed_weights(f"cost_reg_{i}"), m.in_channels, m.full, vin)
        assert _rel(feat.permute(0, 4, 1, 2, 3).cpu(), g[f"mid/feat3d_{i}"]) < 2e-5
        assert _rel(prob.cpu(), g[f"mid/prob_{i}"]) < 2e-5
        d, s = lib.depth_regression(T(f"prob_{i}"), T(f"dv_{i}"), cas.depth_inv[i])
        assert _rel(d.cpu(), g[f"mid/depth_{i}"]) < 1e-5 and _rel(s.cpu(), g[f"mid/std_{i}"]) < 1e-4
        Hr, Wr = int(32 * cas.render_scale[i]), int(64 * cas.render_scale[i])
        r = lib.build_rays(batch[f"rays_{i}"], T(f"depth_{i}"), T(f"std_{i}"), T(f"nf_{i}"), Hr, Wr, cas.depth_inv[i])
        assert _rel(r.cpu(), g[f"mid/rays12_{i}"]) < 1e-6
        out = net.render_rays(T(f"rays12_{i}"), level=i, batch=batch,
                              im_feat=T(f"feat_l{cas.render_im_feat_level[i]}")[None],
                              feature_volume=T(f"feat3d_{i}"), nerf_model=getattr(net, f"nerf_{i}"))
        assert _rel(out["rgb"].cpu(), g[f"out/rgb_level{i}"]) < 2e-5
        assert _rel(out["depth"].cpu(), g[f"out/depth_level{i}"]) < 2e-5
        assert _rel(out["weights"].cpu(), g[f"out/weights_level{i}"]) < 2e-5
        prev = (T(f"depth_{i}"), T(f"std_{i}"), T(f"nf_{i}"))


def test_hip_feature_net_vs_reference_maps():
    from enerf_amd.lib import get_lib
    name = "tiny_s3"
    cfg, batch, g = case_config(name), _to(case_batch(name)), load_golden(name)
    net, lib = _net(cfg), get_lib()
    f0, f1, f2, _ = lib.feature_net(net._packed_weights("feature_net"), batch["src_inps"][0].contiguous(), 8)
    for a, k in ((f0, "feat_l0"), (f1, "feat_l1"), (f2, "feat_l2")):
        assert _rel(a.permute(0, 3, 1, 2).cpu(), g["mid/" + k]) < 1e-5, k


@pytest.mark.parametrize("S,B,hw", [(3, 1, (128, 160)), (4, 2, (64, 96)), (2, 1, (96, 128))])
def test_against_oracle_medium(S, B, hw):
    """Seeded medium-size frames, both levels rendered, ragged ray lists; oracle on CPU in seconds."""
    cfg = EnerfConfig().with_cas(volume_planes=(16, 8))
    b = make_batch(hw[0], hw[1], S, cfg, seed=100 + S, B=B, textured=True)
    keep = np.random.default_rng(S).permutation(hw[0] * hw[1])[: hw[0] * hw[1] - 37]
    b["rays_1"] = np.ascontiguousarray(b["rays_1"][:, keep])
    batch = {k: torch.from_numpy(v) for k, v in b.items()}
    out = _net(cfg)(_to(batch))
    with torch.no_grad():
        ref = O.forward(cfg, load_weights(), batch)
    for k in ref:
        assert _rel(out[k].cpu(), ref[k]) < REL_TOL, (k, _rel(out[k].cpu(), ref[k]))
    assert O.psnr(out["rgb_level1"].cpu(), ref["rgb_level1"]) > 70.0


def test_full_size_dtu_eval_vs_oracle_and_properties():
    """BASELINE config 2 (512x640, 3 views, planes 48,8, render_if False,True)."""
    cfg = EnerfConfig.dtu_eval()
    b = make_batch(512, 640, 3, cfg, seed=0, textured=True)
    batch = {k: torch.from_numpy(v) for k, v in b.items()}
    net = _net(cfg)
    out = net(_to(batch))
    out2 = net(_to(batch))
    torch.cuda.synchronize()
    for k in out:                                      # run-to-run determinism
        assert torch.equal(out[k], out2[k]), k
    rgb, w, d = out["rgb_level1"].cpu(), out["weights_level1"].cpu(), out["depth_level1"].cpu()
    assert torch.isfinite(rgb).all() and torch.isfinite(d).all()
    assert torch.allclose(w.sum(-1), torch.ones_like(w.sum(-1)), atol=1e-5)          # softmaxed weights
    src = torch.from_numpy(b["src_inps"]) * 0.5 + 0.5                                # rgb is a convex blend x alpha
    assert rgb.min() >= -1e-5 and rgb.max() <= float(src.max()) + 1e-5
    assert d.min() >= 425.0 - 1e-2 and d.max() <= 905.0 + 1e-2                       # inside [near, far]
    # pinned to the REFERENCE itself at this size: sparse digest of the unmodified reference's outputs (every 97th ray + norms)
    check_sparse_golden("dtu_full", out, REL_TOL)
    with torch.no_grad():
        ref = O.forward(cfg, load_weights(), batch)
    for k in ref:
        assert _rel(out[k].cpu(), ref[k]) < REL_TOL, (k, _rel(out[k].cpu(), ref[k]))
    psnr = O.psnr(rgb, ref["rgb_level1"])
    assert psnr > 70.0, psnr
    # north_star tolerance: PSNR against a common pseudo ground truth moves by < 1e-3 dB
    gt = torch.clamp(ref["rgb_level1"] + 0.05 * torch.randn_like(ref["rgb_level1"]), 0, 1)
    assert abs(O.psnr(rgb, gt) - O.psnr(ref["rgb_level1"], gt)) < 1e-3
    # every optional kernel choice (enerf_options_t) at the full size too: the throughput set the frame pipeline uses, and
    # the plain fallbacks — same oracle, same tolerance
    from enerf_amd.lib import Options, throughput_options
    variants = {"throughput": throughput_options(), "global_only": Options(conv3d_global_only=1),
                "pk8_off": Options(conv3d_pk8=1), "unfused_fpn": Options(featnet_unfused=1),
                "plain_smooth0": Options(featnet_smooth0_plain=1),
                "t2_round2": Options(conv3d_t2_variant=1), "t2_all": Options(conv3d_t2_variant=2),
                "small_rt2ct2": Options(conv3d_small_variant=2), "small_ct4": Options(conv3d_small_variant=3),
                "small_split3": Options(conv3d_small_variant=1), "b4_round2": Options(conv3d_b4=3), "separate_depth_prep": Options(fuse_depth_prep=1),
                "side_gate": Options(side_gate=2),
                # the opt-in bf16 variants of the render MLP (two / three bf16 pieces per fp32 operand on the bf16 matrix cores)
                "render_bf16x3": Options(render_precision=2), "render_bf16x6": Options(render_precision=3)}
    for name, opt in variants.items():
        o = net._forward(_to(batch), opt)
        for k in ref:
            assert _rel(o[k].cpu(), ref[k]) < REL_TOL, (name, k, _rel(o[k].cpu(), ref[k]))
    # the side lane only changes WHEN the FeatureNet's top-down half runs, not what is computed: bit-identical frames
    o = net._forward(_to(batch), Options(single_stream=1))
    torch.cuda.synchronize()
    for k in out:
        assert torch.equal(o[k], out[k]), k


def test_whole_frame_hip_graph_replay_matches_eager():
    """enerf_amd.graph.GraphedFrame: one captured forward replayed on new inputs == eager forward, bit for bit."""
    from enerf_amd.graph import GraphedFrame
    cfg = EnerfConfig.dtu_eval()
    net = _net(cfg)
    batches = [_to({k: torch.from_numpy(v) for k, v in make_batch(256, 320, 3, cfg, seed=s, textured=True).items()})
               for s in range(3)]
    eager = [{k: v.clone() for k, v in net(b).items()} for b in batches]
    frame = GraphedFrame(net, batches[0])
    for b, e in zip(batches, eager):
        out = frame(b)
        torch.cuda.synchronize()
        for k in e:
            assert torch.equal(out[k], e[k]), k


def test_frame_pipeline_two_streams_matches_sequential():
    """enerf_amd.pipeline.FramePipeline: frames in flight on two HIP streams == the same frames one after the other."""
    from enerf_amd.pipeline import FramePipeline
    cfg = EnerfConfig.dtu_eval()
    net = _net(cfg)
    batches = [_to({k: torch.from_numpy(v) for k, v in make_batch(256, 320, 3, cfg, seed=s, textured=True).items()})
               for s in range(5)]
    ref = [{k: v.clone() for k, v in net(b).items()} for b in batches]
    torch.cuda.synchronize()
    for depth in (2, 4):
        pipe = FramePipeline(net, depth=depth)
        outs = [pipe.submit(b) for b in batches + batches]
        pipe.join()
        torch.cuda.synchronize()
        for (o, ev), r in zip(outs, ref + ref):
            assert ev.query()
            for k in r:
                assert torch.equal(o[k], r[k]), k
    # throughput tuning swaps kernel variants (summation order changes): same frames within the parity tolerance
    pipe = FramePipeline(net, depth=4, throughput_tuning=True)
    outs = [pipe.submit(b) for b in batches]
    pipe.close()
    torch.cuda.synchronize()
    for (o, _), r in zip(outs, ref):
        for k in r:
            assert _rel(o[k].cpu(), r[k].cpu()) < 2e-5, (k, _rel(o[k].cpu(), r[k].cpu()))
    assert net.options is None                       # the pipeline's choices never leak into the network


def _full_size_check(cfg, b, human, keys_psnr, golden):
    """HIP path on one full-size frame vs (i) the sparse digest of the unmodified REFERENCE's outputs at this size
    (tests/golden/<golden>.npz, oracle/make_golden.py::FULL_CASES) and (ii) the oracle on every element: every output within
    REL_TOL, PSNR(ours, oracle) > 70 dB, and the north_star bound (PSNR against a common pseudo ground truth moves by
    < 1e-3 dB)."""
    batch = {k: torch.from_numpy(v) for k, v in b.items()}
    net = _net(cfg, human)
    out = net(_to(batch))
    out2 = net(_to(batch))
    torch.cuda.synchronize()
    for k in out:                                      # run-to-run determinism
        assert torch.equal(out[k], out2[k]), k
    check_sparse_golden(golden, out, REL_TOL)
    with torch.no_grad():
        ref = O.forward(cfg, load_weights(), batch)
    assert sorted(out) == sorted(ref)
    for k in ref:
        assert out[k].shape == ref[k].shape, k
        assert _rel(out[k].cpu(), ref[k]) < REL_TOL, (k, _rel(out[k].cpu(), ref[k]))
    g = torch.Generator().manual_seed(0)
    for k in keys_psnr:
        rgb, r = out[k].cpu(), ref[k]
        assert O.psnr(rgb, r) > 70.0, (k, O.psnr(rgb, r))
        gt = torch.clamp(r + 0.05 * torch.randn(r.shape, generator=g), 0, 1)
        assert abs(O.psnr(rgb, gt) - O.psnr(r, gt)) < 1e-3, k
    return out, ref


def test_full_size_dtu_both_levels_vs_oracle():
    """BASELINE configs[0] (configs/enerf/dtu/scan114.yaml's shape; the configuration BASELINE.md's CPU figure is quoted on):
    512x640, 3 views, planes 48,8, render_if True,True — k_render_rays<9,3,...> (level 0: C = 32, 8 samples per ray) on
    20,480 rays and <3,3,...> on 327,680, pinned to the unmodified reference's digest (dtu_full_tt) and to the oracle."""
    cfg = EnerfConfig().with_cas(volume_planes=(48, 8), render_if=(True, True))
    out, ref = _full_size_check(cfg, make_batch(512, 640, 3, cfg, seed=0, textured=True), False, ("rgb_level0", "rgb_level1"),
                                "dtu_full_tt")
    assert out["rgb_level0"].shape == (1, 128 * 160, 3) and out["rgb_level1"].shape == (1, 512 * 640, 3)


def test_full_size_lego_800x800_4views_both_levels_vs_oracle():
    """BASELINE config 3 at its real shape (configs/enerf/nerf/lego.yaml:4-8, lib/datasets/nerf/enerf.py:46-49,92):
    H=W=800, S=4, planes 64,8, render_if True,True — lego pinhole intrinsics, near_far [2.5, 5.5]; exercises
    k_render_rays<9,4,2> (level 0: C=32, 8 samples) and <3,4,*> (level 1) on 40,000 + 640,000 rays."""
    from enerf_amd.synth import make_lego_batch
    cfg = EnerfConfig()
    out, ref = _full_size_check(cfg, make_lego_batch(800, 800, 4, cfg, seed=5), False, ("rgb_level0", "rgb_level1"), "lego_full")
    assert out["rgb_level1"].shape == (1, 640000, 3) and out["rgb_level0"].shape == (1, 40000, 3)
    d = out["depth_level1"].cpu()
    assert d.min() >= 2.5 - 1e-4 and d.max() <= 5.5 + 1e-4


def test_full_size_llff_640x960_vs_oracle():
    """The reference's fourth eval config on this path (configs/enerf/llff_eval.yaml:9-16,26-28: planes 32,8, render_if False,True,
    input_h_w 640 x 960; not a BASELINE config) at its real shape: a 2:3 aspect, level-0 volume 32 x 80 x 120, 614,400 rays, pinned
    to the unmodified reference's digest (llff_full) and to the oracle (VERDICT r05 #6a)."""
    cfg = EnerfConfig().with_cas(volume_planes=(32, 8), render_if=(False, True))
    out, ref = _full_size_check(cfg, make_batch(640, 960, 3, cfg, seed=7, textured=True), False, ("rgb_level1",), "llff_full")
    assert out["rgb_level1"].shape == (1, 640 * 960, 3) and out["depth_mvs_level1"].shape == (1, 320, 480)


def test_full_size_zju_1024_4views_masked_vs_oracle():
    """BASELINE config 4 at its real shape (configs/enerf/zjumocap_eval.yaml:14,20,39 with input_ratio 1.0 and 4 input
    views; lib/networks/enerf/network_human.py:90-107): 1024x1024, S=4, planes 32,8, render_if False,True, rays compacted
    by mask_at_box on the device and rgb scattered back into zeros."""
    from enerf_amd.synth import make_zju_batch
    cfg = EnerfConfig().with_cas(volume_planes=(32, 8), render_if=(False, True))
    b = make_zju_batch(1024, 1024, 4, cfg, seed=6)
    out, ref = _full_size_check(cfg, b, True, ("rgb_level1",), "zju_full")
    m = torch.from_numpy(b["mask_at_box"]).bool().reshape(-1)
    assert out["rgb_level1"].shape == (1, 1024 * 1024, 3)
    assert float(out["rgb_level1"][0].cpu()[~m].abs().max()) == 0.0              # outside the box: exact zeros
    assert out["depth_level1"].shape == (1, int(m.sum()))                          # compacted, like the reference


def test_human_path_has_no_implicit_host_sync_and_runs_under_graph_and_pipeline():
    """network_human.py:90-107 on the device (index-list compaction + scatter): under
    ``torch.cuda.set_sync_debug_mode("error")`` any implicit synchronisation (``.item()``, boolean-mask indexing,
    pageable copies) raises.  static_shapes=True must be clean at the tiny golden size and at 1024x1024; the default
    (reference shapes) reads the count back through a pinned non-blocking copy + an explicit event wait on three tiny
    kernels, which the debug mode accepts as well.  The static mode is then captured in a HIP graph and pipelined."""
    from enerf_amd.graph import GraphedFrame
    from enerf_amd.pipeline import FramePipeline
    from enerf_amd.synth import make_zju_batch
    name = "tiny_s4_mask"
    cfg, gold = case_config(name), load_golden(name)
    net = _net(cfg, human=True)
    batch = _to(case_batch(name))
    ref = {k: v.clone() for k, v in net(batch).items()}
    cfg2 = EnerfConfig().with_cas(volume_planes=(32, 8), render_if=(False, True))
    big = _to({k: torch.from_numpy(v) for k, v in make_zju_batch(1024, 1024, 4, cfg2, seed=6).items()})   # uploads sync
    net2 = _net(cfg2, human=True)
    net2.static_shapes = True
    net2.prepare()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = net(batch)                                   # reference shapes: explicit event wait only
        net.static_shapes = True
        outs = net(batch)
        obig = net2(big)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    m = int(batch["mask_at_box"].sum())
    for k in ref:
        assert torch.equal(out[k], ref[k]), k
    assert int(outs["num_rays_level1"][0]) == m and torch.equal(outs["depth_level1"][:, :m], ref["depth_level1"])
    assert torch.equal(outs["rgb_level1"], ref["rgb_level1"])
    assert _rel(ref["rgb_level1"].cpu(), gold["out/rgb_level1"]) < REL_TOL
    assert int(obig["num_rays_level1"][0]) == int(big["mask_at_box"].sum()) and bool(torch.isfinite(obig["rgb_level1"]).all())
    # graph capture + frames in flight (impossible with the host-side boolean-mask indexing of round 1)
    frame = GraphedFrame(net, batch)
    g = frame(batch)
    torch.cuda.synchronize()
    assert torch.equal(g["rgb_level1"], ref["rgb_level1"])
    pipe = FramePipeline(net, depth=3)
    res = [pipe.submit(batch)[0] for _ in range(6)]
    pipe.close()
    torch.cuda.synchronize()
    for r in res:
        assert torch.equal(r["rgb_level1"], ref["rgb_level1"])
