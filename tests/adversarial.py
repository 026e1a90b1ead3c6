"""Adversarial-input parity cases (VERDICT r03 "next" #3): regimes a trained checkpoint produces and random-init weights
on smooth synthetic frames only reach by accident.  TEST INFRASTRUCTURE: shared by ``oracle/make_golden.py`` (which runs
the UNMODIFIED reference on these inputs, ``--case adv_*``) and by the emulator / GPU parity tests.

Each case = the seeded base batch + the seeded reference weights, then
  * ``weights``: multiplicative / sign edits of named parameters (applied to the state dict before it is loaded), and
  * ``batch``:   an edit of the batch dict (camera placement, near/far planes),
  * ``white_bkgd``: the reference's ``cfg.enerf.white_bkgd`` (utils.py:598-601).

Regimes and the reference lines they stress:
  adv_onehot  softmax_D near one-hot -> variance below the 1e-10 clamp (utils.py:658-667); the cascade's [d +- std] interval
              collapses to ~1e-5 wide, so level 1's depth planes and the per-ray sample interval are (nearly) degenerate:
              ``max(vf - vn, 1e-6)`` (utils.py:433-436) and the end-plane clamps (utils.py:122-127, 400-413)
  adv_clamp   the depth head's weights made negative on an input pushed positive (conv11's BatchNorm bias = 2): the zero padding
              then favours the first / last plane without saturating the softmax, so [d +- std] hits the ``near_far`` clamps on
              > 90 % of the pixels at both levels (utils.py:122-127, 400-413)
  adv_behind  one source camera sits INSIDE the depth range looking sideways: its pz changes sign across the volume
              (``max(pz, 1e-6)``: utils.py:80, 703) and most of its projections fall outside the image (zeros padding in the
              warp, border padding in the render gathers, utils.py:86-88, 705-711)
  adv_sigma   density head scaled up and biased: sigma up to ~1e2, ``exp(-sigma)`` reaches 0 on part of the samples and the
              transmittance products (1 - alpha + 1e-10)^k run into the denormal range (utils.py:584-589)
  adv_white   ``white_bkgd=True`` (utils.py:598-601; the softmaxed weights sum to 1, so the term is a rounding residue)
"""
from __future__ import annotations

import numpy as np

BASE = dict(H=64, W=96, S=3, planes=(16, 8), render_if=(True, True), textured=True)

ADV_CASES = {
    "adv_onehot": dict(BASE, seed=21, weights={"cost_reg_0.depth_conv.0.weight": ("scale", 400.0),
                                               "cost_reg_1.depth_conv.0.weight": ("scale", 400.0)}),
    "adv_clamp": dict(BASE, seed=22, weights={"cost_reg_0.conv11.1.bias": ("set", 2.0), "cost_reg_1.conv11.1.bias": ("set", 2.0),
                                              "cost_reg_0.depth_conv.0.weight": ("negabs", 10.0),
                                              "cost_reg_1.depth_conv.0.weight": ("negabs", 10.0)}),
    "adv_behind": dict(BASE, seed=23, batch="behind"),
    "adv_sigma": dict(BASE, seed=24, weights={"nerf_0.sigma.0.weight": ("scale", 150.0), "nerf_1.sigma.0.weight": ("scale", 400.0),
                                              "nerf_0.sigma.0.bias": ("set", 20.0), "nerf_1.sigma.0.bias": ("set", 120.0)}),
    "adv_white": dict(BASE, seed=25, white_bkgd=True),
}


def tweak_weights(sd: dict, case: str) -> dict:
    """Returns a NEW dict; values keep their container type (torch tensors or numpy arrays)."""
    out = dict(sd)
    for name, (op, k) in ADV_CASES[case].get("weights", {}).items():
        v = out[name]
        if op == "scale":
            out[name] = v * k
        elif op == "negabs":
            out[name] = -abs(v) * k
        elif op == "set":
            out[name] = v * 0 + k
        else:
            raise KeyError(op)
    return out


def _look_at(center, target):
    from enerf_amd.synth import look_at_w2c
    return look_at_w2c(np.asarray(center, np.float64), np.asarray(target, np.float64))


def tweak_batch(batch: dict, case: str) -> dict:
    """numpy batch in, numpy batch out (a copy where edited)."""
    kind = ADV_CASES[case].get("batch")
    if kind is None:
        return batch
    b = {k: v.copy() for k, v in batch.items()}
    if kind == "behind":
        # source view 2: inside the scene's depth range (z = 640, off-axis), looking across the target's frustum
        b["src_exts"][0, 2] = _look_at((12.0, -7.0, 655.0), (1000.0, 40.0, 700.0)).astype(np.float32)
    else:
        raise KeyError(kind)
    return b


def adv_config(case: str):
    from enerf_amd.config import EnerfConfig
    import dataclasses
    c = ADV_CASES[case]
    cfg = EnerfConfig().with_cas(volume_planes=c["planes"], render_if=c["render_if"])
    if c.get("white_bkgd"):
        cfg = dataclasses.replace(cfg, white_bkgd=True)
    return cfg


def adv_batch(case: str) -> dict:
    from enerf_amd.synth import make_batch
    c = ADV_CASES[case]
    return tweak_batch(make_batch(c["H"], c["W"], c["S"], adv_config(case), seed=c["seed"], textured=c["textured"]), case)


def regime_stats(cfg, sd, batch) -> dict:
    """How far into its regime a case is, measured with the oracle (torch CPU tensors in).  Used by the generator to record
    the fractions in the fixture and by the tests to assert the case still exercises what it claims."""
    import torch
    from oracle import enerf_oracle as O
    mid: dict = {}
    with torch.no_grad():
        O.forward(cfg, sd, batch, intermediates=mid)
        st = {}
        cas = cfg.cas
        for i in range(cas.num):
            p = torch.softmax(mid[f"prob_{i}"], 1)
            st[f"onehot_frac_{i}"] = float((p.max(1).values > 0.999).float().mean())
            st[f"std_clamped_frac_{i}"] = float((mid[f"std_{i}"] <= 1.0001e-5).float().mean())
            r = mid.get(f"rays12_{i}")
            if r is not None:
                st[f"ray_clamp_frac_{i}"] = float(((r[..., 8] == r[..., 10]) | (r[..., 9] == r[..., 11])).float().mean())
                out = O.render_rays(cfg, sd, r, i, batch, O.forward_feat(sd, batch["src_inps"])[f"level_{cas.render_im_feat_level[i]}"],
                                    mid[f"feat3d_{i}"], return_intermediates=True)
                xyz = out["_xyz"].reshape(1, -1, 3)
                ph = torch.cat([xyz, torch.ones_like(xyz[..., :1])], -1)
                S = batch["src_exts"].shape[1]
                pz = torch.stack([(ph @ batch["src_exts"][:, s].transpose(-1, -2))[..., 2] for s in range(S)], -1)
                st[f"render_pz_le_eps_frac_{i}"] = float((pz <= 1e-6).float().mean())
                sig = out["_raw"][..., 3]
                st[f"sigma_max_{i}"] = float(sig.max())
                st[f"alpha_one_frac_{i}"] = float((torch.exp(-sig) == 0).float().mean())
        # level-1 depth-plane clamp (utils.py:122-127): [d +- std] against level 0's near_far, at level 1's resolution
        if cas.num > 1:
            k = cas.volume_scale[1] / cas.volume_scale[0]
            d = O._resize_ac(mid["depth_0"][:, None], k, True)[:, 0]
            s = O._resize_ac(mid["std_0"][:, None], k, True)[:, 0]
            nf = O._resize_ac(mid["nf_0"], k, True)
            st["dv1_clamp_frac"] = float(((d + s > nf[:, 0]) | (d - s < nf[:, 1])).float().mean())
        # warp: share of (view, voxel) projections behind the camera / outside the source map, level 0
        P = O.proj_mats(batch, cas.im_feat_scale[0], cas.volume_scale[0])
        dv = mid["dv_0"]
        B, D, h, w = dv.shape
        ys, xs = torch.meshgrid(torch.linspace(0, h - 1, h), torch.linspace(0, w - 1, w), indexing="ij")
        g = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(h * w)], 0).repeat(1, D)
        beh, outside = [], []
        Hs, Ws = int(batch["src_inps"].shape[-2] * cas.im_feat_scale[0]), int(batch["src_inps"].shape[-1] * cas.im_feat_scale[0])
        for s_ in range(P.shape[1]):
            p = P[0, s_, :, :3] @ g + P[0, s_, :, 3:] / dv.reshape(1, -1)
            beh.append((p[2] <= 1e-6).float().mean())
            uv = p[:2] / p[2:].clamp_min(1e-6)
            outside.append(((uv[0] < 0) | (uv[0] > Ws - 1) | (uv[1] < 0) | (uv[1] > Hs - 1)).float().mean())
        st["warp_pz_le_eps_frac_per_view"] = [float(x) for x in beh]
        st["warp_outside_frac_per_view"] = [float(x) for x in outside]
    return st
