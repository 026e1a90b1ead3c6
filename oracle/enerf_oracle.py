"""CPU oracle for the ENeRF rendering hot path  —  TEST INFRASTRUCTURE ONLY.

A functional, cfg-global-free restatement (torch fp32 on CPU) of one ``Network.forward(batch)`` of
zju3dv/ENeRF.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module, and only as the checker; the product path (``enerf_amd``) never does.

PARITY PIN: the reference ships no tests or golden vectors for this path, so the pin
is the reference itself: ``oracle/make_golden.py`` imports the unmodified reference modules in the
build container, runs them on seeded inputs and commits the tensors under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors stage by stage (they
agree bit-for-bit on the same torch build because the same torch primitives are applied in the same
order).  The third-party arithmetic at the boundary is torch's (``F.grid_sample``,
``F.interpolate``, ``conv3d``, ``softmax``, ``softplus``, ``var``, ``cumprod``, ``inverse``); their
tap/weight semantics are restated explicitly in ``enerf_amd/csrc/common.h`` and checked against torch.

Every function cites the reference lines it follows (paths relative to /root/reference).
All weights come from a reference ``state_dict`` (names: SURVEY.md §8b).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


# --------------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------------
_BN_TRAIN = False      # set by train_step(): BatchNorm with batch statistics (the reference network in .train() mode)


def _bn(sd, p, x):
    """BatchNorm{2,3}d (torch default eps=1e-5): running statistics in eval mode, batch statistics inside train_step()."""
    if _BN_TRAIN:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], training=True, eps=BN_EPS)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], training=False, eps=BN_EPS)


def _resize_ac(x, scale, recompute=None):
    """Bilinear resize, align_corners=True; out = floor(in*scale) (utils.py:115-117,394-396,611)."""
    kw = {} if recompute is None else {"recompute_scale_factor": recompute}
    return F.interpolate(x, None, scale_factor=scale, mode="bilinear", align_corners=True, **kw)


# --------------------------------------------------------------------------------------------------
# S0  FeatureNet  (feature_net.py:4-36; ConvBnReLU utils.py:10-20)   — stays PyTorch in the product
# --------------------------------------------------------------------------------------------------
def feature_net(sd, x, prefix="feature_net"):
    def cbr(p, x, stride, pad):
        return F.relu(_bn(sd, p + ".bn", F.conv2d(x, sd[p + ".conv.weight"], None, stride, pad)))

    def conv(p, x, pad=0):
        return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], 1, pad)

    def up_add(a, b):                                                         # feature_net.py:24-25
        return F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=True) + b

    p = prefix
    c0 = cbr(p + ".conv0.1", cbr(p + ".conv0.0", x, 1, 1), 1, 1)
    c1 = cbr(p + ".conv1.1", cbr(p + ".conv1.0", c0, 2, 2), 1, 1)
    c2 = cbr(p + ".conv2.1", cbr(p + ".conv2.0", c1, 2, 2), 1, 1)
    f2 = conv(p + ".toplayer", c2)
    f1 = up_add(f2, conv(p + ".lat1", c1))
    f0 = up_add(f1, conv(p + ".lat0", c0))
    f1 = conv(p + ".smooth1", f1, 1)
    f0 = conv(p + ".smooth0", f0, 1)
    return f2, f1, f0


def forward_feat(sd, src_inps):
    """network.py:58-67."""
    B, S, C, H, W = src_inps.shape
    f2, f1, f0 = feature_net(sd, src_inps.reshape(B * S, C, H, W))
    return {"level_2": f0.reshape(B, S, -1, H, W),
            "level_1": f1.reshape(B, S, -1, H // 2, W // 2),
            "level_0": f2.reshape(B, S, -1, H // 4, W // 4)}


# --------------------------------------------------------------------------------------------------
# S1  depth hypotheses  (utils.py:98-151)
# --------------------------------------------------------------------------------------------------
def depth_values(cfg, batch, level, D, depth, std, near_far):
    cas = cfg.cas
    nf = batch["near_far"]
    B = nf.shape[0]
    H, W = batch["src_inps"].shape[-2:]
    h, w = int(H * cas.volume_scale[level]), int(W * cas.volume_scale[level])
    t = torch.linspace(0.0, 1.0, steps=D, dtype=torch.float32)
    if depth is None:
        tt = t.view(1, -1).repeat(B, 1)
        if cas.depth_inv[level]:                                              # utils.py:104-107
            dv = 1.0 / (1.0 / nf[:, :1] + tt * (1.0 / nf[:, 1:] - 1.0 / nf[:, :1]))
        else:                                                                 # utils.py:109-110
            dv = nf[:, :1] + (nf[:, 1:] - nf[:, :1]) * tt
        dv = dv.view(B, D, 1, 1).repeat(1, 1, h, w)
    else:
        k = cas.volume_scale[level] / cas.volume_scale[level - 1]
        if k != 1.0:                                                          # utils.py:113-117
            depth = _resize_ac(depth[:, None], k, True)[:, 0]
            std = _resize_ac(std[:, None], k, True)[:, 0]
            near_far = _resize_ac(near_far, k, True)
        if not cas.depth_inv[level - 1]:
            raise NotImplementedError("reference traps here (utils.py:130)")
        lo = torch.minimum(depth + std, near_far[:, 0])                       # utils.py:123-125
        hi = torch.maximum(depth - std, near_far[:, 1])                       # utils.py:126-127
        nn_, ff_ = 1.0 / lo, 1.0 / hi                                         # utils.py:128
        tt = t.view(1, D, 1, 1)
        if cas.depth_inv[level]:                                              # utils.py:137-141
            dv = 1.0 / (1.0 / nn_[:, None] + tt * (1.0 / ff_[:, None] - 1.0 / nn_[:, None]))
        else:                                                                 # utils.py:143
            dv = nn_[:, None] + tt * (ff_[:, None] - nn_[:, None])
    out_nf = dv[:, [0, -1]].detach()                                          # utils.py:148
    if cas.depth_inv[level]:
        out_nf = 1.0 / torch.clamp_min(out_nf, 1e-6)                          # utils.py:149-150
    return dv.contiguous(), out_nf


# --------------------------------------------------------------------------------------------------
# S2  projection matrices  (utils.py:35-55)
# --------------------------------------------------------------------------------------------------
def proj_mats(batch, src_scale, tar_scale):
    B, S = batch["src_inps"].shape[:2]
    Ks = batch["src_ixts"].clone()
    Ks[:, :, :2] *= src_scale
    src = Ks @ batch["src_exts"][:, :, :3]
    Kt = batch["tar_ixt"].clone()
    Kt[:, :2] *= tar_scale
    tar = Kt @ batch["tar_ext"][:, :3]
    last = torch.zeros(B, 1, 4)
    last[:, :, 3] = 1
    tar_inv = torch.inverse(torch.cat([tar, last], 1))
    return src.view(B, S, 3, 4) @ tar_inv.view(B, 1, 4, 4)


# --------------------------------------------------------------------------------------------------
# S3/S4  homography warp + variance  (utils.py:57-95, 322-349)
# --------------------------------------------------------------------------------------------------
def homo_warp(src_feat, proj, dv):
    B, D, h, w = dv.shape
    C, Hs, Ws = src_feat.shape[1:]
    R, T = proj[:, :, :3], proj[:, :, 3:]
    ys, xs = torch.meshgrid(torch.linspace(0, h - 1, h), torch.linspace(0, w - 1, w), indexing="ij")
    g = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(h * w)], 0)[None].expand(B, -1, -1)
    g = g.repeat(1, 1, D)                                                     # utils.py:70
    p = R @ g + T / dv.reshape(B, 1, D * h * w)                               # utils.py:72
    uv = p[:, :2] / torch.clamp_min(p[:, 2:], 1e-6)                           # utils.py:80
    gx = uv[:, 0] / ((Ws - 1) / 2) - 1                                        # utils.py:82
    gy = uv[:, 1] / ((Hs - 1) / 2) - 1                                        # utils.py:83
    grid = torch.stack([gx, gy], -1).view(B, D, h * w, 2)
    out = F.grid_sample(src_feat, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.view(B, C, D, h, w)


def feature_volume(cfg, feats_level, batch, D, depth, std, near_far, level):
    cas = cfg.cas
    B, S = feats_level.shape[:2]
    dv, nf = depth_values(cfg, batch, level, D, depth, std, near_far)
    P = proj_mats(batch, cas.im_feat_scale[level], cas.volume_scale[level])   # utils.py:326
    s1, s2 = 0, 0
    for s in range(S):                                                        # utils.py:331-338
        wv = homo_warp(feats_level[:, s], P[:, s], dv)
        s1 = s1 + wv
        s2 = s2 + wv ** 2
    var = s2.div_(S).sub_(s1.div_(S).pow_(2))                                 # utils.py:345
    return var, dv, nf


# --------------------------------------------------------------------------------------------------
# S5  3D cost regularisation  (cost_reg_net.py:4-86; ConvBnReLU3D utils.py:22-33)
# --------------------------------------------------------------------------------------------------
def cost_reg(sd, prefix, x):
    """MinCostRegNet (no conv5-7) or CostRegNet, decided by the keys present in the state dict."""
    def cbr(n, x, stride=1):
        p = f"{prefix}.{n}"
        return F.relu(_bn(sd, p + ".bn", F.conv3d(x, sd[p + ".conv.weight"], None, stride, 1)))

    def up(n, x):
        p = f"{prefix}.{n}"
        y = F.conv_transpose3d(x, sd[p + ".0.weight"], None, stride=2, padding=1, output_padding=1)
        return _bn(sd, p + ".1", y)

    c0 = cbr("conv0", x)
    c2 = cbr("conv2", cbr("conv1", c0, 2))
    c4 = cbr("conv4", cbr("conv3", c2, 2))
    y = c4
    if f"{prefix}.conv5.conv.weight" in sd:                                    # CostRegNet :38-40
        y = cbr("conv6", cbr("conv5", c4, 2))
        y = c4 + up("conv7", y)
    y = c2 + up("conv9", y)
    y = c0 + up("conv11", y)
    feat = F.conv3d(y, sd[f"{prefix}.feat_conv.0.weight"], None, 1, 1)
    prob = F.conv3d(y, sd[f"{prefix}.depth_conv.0.weight"], None, 1, 1).squeeze(1)
    return feat, prob


# --------------------------------------------------------------------------------------------------
# S6  depth regression  (utils.py:658-667)
# --------------------------------------------------------------------------------------------------
def depth_regression(cfg, prob, dv, level):
    p = F.softmax(prob, 1)
    v = 1.0 / torch.clamp_min(dv, 1e-6) if cfg.cas.depth_inv[level] else dv
    mu = torch.sum(p * v, 1)
    var = (p * (v - mu.unsqueeze(1)) ** 2).sum(1)
    return mu, torch.clamp_min(var, 1e-10).sqrt()


# --------------------------------------------------------------------------------------------------
# S7  per-ray bounds  (utils.py:390-420)
# --------------------------------------------------------------------------------------------------
def build_rays(cfg, depth, std, batch, near_far, level):
    cas = cfg.cas
    k = cas.render_scale[level] / cas.volume_scale[level]
    if k != 1.0:
        depth = _resize_ac(depth[:, None], k)[:, 0]
        std = _resize_ac(std[:, None], k)[:, 0]
        near_far = _resize_ac(near_far, k)
    if cas.depth_inv[level]:                                                  # utils.py:402-407
        rn = torch.minimum(depth + std, near_far[:, 0])
        rf = torch.maximum(depth - std, near_far[:, 1])
    else:                                                                     # utils.py:409-413
        rn = torch.maximum(depth - std, near_far[:, 0])
        rf = torch.minimum(depth + std, near_far[:, 1])
    rays = batch[f"rays_{level}"]
    uv = rays[:, :, 6:].long()
    B = rays.shape[0]
    pick = lambda m: torch.stack([m[i][uv[i][:, 1], uv[i][:, 0]] for i in range(B)])
    return torch.cat([rays, pick(rn)[..., None], pick(rf)[..., None],
                      pick(near_far[:, 0])[..., None], pick(near_far[:, 1])[..., None]], -1)


# --------------------------------------------------------------------------------------------------
# S8  sample placement  (utils.py:422-441)
# --------------------------------------------------------------------------------------------------
def sample_along_depth(cfg, rays, n_samples, level):
    o, d, uv = rays[..., :3], rays[..., 3:6], rays[..., 6:8]
    rn, rf, vn, vf = rays[..., 8:9], rays[..., 9:10], rays[..., 10:11], rays[..., 11:12]
    if n_samples == 1:
        z = rn + (rf - rn) * 0.5
    else:
        z = rn + (rf - rn) * torch.linspace(0.0, 1.0, n_samples)[None, None]
    if cfg.cas.depth_inv[level]:
        xyz = o[..., None, :] + d[..., None, :] * (1 / torch.clamp_min(z[..., None], 1e-6))
        dn = (vn - z) / torch.clamp_min(vn - vf, 1e-6)
    else:
        xyz = o[..., None, :] + d[..., None, :] * z[..., None]
        dn = (z - vn) / torch.clamp_min(vf - vn, 1e-6)
    uvd = torch.cat([uv[..., None, :].repeat(1, 1, n_samples, 1), dn[..., None]], -1)
    return xyz, uvd, z


def unpreprocess(src_inps, render_scale):
    """utils.py:605-612."""
    img = src_inps * 0.5 + 0.5
    B, S, C, H, W = img.shape
    img = _resize_ac(img.reshape(B * S, C, H, W), render_scale, True)
    return img.reshape(B, S, C, int(H * render_scale), int(W * render_scale))


# --------------------------------------------------------------------------------------------------
# S9/S10  feature fetches  (utils.py:456-458, 689-722)
# --------------------------------------------------------------------------------------------------
def vox_feat(ndc, vol):
    g = ndc[:, None, None] * 2.0 - 1.0
    return F.grid_sample(vol, g, align_corners=True)[:, :, 0, 0].permute(0, 2, 1)


def img_feat(cfg, xyz, img_feat_rgb, batch, level):
    B, S, C, H, W = img_feat_rgb.shape
    n_rays, n_samp = xyz.shape[1:3]
    p = xyz.reshape(B, n_rays * n_samp, 3)
    p = torch.cat([p, torch.ones_like(p[..., :1])], -1)
    rs = cfg.cas.render_scale[level]
    per_view = []
    for s in range(S):
        cam = (p @ batch["src_exts"][:, s].transpose(-1, -2))[..., :3]
        K = batch["src_ixts"][:, s].clone()
        K[:, :2] *= rs
        pix = cam @ K.transpose(-1, -2)
        g = pix[..., :2] / torch.clamp_min(pix[..., 2:], 1e-6)
        g = torch.stack([g[..., 0] / (W - 1), g[..., 1] / (H - 1)], -1) * 2.0 - 1.0
        f = F.grid_sample(img_feat_rgb[:, s], g[:, None], align_corners=True, mode="bilinear",
                          padding_mode="border").permute(0, 2, 3, 1)[:, 0]
        ct = batch["tar_ext"].inverse()[:, :3, 3]
        cs = batch["src_exts"][:, s].inverse()[:, :3, 3]
        dt = p[..., :3] - ct[:, None]
        ds = p[..., :3] - cs[:, None]
        dt = dt / (torch.norm(dt, dim=-1, keepdim=True) + 1e-6)
        ds = ds / (torch.norm(ds, dim=-1, keepdim=True) + 1e-6)
        df = dt - ds
        dirc = df / torch.clamp(torch.norm(df, dim=-1, keepdim=True), min=1e-6)
        dot = torch.sum(dt * ds, -1, keepdim=True)
        per_view.append(torch.cat([f, dirc, dot], -1))
    return torch.stack(per_view, -2)


# --------------------------------------------------------------------------------------------------
# S11  MLP  (nerf.py:29-43, 74-89)
# --------------------------------------------------------------------------------------------------
def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def agg(cfg, sd, prefix, x):
    B, S = x.shape[0], x.shape[-2]
    Fc = x.shape[-1] - 4
    if cfg.viewdir_agg:
        a = x[..., :-4] + F.relu(_lin(sd, prefix + ".view_fc.0", x[..., -4:]))
    else:
        a = x[..., :-4]
    var = torch.var(a, dim=-2).view(B, -1, 1, Fc).repeat(1, 1, S, 1)           # unbiased (nerf.py:82)
    avg = torch.mean(a, dim=-2).view(B, -1, 1, Fc).repeat(1, 1, S, 1)
    g = F.relu(_lin(sd, prefix + ".global_fc.0", torch.cat([a, var, avg], -1)))
    w = F.softmax(F.relu(_lin(sd, prefix + ".agg_w_fc.0", g)), dim=-2)
    return F.relu(_lin(sd, prefix + ".fc.0", (g * w).sum(-2)))


def nerf_mlp(cfg, sd, prefix, vox, x):
    B = x.shape[0]
    S = x.shape[2]
    im = agg(cfg, sd, prefix + ".agg", x)
    vi = torch.cat([vox, im], -1)
    h = F.relu(_lin(sd, prefix + ".lr0.0", vi))
    sigma = F.softplus(_lin(sd, prefix + ".sigma.0", h))
    y = torch.cat([h, vi], -1)
    y = y.view(B, -1, 1, y.shape[-1]).repeat(1, 1, S, 1)
    y = torch.cat([y, x], -1)
    c = F.relu(_lin(sd, prefix + ".color.2", F.relu(_lin(sd, prefix + ".color.0", y))))
    cw = F.softmax(c, dim=-2)
    col = torch.sum(x[..., -7:-4] * cw, dim=-2)
    return torch.cat([col, sigma], -1)


# --------------------------------------------------------------------------------------------------
# S12  compositing  (utils.py:571-603)
# --------------------------------------------------------------------------------------------------
def raw2outputs(raw, z, white_bkgd=False):
    alpha = 1.0 - torch.exp(-raw[..., 3])
    T = torch.cumprod(1.0 - alpha + 1e-10, -1)[..., :-1]
    T = torch.cat([torch.ones_like(alpha[..., :1]), T], -1)
    w = alpha * T
    rgb = torch.sum(w[..., None] * raw[..., :3], -2)
    w = F.softmax(w, -1)                                                      # utils.py:594
    depth = torch.sum(w * z, -1)
    if white_bkgd:
        rgb = rgb + (1.0 - torch.sum(w, -1)[..., None])
    return {"rgb": rgb, "depth": depth, "weights": w}


# --------------------------------------------------------------------------------------------------
# render_rays / forward  (network.py:24-43, 76-113; network_human.py:90-107)
# --------------------------------------------------------------------------------------------------
def render_rays(cfg, sd, rays, level, batch, im_feat, feat_vol, return_intermediates=False):
    cas = cfg.cas
    xyz, uvd, z = sample_along_depth(cfg, rays, cas.num_samples[level], level)
    B, n_rays, n_samp = xyz.shape[:3]
    rgbs = unpreprocess(batch["src_inps"], cas.render_scale[level])
    up = cas.render_scale[level] / cas.im_ibr_scale[level]
    if up != 1.0:                                                             # network.py:29-32
        b, s, c, h, w = im_feat.shape
        im_feat = _resize_ac(im_feat.reshape(b * s, c, h, w), up).view(b, s, c, int(h * up), int(w * up))
    tex = torch.cat([im_feat, rgbs], 2)
    Ho, Wo = batch["src_inps"].shape[-2:]
    Hr, Wr = int(Ho * cas.render_scale[level]), int(Wo * cas.render_scale[level])
    uvd = uvd.clone()
    uvd[..., 0], uvd[..., 1] = uvd[..., 0] / (Wr - 1), uvd[..., 1] / (Hr - 1)  # network.py:37
    vf = vox_feat(uvd.reshape(B, -1, 3), feat_vol)
    xf = img_feat(cfg, xyz, tex, batch, level)
    raw = nerf_mlp(cfg, sd, f"nerf_{level}", vf, xf).reshape(B, -1, n_samp, 4)
    out = raw2outputs(raw, z, cfg.white_bkgd)
    if return_intermediates:
        out = dict(out, _xyz=xyz, _uvd=uvd, _z=z, _vox=vf, _img=xf, _raw=raw, _tex=tex)
    return out


def forward(cfg, sd, batch, feats=None, intermediates=None):
    """One frame.  ``intermediates`` (a dict) is filled with every stage-boundary tensor."""
    cas = cfg.cas
    feats = feats if feats is not None else forward_feat(sd, batch["src_inps"])
    keep = intermediates if intermediates is not None else {}
    ret = {}
    depth = std = near_far = None
    for i in range(cas.num):
        vol, dv, near_far = feature_volume(cfg, feats[f"level_{i}"], batch, cas.volume_planes[i],
                                           depth, std, near_far, i)
        feat, prob = cost_reg(sd, f"cost_reg_{i}", vol)
        depth, std = depth_regression(cfg, prob, dv, i)
        keep.update({f"vol_{i}": vol, f"dv_{i}": dv, f"nf_{i}": near_far, f"feat3d_{i}": feat,
                     f"prob_{i}": prob, f"depth_{i}": depth, f"std_{i}": std})
        if not cas.render_if[i]:
            continue
        rays = build_rays(cfg, depth, std, batch, near_far, i)
        keep[f"rays12_{i}"] = rays
        masked = "mask_at_box" in batch and i == cas.num - 1                  # network_human.py:90-93
        if masked:
            m = batch["mask_at_box"].bool().reshape(1, -1)
            rays = rays[m][None]
        out = render_rays(cfg, sd, rays, i, batch, feats[f"level_{cas.render_im_feat_level[i]}"], feat)
        if masked:                                                            # network_human.py:102-107
            rgb = torch.zeros_like(batch["mask_at_box"].reshape(1, -1))[..., None].repeat(1, 1, 3).float()
            if m.sum() > 1:
                rgb[m] = out["rgb"][0]
            out["rgb"] = rgb
        out["depth_mvs"] = 1.0 / depth if cas.depth_inv[i] else depth          # network.py:105-108
        out["std"] = std
        ret.update({f"{k}_level{i}": v for k, v in out.items()})
    return ret


def train_step(cfg, sd, batch, loss_weight=(0.1, 1.0)):
    """One training step's forward + loss + backward on this restatement (BASELINE config 5): the reference network in
    ``.train()`` mode (BatchNorm batch statistics, lib/networks/enerf/utils.py:10-33), the MSE part of
    lib/train/losses/enerf.py:21-24 with ``loss_weight`` (dtu_pretrain.yaml:43), ``loss.backward()`` (trainer.py:56-63).
    Returns (loss, {parameter name: gradient}).  Pinned to the reference's own gradients (tests/golden/train_tiny.npz) by
    tests/test_oracle_golden.py; bench.py --train times it as the CPU baseline of the training step."""
    global _BN_TRAIN
    params = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v)
              for k, v in sd.items()}
    _BN_TRAIN = True
    try:
        out = forward(cfg, params, batch)
        loss = sum(loss_weight[i] * F.mse_loss(batch[f"rgb_{i}"], out[f"rgb_level{i}"]) for i in range(cfg.cas.num)
                   if cfg.cas.render_if[i])
        loss.backward()
    finally:
        _BN_TRAIN = False
    return loss.detach(), {k: v.grad for k, v in params.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}


def psnr(a, b):
    """evaluators/enerf.py PSNR formula on [0,1] images: -10 log10(mse)."""
    mse = torch.mean((a - b) ** 2)
    return float(-10.0 * torch.log10(mse))
