"""Torch-op twins of the training path's stages — TEST INFRASTRUCTURE (they used to live in the product
package, enerf_amd/train_path.py, and ran whenever no library was available: a silent eager fallback).

Each function restates the reference's expression for one stage (citations in the docstrings) in plain PyTorch ops; the tests
compare a HIP stage (forward values AND gradients) against its twin, and pin the twins themselves to the unmodified reference's
training step (tests/golden/train_*.npz).  ``install(net, *stages)`` routes the named stages of ``enerf_amd.train_path.forward_train``
through their twins (the product path has a hook for exactly this and nothing else); ``install(net)`` routes ALL of them, which is
the only way ``forward_train`` runs without the HIP library (CPU tensors, no emulator injected).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

STAGES = ("feature_net", "camera_tables", "depth_values", "feature_volume", "cost_reg", "depth_regression", "rays", "gather",
          "mlp", "composite")


def _resize_ac(x, scale, recompute=None):
    """Bilinear, align_corners=True (utils.py:115-117, 394-396, 611; network.py:32)."""
    kw = {} if recompute is None else {"recompute_scale_factor": recompute}
    return F.interpolate(x, None, scale_factor=scale, mode="bilinear", align_corners=True, **kw)



def agg_forward(m, x):
    """Agg (nerf.py:74-89): x (B,P,S,F+4) -> (B,P,16)."""
    Fc = m.feat_ch
    S = x.shape[-2]
    a = x[..., :Fc]
    if hasattr(m, "view_fc"):
        a = a + m.view_fc(x[..., Fc:])
    var = torch.var(a, dim=-2, keepdim=True).expand(-1, -1, S, -1)            # unbiased (nerf.py:82)
    avg = torch.mean(a, dim=-2, keepdim=True).expand(-1, -1, S, -1)
    g = m.global_fc(torch.cat([a, var, avg], -1))
    w = F.softmax(m.agg_w_fc(g), dim=-2)
    return m.fc((g * w).sum(-2))


def nerf_forward(m, vox, x):
    """NeRF (nerf.py:29-43): vox (B,P,8), x (B,P,S,F+4) -> (B,P,4) = [rgb, sigma]."""
    S = x.shape[2]
    im = agg_forward(m.agg, x)
    vi = torch.cat([vox, im], -1)
    h = m.lr0(vi)
    sigma = m.sigma(h)
    y = torch.cat([h, vi], -1).unsqueeze(2).expand(-1, -1, S, -1)
    c = m.color(torch.cat([y, x], -1))
    cw = F.softmax(c, dim=-2)
    col = torch.sum(x[..., -7:-4] * cw, dim=-2)
    return torch.cat([col, sigma], -1)


def _clamp_pair(a, b, lo_bound, hi_bound, inv: bool):
    """utils.py:122-127 / 400-413: stack([a, b]); masked in-place replacement by the (detached) volume bounds."""
    if inv:       # disparity space: first entry clamped from above by bound 0, second from below by bound 1
        a = torch.where(a > lo_bound, lo_bound, a)
        b = torch.where(b < hi_bound, hi_bound, b)
    else:
        a = torch.where(a < lo_bound, lo_bound, a)
        b = torch.where(b > hi_bound, hi_bound, b)
    return a, b


def depth_values(cas, batch, level, D, depth, std, near_far):
    """get_depth_values (utils.py:98-151) -> depth_values (B,D,h,w), near_far (B,2,h,w) (detached)."""
    nf = batch["near_far"]
    B = nf.shape[0]
    H, W = batch["src_inps"].shape[-2:]
    h, w = int(H * cas.volume_scale[level]), int(W * cas.volume_scale[level])
    t = torch.linspace(0.0, 1.0, steps=D, device=nf.device, dtype=torch.float32)
    if depth is None:
        tt = t.view(1, -1)
        if cas.depth_inv[level]:
            dv = 1.0 / (1.0 / nf[:, :1] + tt * (1.0 / nf[:, 1:] - 1.0 / nf[:, :1]))
        else:
            dv = nf[:, :1] + (nf[:, 1:] - nf[:, :1]) * tt
        dv = dv.view(B, D, 1, 1).repeat(1, 1, h, w)
    else:
        k = cas.volume_scale[level] / cas.volume_scale[level - 1]
        if k != 1.0:
            depth = _resize_ac(depth[:, None], k, True)[:, 0]
            std = _resize_ac(std[:, None], k, True)[:, 0]
            near_far = _resize_ac(near_far, k, True)
        if not cas.depth_inv[level - 1]:
            raise RuntimeError("cascade levels after a depth-space level are undefined in the reference (utils.py:130)")
        lo, hi = _clamp_pair(depth + std, depth - std, near_far[:, 0], near_far[:, 1], True)
        nn_, ff_ = 1.0 / lo, 1.0 / hi
        tt = t.view(1, D, 1, 1)
        if cas.depth_inv[level]:
            dv = 1.0 / (1.0 / nn_[:, None] + tt * (1.0 / ff_[:, None] - 1.0 / nn_[:, None]))
        else:
            dv = nn_[:, None] + tt * (ff_[:, None] - nn_[:, None])
    out_nf = torch.stack([dv[:, 0], dv[:, -1]], 1).detach()             # (index lists become host->device copies: not capturable)
    if cas.depth_inv[level]:
        out_nf = 1.0 / torch.clamp_min(out_nf, 1e-6)
    return dv.contiguous(), out_nf


def proj_mats(batch, src_scale, tar_scale):
    """get_proj_mats (utils.py:35-55) -> (B,S,3,4)."""
    B, S = batch["src_exts"].shape[:2]
    Ks = batch["src_ixts"].clone()
    Ks[:, :, :2] *= src_scale
    src = Ks @ batch["src_exts"][:, :, :3]
    Kt = batch["tar_ixt"].clone()
    Kt[:, :2] *= tar_scale
    tar = Kt @ batch["tar_ext"][:, :3]
    last = torch.zeros(B, 1, 4, device=tar.device, dtype=tar.dtype)
    last[:, :, 3] = 1
    return src @ torch.inverse(torch.cat([tar, last], 1))[:, None]


def feature_volume(feats_level, proj, dv):
    """homo_warp for every view at once + biased variance (utils.py:57-95, 322-349): feats (B,S,C,Hs,Ws) -> (B,C,D,h,w)."""
    B, S, C, Hs, Ws = feats_level.shape
    _, D, h, w = dv.shape
    dev = dv.device
    ys, xs = torch.meshgrid(torch.linspace(0, h - 1, h, device=dev), torch.linspace(0, w - 1, w, device=dev), indexing="ij")
    g = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(h * w, device=dev)], 0)          # (3, hw)
    R, T = proj[..., :3], proj[..., 3:]                                                          # (B,S,3,3), (B,S,3,1)
    rot = (R @ g).unsqueeze(3)                                                                   # (B,S,3,1,hw)
    p = rot + T.unsqueeze(-1) / dv.reshape(B, 1, 1, D, h * w)                                    # (B,S,3,D,hw)
    uv = p[:, :, :2] / torch.clamp_min(p[:, :, 2:], 1e-6)
    gx = uv[:, :, 0] / ((Ws - 1) / 2) - 1
    gy = uv[:, :, 1] / ((Hs - 1) / 2) - 1
    grid = torch.stack([gx, gy], -1).reshape(B * S, D, h * w, 2)
    warped = F.grid_sample(feats_level.reshape(B * S, C, Hs, Ws), grid, mode="bilinear", padding_mode="zeros",
                           align_corners=True).view(B, S, C, D, h, w)
    mean = warped.mean(1)
    return (warped ** 2).mean(1) - mean ** 2


def depth_regression(cas, prob, dv, level):
    """utils.py:658-667."""
    p = F.softmax(prob, 1)
    v = 1.0 / torch.clamp_min(dv, 1e-6) if cas.depth_inv[level] else dv
    mu = torch.sum(p * v, 1)
    var = (p * (v - mu.unsqueeze(1)) ** 2).sum(1)
    return mu, torch.clamp_min(var, 1e-10).sqrt()


def build_rays(cas, depth, std, rays, near_far, level):
    """utils.py:390-420 -> (B,N,12)."""
    k = cas.render_scale[level] / cas.volume_scale[level]
    if k != 1.0:
        depth = _resize_ac(depth[:, None], k)[:, 0]
        std = _resize_ac(std[:, None], k)[:, 0]
        near_far = _resize_ac(near_far, k)
    if cas.depth_inv[level]:
        rn, rf = _clamp_pair(depth + std, depth - std, near_far[:, 0], near_far[:, 1], True)
    else:
        rn, rf = _clamp_pair(depth - std, depth + std, near_far[:, 0], near_far[:, 1], False)
    B, N = rays.shape[:2]
    uv = rays[:, :, 6:].long()
    flat = uv[..., 1] * depth.shape[-1] + uv[..., 0]                     # (B,N) index into the flattened (h*w) maps
    # m[b, v, u] as a gather on the flattened map: same values as the reference's advanced indexing (utils.py:414-417), but its
    # backward is a scatter-add instead of index_put_(accumulate=True), which sorts the 327,680 indices of a full-image level
    pick = lambda m: m.reshape(B, -1).gather(1, flat)
    return torch.cat([rays, pick(rn)[..., None], pick(rf)[..., None], pick(near_far[:, 0])[..., None],
                      pick(near_far[:, 1])[..., None]], -1)


def sample_along_depth(cas, rays, n_samples, level):
    """utils.py:422-441."""
    o, d, uv = rays[..., :3], rays[..., 3:6], rays[..., 6:8]
    rn, rf, vn, vf = rays[..., 8:9], rays[..., 9:10], rays[..., 10:11], rays[..., 11:12]
    if n_samples == 1:
        z = rn + (rf - rn) * 0.5
    else:
        z = rn + (rf - rn) * torch.linspace(0.0, 1.0, n_samples, device=rays.device)[None, None]
    if cas.depth_inv[level]:
        xyz = o[..., None, :] + d[..., None, :] * (1 / torch.clamp_min(z[..., None], 1e-6))
        dn = (vn - z) / torch.clamp_min(vn - vf, 1e-6)
    else:
        xyz = o[..., None, :] + d[..., None, :] * z[..., None]
        dn = (z - vn) / torch.clamp_min(vf - vn, 1e-6)
    uvd = torch.cat([uv[..., None, :].expand(-1, -1, n_samples, -1), dn[..., None]], -1)
    return xyz, uvd, z


def img_feat(cas, xyz, tex, batch, level):
    """get_img_feat (utils.py:689-722), all views at once: xyz (B,N,Ns,3), tex (B,S,C,H,W) -> (B,N*Ns,S,C+4)."""
    B, S, C, H, W = tex.shape
    P = xyz.shape[1] * xyz.shape[2]
    p = xyz.reshape(B, 1, P, 3)
    E = batch["src_exts"]                                                       # (B,S,4,4)
    cam = p @ E[:, :, :3, :3].transpose(-1, -2) + E[:, :, None, :3, 3]          # (B,S,P,3)
    K = batch["src_ixts"].clone()
    K[:, :, :2] *= cas.render_scale[level]
    pix = cam @ K.transpose(-1, -2)
    g = pix[..., :2] / torch.clamp_min(pix[..., 2:], 1e-6)
    g = torch.stack([g[..., 0] / (W - 1), g[..., 1] / (H - 1)], -1) * 2.0 - 1.0
    f = F.grid_sample(tex.reshape(B * S, C, H, W), g.reshape(B * S, 1, P, 2), align_corners=True, mode="bilinear",
                      padding_mode="border").view(B, S, C, P).permute(0, 3, 1, 2)           # (B,P,S,C)
    ct = torch.inverse(batch["tar_ext"])[:, :3, 3]                              # (B,3)
    cs = torch.inverse(E)[:, :, :3, 3]                                          # (B,S,3)
    dt = xyz.reshape(B, P, 1, 3) - ct[:, None, None]
    ds = xyz.reshape(B, P, 1, 3) - cs[:, None]
    dt = dt / (torch.norm(dt, dim=-1, keepdim=True) + 1e-6)
    ds = ds / (torch.norm(ds, dim=-1, keepdim=True) + 1e-6)
    df = dt - ds
    dirc = df / torch.clamp(torch.norm(df, dim=-1, keepdim=True), min=1e-6)
    dot = torch.sum(dt * ds, -1, keepdim=True)
    return torch.cat([f, dirc, dot.expand(-1, -1, S, -1)], -1)


def raw2outputs(raw, z, white_bkgd=False):
    """utils.py:571-603."""
    alpha = 1.0 - torch.exp(-raw[..., 3])
    T = torch.cumprod(1.0 - alpha + 1e-10, -1)[..., :-1]
    T = torch.cat([torch.ones_like(alpha[..., :1]), T], -1)
    w = alpha * T
    rgb = torch.sum(w[..., None] * raw[..., :3], -2)
    w = F.softmax(w, -1)
    depth = torch.sum(w * z.detach(), -1)             # utils.py:595: z_vals.detach()
    if white_bkgd:
        rgb = rgb + (1.0 - torch.sum(w, -1)[..., None])
    return {"rgb": rgb, "depth": depth, "weights": w}




def feature_net_forward(m, x):
    """FeatureNet.forward (feature_net.py:27-36) on the ``FeatureNet`` parameter module, plain torch modules."""
    def cbr(blk, t):
        return F.relu(blk.bn(blk.conv(t)), inplace=True)
    c0 = cbr(m.conv0[1], cbr(m.conv0[0], x))
    c1 = cbr(m.conv1[1], cbr(m.conv1[0], c0))
    c2 = cbr(m.conv2[1], cbr(m.conv2[0], c1))
    f2 = m.toplayer(c2)
    f1 = m._up2(f2) + m.lat1(c1)
    f0 = m._up2(f1) + m.lat0(c0)
    return f2, m.smooth1(f1), m.smooth0(f0)


def cost_reg_forward(m, x):
    """MinCostRegNet / CostRegNet (cost_reg_net.py:35-48, 75-86) on ``CostRegParams``, plain torch modules."""
    def cbr(blk, t):
        return F.relu(blk.bn(blk.conv(t)), inplace=True)

    def up(seq, t):
        return seq[1](seq[0](t))
    c0 = cbr(m.conv0, x)
    c2 = cbr(m.conv2, cbr(m.conv1, c0))
    c4 = cbr(m.conv4, cbr(m.conv3, c2))
    y = c4
    if m.full:
        y = cbr(m.conv6, cbr(m.conv5, c4))
        y = c4 + up(m.conv7, y)
    y = c2 + up(m.conv9, y)
    y = c0 + up(m.conv11, y)
    return m.feat_conv[0](y), m.depth_conv[0](y).squeeze(1)


def gather_cameras(batch, render_scale: float):
    """autograd.gather_cameras (enerf_camera_tables) as torch ops (torch.inverse synchronises): K'E33 | K't | source centre | 0
    per view and the target centre (utils.py:697-704), products in fp64, stored fp32."""
    E = batch["src_exts"].double()
    K = batch["src_ixts"].double().clone()
    K[:, :, :2] *= render_scale
    B, S = E.shape[:2]
    M = K @ E[:, :, :3, :3]
    v = (K @ E[:, :, :3, 3:4])[..., 0]
    cs = torch.inverse(E)[:, :, :3, 3]
    z1 = torch.zeros(B, S, 1, dtype=torch.float64, device=E.device)
    cam = torch.cat([M.reshape(B, S, 9), v, cs, z1], -1).float().contiguous()
    ct = torch.inverse(batch["tar_ext"].double())[:, :3, 3]
    tcen = torch.cat([ct, z1[:, 0]], -1).float().contiguous()
    return cam, tcen


def camera_tables(cas, batch) -> Dict[str, torch.Tensor]:
    t: Dict[str, torch.Tensor] = {}
    for i in range(cas.num):
        t[f"proj_{i}"] = proj_mats(batch, cas.im_feat_scale[i], cas.volume_scale[i])
        if cas.render_if[i]:
            t[f"cam_{i}"], t["tcen"] = gather_cameras(batch, cas.render_scale[i])
    return t


def rays_twin(cas, depth, std, rays, near_far, level, n_samples):
    """build_rays + sample_along_depth (utils.py:390-441) -> (z, xyz, dn, uv) as RaySamplesFn returns them."""
    rays12 = build_rays(cas, depth, std, rays, near_far, level)
    xyz, uvd, z = sample_along_depth(cas, rays12, n_samples, level)
    return z, xyz, uvd[..., 2], uvd[..., :2]


def gather_twin(cas, xyz, dn, uv, im_feat, batch, feat_vol, level):
    """unpreprocess + cat (network.py:30-34), get_vox_feat (utils.py:456-458) and get_img_feat (utils.py:689-722) -> (x, vox)."""
    src = batch["src_inps"]
    B, S, _, H, W = src.shape
    N, Ns = xyz.shape[1], xyz.shape[2]
    rs = cas.render_scale[level]
    Hr, Wr = int(H * rs), int(W * rs)
    up = rs / cas.im_ibr_scale[level]
    rgbs = _resize_ac((src * 0.5 + 0.5).reshape(B * S, 3, H, W), rs, True).reshape(B, S, 3, Hr, Wr)
    if up != 1.0:
        b, s, c, h, w = im_feat.shape
        im_feat = _resize_ac(im_feat.reshape(b * s, c, h, w), up).view(b, s, c, int(h * up), int(w * up))
    tex = torch.cat([im_feat, rgbs], 2)
    uvd = torch.cat([uv, dn[..., None]], -1)
    nd = torch.stack([uvd[..., 0] / (Wr - 1), uvd[..., 1] / (Hr - 1), uvd[..., 2]], -1)      # network.py:36-38
    g = nd.reshape(B, 1, 1, N * Ns, 3) * 2.0 - 1.0
    vox = F.grid_sample(feat_vol, g, align_corners=True)[:, :, 0, 0].permute(0, 2, 1)        # get_vox_feat utils.py:456-458
    return img_feat(cas, xyz, tex, batch, level), vox


TWINS = {
    "feature_net": feature_net_forward, "camera_tables": camera_tables, "depth_values": depth_values,
    "feature_volume": feature_volume, "cost_reg": cost_reg_forward, "depth_regression": depth_regression,
    "rays": rays_twin, "gather": gather_twin, "mlp": nerf_forward, "composite": raw2outputs,
}


def install(net, *stages, **overrides):
    """Route the named stages of the network's training forward through their torch twins (no name: every stage — the path then
    needs no library).  ``overrides`` replace a twin by a caller's function (perturbation experiments)."""
    names = stages or (() if overrides else STAGES)
    tw = dict(getattr(net, "_stage_twins", None) or {})
    for s in names:
        tw[s] = TWINS[s]
    tw.update(overrides)
    net._stage_twins = tw
    return net


def uninstall(net):
    net._stage_twins = None
    return net
