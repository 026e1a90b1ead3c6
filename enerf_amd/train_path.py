"""Training-mode forward of the ENeRF path (config 5: ``train_net.py`` under DDP).

``Network.forward`` in ``.train()`` mode runs this differentiable path, so ``trainer.py:56-63`` (loss, ``backward``,
``clip_grad_value_``, ``optimizer.step``) and ``trainer.py:15-22`` (``SyncBatchNorm.convert_sync_batchnorm`` + DDP over
RCCL: the gradient all-reduce of the 436,012 parameters) work unchanged on the drop-in network.

What runs where in training (state of this round, stated plainly):
  * every stage is expressed on the network's OWN parameter modules (``feature_net``, ``cost_reg_i.conv*.{conv,bn}``,
    ``nerf_i.*``), so BatchNorm uses batch statistics (SyncBatchNorm under the reference trainer), the running statistics
    are updated exactly like the reference's, and autograd sees every parameter;
  * HIP kernels in BOTH directions (``enerf_amd/autograd.py``): the whole FeatureNet (every convolution and the input
    gradients of its stride-1 layers on the inference path's MFMA kernel, BatchNorm2d-train on the channel kernels, weight
    gradients on the matrix cores, channels-last from the image to the three output maps: ``FeatureNetTrainFn``), the
    cost-volume warp + variance (feature scatter-add and the depth gradient through the warp grid), both
    cost-regularisation networks end to end (MFMA convolutions and input gradients, BatchNorm-train kernels, MFMA weight
    gradients), depth regression, the render-side fetches (bilinear texel + trilinear volume gathers, direction code), the
    Agg + NeRF MLP (fused forward, fused recompute-backward, weight gradients as position reductions on the matrix cores)
    and alpha compositing;
  * still PyTorch-ROCm ops under autograd: the input gradients of the FeatureNet's two stride-2 5x5 convolutions (a
    transposed 5x5 convolution) and the per-ray geometry glue.  With frozen
    BatchNorm (``bn.eval()`` fine-tuning) the FeatureNet / cost-reg nets run through their modules instead (library
    convolutions, HIP weight gradients).  This is GPU code (no CPU fallback, nothing from ``oracle/``), checked against the
    reference's own gradients (tests/test_training.py, tests/golden/train_tiny.npz) with the HIP stages switched on and off.
The inference path (eval mode) never comes here: it is the single ``enerf_forward`` C call.

Semantics follow the reference line by line where gradients are concerned: the in-place masked clamps of
``get_depth_values`` / ``build_rays`` (utils.py:122-127, 400-413: gradient flows through the un-clamped branch only),
``near_far`` detached (utils.py:148), disparity-space level 0, ``grid_sample`` gradients w.r.t. both the features and the
sampling grid (level 1's warp grid back-propagates into level 0's depth/std).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F


def _resize_ac(x, scale, recompute=None):
    """Bilinear, align_corners=True (utils.py:115-117, 394-396, 611; network.py:32)."""
    kw = {} if recompute is None else {"recompute_scale_factor": recompute}
    return F.interpolate(x, None, scale_factor=scale, mode="bilinear", align_corners=True, **kw)


# ---------------------------------------------------------------------------------------------------------------------
# parameter-module forwards (train or eval mode: whatever the modules are in)
# ---------------------------------------------------------------------------------------------------------------------
def _conv(lib, m, t):
    """A convolution module of the network; with the HIP library its weight gradient runs on the matrix cores."""
    if lib is None:
        return m(t)
    from .autograd import conv_module
    return conv_module(lib, m, t)


def feature_net_forward(m, x, lib=None, hip_train=True):
    """FeatureNet.forward (feature_net.py:27-36) on the ``FeatureNet`` parameter module.  With the HIP library and every
    BatchNorm2d in training mode the whole net runs on the HIP kernels in both directions (autograd.FeatureNetTrainFn);
    otherwise (frozen BatchNorm, ``hip_train=False``, no library) through the modules, convolutions through ``_conv``."""
    if lib is not None and hip_train and all(b.training for b in m.modules() if isinstance(b, torch.nn.modules.batchnorm._BatchNorm)):
        from .autograd import feature_net_train
        return feature_net_train(lib, m, x)

    def cbr(blk, t):
        return F.relu(blk.bn(_conv(lib, blk.conv, t)), inplace=True)
    c0 = cbr(m.conv0[1], cbr(m.conv0[0], x))
    c1 = cbr(m.conv1[1], cbr(m.conv1[0], c0))
    c2 = cbr(m.conv2[1], cbr(m.conv2[0], c1))
    f2 = _conv(lib, m.toplayer, c2)
    f1 = m._up2(f2) + _conv(lib, m.lat1, c1)
    f0 = m._up2(f1) + _conv(lib, m.lat0, c0)
    return f2, _conv(lib, m.smooth1, f1), _conv(lib, m.smooth0, f0)


def cost_reg_forward(m, x, lib=None):
    """MinCostRegNet / CostRegNet (cost_reg_net.py:35-48, 75-86) on ``CostRegParams``: x (B,C,D,h,w) -> feat, prob."""
    def cbr(blk, t):
        return F.relu(blk.bn(_conv(lib, blk.conv, t)), inplace=True)

    def up(seq, t):
        return seq[1](_conv(lib, seq[0], t))
    c0 = cbr(m.conv0, x)
    c2 = cbr(m.conv2, cbr(m.conv1, c0))
    c4 = cbr(m.conv4, cbr(m.conv3, c2))
    y = c4
    if m.full:
        y = cbr(m.conv6, cbr(m.conv5, c4))
        y = c4 + up(m.conv7, y)
    y = c2 + up(m.conv9, y)
    y = c0 + up(m.conv11, y)
    return _conv(lib, m.feat_conv[0], y), _conv(lib, m.depth_conv[0], y).squeeze(1)


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


# ---------------------------------------------------------------------------------------------------------------------
# geometry
# ---------------------------------------------------------------------------------------------------------------------
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


def camera_tables(cas, batch, lib=None) -> Dict[str, torch.Tensor]:
    """Everything the step derives from the cameras alone (the matrix inverses live here): the warp matrices of each
    level and the per-view constants of the render-side fetches.  forward_train computes it unless the batch already
    carries it under "camera_tables".  With the library the 4x4 inverses run on the device (enerf_get_proj_mats,
    enerf_camera_tables: 2-4 launches, capturable); the torch twin (torch.inverse: synchronises) stays for lib=None."""
    from .autograd import gather_cameras
    t: Dict[str, torch.Tensor] = {}
    for i in range(cas.num):
        if lib is not None:      # the inference path's kernel (fp64 inverse on the device: no host synchronisation)
            t[f"proj_{i}"] = lib.get_proj_mats(batch["src_ixts"], batch["src_exts"], batch["tar_ixt"], batch["tar_ext"],
                                               cas.im_feat_scale[i], cas.volume_scale[i])
        else:
            t[f"proj_{i}"] = proj_mats(batch, cas.im_feat_scale[i], cas.volume_scale[i])
        if cas.render_if[i]:
            t[f"cam_{i}"], t["tcen"] = gather_cameras(batch, cas.render_scale[i], lib)
    return t


def render_rays(net, rays, level, batch, im_feat, feat_vol, lib=None, tables=None, maps=None):
    """Network.render_rays (network.py:24-43), differentiable: rays (B,N,12), im_feat (B,S,C,Hf,Wf), feat_vol (B,8,D,h,w).
    ``maps`` = (depth, std, near_far) of the level: with the library, build_rays + sample_along_depth run as ONE kernel pair
    (RaySamplesFn) on the 8-float rays and ``rays`` may be the batch's (B,N,8) list."""
    cas = net.cfg.cas
    Ns = cas.num_samples[level]
    B, N = rays.shape[:2]
    src = batch["src_inps"]
    S, H, W = src.shape[1], src.shape[-2], src.shape[-1]
    rs = cas.render_scale[level]
    Hr, Wr = int(H * rs), int(W * rs)
    up = rs / cas.im_ibr_scale[level]
    hip_geo = lib is not None and maps is not None and getattr(net, "hip_geometry", True)
    if hip_geo:
        from .autograd import RaySamplesFn
        z, xyz, dn, uv = RaySamplesFn.apply(lib, maps[0], maps[1], maps[2], rays, Ns, Hr, Wr, bool(cas.depth_inv[level]))
        uvd = None
    else:
        xyz, uvd, z = sample_along_depth(cas, rays, Ns, level)
        dn, uv = uvd[..., 2], uvd[..., :2]
    hip_gather = lib is not None and getattr(net, "hip_gather", True)
    if hip_gather and hip_geo and up == 1.0 and tuple(im_feat.shape[-2:]) == (Hr, Wr):
        from .autograd import TexelsFn
        tex_cl = TexelsFn.apply(lib, im_feat, src, Hr, Wr)                 # channels-last texels, one launch
        tex = None
    else:
        rgbs = _resize_ac((src * 0.5 + 0.5).reshape(B * S, 3, H, W), rs, True).reshape(B, S, 3, Hr, Wr)
        if up != 1.0:
            b, s, c, h, w = im_feat.shape
            im_feat = _resize_ac(im_feat.reshape(b * s, c, h, w), up).view(b, s, c, int(h * up), int(w * up))
        tex = torch.cat([im_feat, rgbs], 2)
        tex_cl = tex.permute(0, 1, 3, 4, 2) if hip_gather else None
    if hip_gather:
        from .autograd import GatherFn, gather_cameras
        cam, tcen = (tables[f"cam_{level}"], tables["tcen"]) if tables is not None else gather_cameras(batch, rs, lib)
        # a full image of rays (train_img, dtu_pretrain.yaml:41; enerf_utils.py:61-71 emits them row-major): the backward owns
        # 2-D ray tiles (a hint: a permuted list of Hr * Wr rays is still scattered correctly, only slower)
        x, vox = GatherFn.apply(lib, xyz.reshape(B, N * Ns, 3), dn.reshape(B, N * Ns), uv.reshape(B, N * Ns, 2), tex_cl,
                                feat_vol.permute(0, 2, 3, 4, 1), cam, tcen, Ns, Wr if N == Hr * Wr else 0)
    else:
        if uvd is None:
            uvd = torch.cat([uv, dn[..., None]], -1)
        nd = torch.stack([uvd[..., 0] / (Wr - 1), uvd[..., 1] / (Hr - 1), uvd[..., 2]], -1)      # network.py:36-38
        g = nd.reshape(B, 1, 1, N * Ns, 3) * 2.0 - 1.0
        vox = F.grid_sample(feat_vol, g, align_corners=True)[:, :, 0, 0].permute(0, 2, 1)        # get_vox_feat utils.py:456-458
        x = img_feat(cas, xyz, tex, batch, level)
    if lib is not None and getattr(net, "hip_mlp_backward", True):
        from .autograd import nerf_mlp
        raw = nerf_mlp(lib, getattr(net, f"nerf_{level}"), nerf_forward, vox, x).reshape(B, N, Ns, 4)   # fused HIP backward
    else:
        raw = nerf_forward(getattr(net, f"nerf_{level}"), vox, x).reshape(B, N, Ns, 4)
    if lib is not None and getattr(net, "hip_composite", True):
        from .autograd import CompositeFn
        rgb, depth, weights = CompositeFn.apply(lib, raw, z, bool(net.cfg.white_bkgd))
        return {"rgb": rgb, "depth": depth, "weights": weights}
    return raw2outputs(raw, z, net.cfg.white_bkgd)


def _hip_lib(net, t: torch.Tensor):
    """The library whose kernels can run on ``t``'s memory: the GPU build for CUDA tensors, a CPU lane-emulator build only
    when a test injected one (``Network(lib=...)``); None -> the stage runs as torch ops."""
    if not getattr(net, "hip_backward", True):
        return None
    if t.is_cuda:
        return net.lib
    lib = net._lib
    return lib if (lib is not None and "emu" in lib.path) else None


def forward_train(net, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Network.forward (network.py:76-113 / network_human.py:69-119) with autograd, on the network's own modules."""
    from .autograd import CompositeFn, DepthRegressionFn, FeatureVolumeFn
    cas = net.cfg.cas
    lib = _hip_lib(net, batch["src_inps"])
    src = batch["src_inps"]
    B, S, _, H, W = src.shape
    f2, f1, f0 = feature_net_forward(net.feature_net, src.view(B * S, 3, H, W), lib,
                                     getattr(net, "hip_feature_net_train", True))                 # network.py:58-67
    feats = {"level_2": f0.reshape(B, S, f0.shape[1], H, W), "level_1": f1.reshape(B, S, f1.shape[1], H // 2, W // 2),
             "level_0": f2.reshape(B, S, f2.shape[1], H // 4, W // 4)}
    ret: Dict[str, torch.Tensor] = {}
    depth: Optional[torch.Tensor] = None
    std = near_far = None
    tables = batch["camera_tables"] if "camera_tables" in batch else camera_tables(cas, batch, lib)
    hip_geo = lib is not None and getattr(net, "hip_geometry", True)
    for i in range(cas.num):
        if hip_geo:      # get_depth_values on the inference kernel; level > 0 differentiable in the previous depth / std
            from .autograd import DepthValuesFn
            hv, wv = int(H * cas.volume_scale[i]), int(W * cas.volume_scale[i])
            if depth is None:
                dv, near_far = lib.get_depth_values(batch["near_far"], None, B, cas.volume_planes[i], hv, wv, cas.depth_inv[i])
            else:
                if not cas.depth_inv[i - 1]:
                    raise RuntimeError("cascade levels after a depth-space level are undefined in the reference (utils.py:130)")
                dv, near_far = DepthValuesFn.apply(lib, depth, std, near_far, batch["near_far"], cas.volume_planes[i], hv, wv,
                                                   bool(cas.depth_inv[i]))
        else:
            dv, near_far = depth_values(cas, batch, i, cas.volume_planes[i], depth, std, near_far)
        P = tables[f"proj_{i}"]
        vol = FeatureVolumeFn.apply(lib, feats[f"level_{i}"], P, dv) if (lib is not None and getattr(net, "hip_volume", True)) \
            else feature_volume(feats[f"level_{i}"], P, dv)
        reg = getattr(net, f"cost_reg_{i}")
        # the HIP training blocks normalise with BATCH statistics: only when every BatchNorm of the net is in training mode;
        # frozen-BN fine-tuning (bn.eval()) goes through the modules, which honour running statistics
        bn_batch = all(m.training for m in reg.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm))
        if lib is not None and getattr(net, "hip_cost_reg_train", True) and bn_batch:
            from .autograd import cost_reg_train
            feat3d, prob = cost_reg_train(lib, getattr(net, f"cost_reg_{i}"), vol)     # conv/BN forward + backward on HIP kernels
        else:
            feat3d, prob = cost_reg_forward(getattr(net, f"cost_reg_{i}"), vol, lib)
        depth, std = DepthRegressionFn.apply(lib, prob, dv, bool(cas.depth_inv[i])) if (lib is not None and getattr(net, "hip_depth_regression", True)) \
            else depth_regression(cas, prob, dv, i)
        if not cas.render_if[i]:
            continue
        # network_human.py:90 only compacts rays in eval mode (`not self.training`): training renders every ray
        if hip_geo:      # build_rays + sample_along_depth inside render_rays, as one kernel pair on the 8-float rays
            out = render_rays(net, batch[f"rays_{i}"], i, batch, feats[f"level_{cas.render_im_feat_level[i]}"], feat3d, lib, tables,
                              maps=(depth, std, near_far))
        else:
            rays = build_rays(cas, depth, std, batch[f"rays_{i}"], near_far, i)
            out = render_rays(net, rays, i, batch, feats[f"level_{cas.render_im_feat_level[i]}"], feat3d, lib, tables)
        if cas.depth_inv[i] and lib is not None:
            from .autograd import ReciprocalFn
            out["depth_mvs"] = ReciprocalFn.apply(lib, depth)
        else:
            out["depth_mvs"] = 1.0 / depth if cas.depth_inv[i] else depth
        out["std"] = std
        ret.update({f"{k}_level{i}": v for k, v in out.items()})
    return ret
