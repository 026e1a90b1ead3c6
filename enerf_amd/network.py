"""Drop-in ``Network`` for the reference's ``lib/networks/enerf/network.py`` (and ``network_human.py``).

Same constructor role, same ``forward(batch) -> dict`` / ``render_rays(rays, **kwargs)`` /
``batchify_rays`` / ``forward_feat`` surface and the same ``state_dict`` names and shapes
(SURVEY.md §8b), so ``run.py`` / ``gui_human.py`` / ``net_utils.load_network(strict=True)`` work
unchanged (INTEGRATION.md shows the 6-line module a maintainer drops into ``lib/networks/``).

What runs where:
  * ``feature_net`` (2-D FPN, feature_net.py:4-36)  — PyTorch-ROCm, as BASELINE.json's north_star asks;
  * everything else (network.py:80-112)            — hand-written HIP kernels through the C ABI
    (``enerf_amd/lib.py`` -> ``libenerf_hip.so``).  The ``cost_reg_*`` / ``nerf_*`` sub-modules here only
    OWN the parameters (so checkpoints, ``.cuda()``, ``SyncBatchNorm.convert_sync_batchnorm`` keep
    working); their weights are re-laid-out once per load into MFMA operand images by device kernels.
There is no eager fallback: without the built library ``forward`` raises.
Training (autograd through the HIP path) is the first "next" row of SURVEY.md §8f and not built yet:
``forward`` raises in ``.train()`` mode instead of silently computing something else.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import EnerfConfig
from .lib import ConvBn, CostRegRaw, EnerfLib, FeatNetRaw, NerfRaw, Options, get_lib


# --------------------------------------------------------------------------------------------------
# parameter containers (names/shapes == reference state_dict)
# --------------------------------------------------------------------------------------------------
class _ConvBnAct2d(nn.Module):
    """utils.py:10-20 (``conv`` / ``bn`` attribute names are part of the checkpoint format)."""

    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class FeatureNet(nn.Module):
    """feature_net.py:4-36 — the part of the frame that stays in PyTorch-ROCm."""

    def __init__(self):
        super().__init__()
        def pair(cin, cout, k, s, p):
            return nn.Sequential(_ConvBnAct2d(cin, cout, k, s, p), _ConvBnAct2d(cout, cout, 3, 1, 1))
        self.conv0 = pair(3, 8, 3, 1, 1)
        self.conv1 = pair(8, 16, 5, 2, 2)
        self.conv2 = pair(16, 32, 5, 2, 2)
        self.toplayer = nn.Conv2d(32, 32, 1)
        self.lat1 = nn.Conv2d(16, 32, 1)
        self.lat0 = nn.Conv2d(8, 32, 1)
        self.smooth1 = nn.Conv2d(32, 16, 3, padding=1)
        self.smooth0 = nn.Conv2d(32, 8, 3, padding=1)

    @staticmethod
    def _up2(x):
        return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)

    def raw(self) -> FeatNetRaw:
        """Parameter pointers for the HIP FeatureNet (enerf_feature_net_pack)."""
        r = FeatNetRaw()
        blocks = [self.conv0[0], self.conv0[1], self.conv1[0], self.conv1[1], self.conv2[0], self.conv2[1]]
        for i, m in enumerate(blocks):
            for t in (m.conv.weight, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var):
                _require_f32c(t)
            r.conv[i] = ConvBn(m.conv.weight.data_ptr(), m.bn.weight.data_ptr(), m.bn.bias.data_ptr(),
                               m.bn.running_mean.data_ptr(), m.bn.running_var.data_ptr())
        for name in ("toplayer", "lat1", "lat0", "smooth1", "smooth0"):
            m = getattr(self, name)
            _require_f32c(m.weight), _require_f32c(m.bias)
            setattr(r, name + "_w", m.weight.data_ptr())
            setattr(r, name + "_b", m.bias.data_ptr())
        return r

    def forward(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        f2 = self.toplayer(c2)
        f1 = self._up2(f2) + self.lat1(c1)
        f0 = self._up2(f1) + self.lat0(c0)
        return f2, self.smooth1(f1), self.smooth0(f0)


class _ConvBn3d(nn.Module):
    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn = nn.BatchNorm3d(cout)


def _deconv_bn3d(cin, cout):
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, 3, padding=1, output_padding=1, stride=2, bias=False),
                         nn.BatchNorm3d(cout))


class CostRegParams(nn.Module):
    """Parameters of CostRegNet (full=True, cost_reg_net.py:4-33) / MinCostRegNet (:51-73)."""

    def __init__(self, in_channels: int, full: bool):
        super().__init__()
        self.in_channels, self.full = in_channels, full
        self.conv0 = _ConvBn3d(in_channels, 8)
        self.conv1 = _ConvBn3d(8, 16, 2)
        self.conv2 = _ConvBn3d(16, 16)
        self.conv3 = _ConvBn3d(16, 32, 2)
        self.conv4 = _ConvBn3d(32, 32)
        if full:
            self.conv5 = _ConvBn3d(32, 64, 2)
            self.conv6 = _ConvBn3d(64, 64)
            self.conv7 = _deconv_bn3d(64, 32)
        self.conv9 = _deconv_bn3d(32, 16)
        self.conv11 = _deconv_bn3d(16, 8)
        self.depth_conv = nn.Sequential(nn.Conv3d(8, 1, 3, padding=1, bias=False))
        self.feat_conv = nn.Sequential(nn.Conv3d(8, 8, 3, padding=1, bias=False))

    def raw(self) -> CostRegRaw:
        r = CostRegRaw()
        r.in_channels, r.full = self.in_channels, int(self.full)

        def fill(slot, w, bn):
            for t in (w, bn.weight, bn.bias, bn.running_mean, bn.running_var):
                _require_f32c(t)
            r.conv[slot] = ConvBn(w.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                  bn.running_mean.data_ptr(), bn.running_var.data_ptr())
        for i in range(7 if self.full else 5):
            m = getattr(self, f"conv{i}")
            fill(i, m.conv.weight, m.bn)
        for i in ((7, 9, 11) if self.full else (9, 11)):
            m = getattr(self, f"conv{i}")
            fill(i, m[0].weight, m[1])
        _require_f32c(self.feat_conv[0].weight), _require_f32c(self.depth_conv[0].weight)
        r.feat_conv_w = self.feat_conv[0].weight.data_ptr()
        r.depth_conv_w = self.depth_conv[0].weight.data_ptr()
        return r


def _kaiming(m):
    """nerf.py:130-134."""
    if isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight.data)
        if m.bias is not None:
            nn.init.zeros_(m.bias.data)


def _fc(cin, cout, act=None):
    layers = [nn.Linear(cin, cout)]
    if act is not None:
        layers.append(act)
    return nn.Sequential(*layers)


class AggParams(nn.Module):
    """nerf.py:45-73."""

    def __init__(self, feat_ch: int, viewdir_agg: bool):
        super().__init__()
        self.feat_ch = feat_ch
        if viewdir_agg:
            self.view_fc = _fc(4, feat_ch, nn.ReLU())
        self.global_fc = _fc(feat_ch * 3, 32, nn.ReLU())
        self.agg_w_fc = _fc(32, 1, nn.ReLU())
        self.fc = _fc(32, 16, nn.ReLU())
        self.apply(_kaiming)


class NerfParams(nn.Module):
    """nerf.py:6-28 (``lrs`` is an empty ModuleList there: no parameters)."""

    def __init__(self, feat_ch: int, viewdir_agg: bool, hid_n: int = 64):
        super().__init__()
        self.feat_ch, self.viewdir_agg = feat_ch, viewdir_agg
        self.agg = AggParams(feat_ch, viewdir_agg)
        self.lr0 = _fc(8 + 16, hid_n, nn.ReLU())
        self.lrs = nn.ModuleList()
        self.sigma = _fc(hid_n, 1, nn.Softplus())
        self.color = nn.Sequential(nn.Linear(64 + 24 + feat_ch + 4, hid_n), nn.ReLU(), nn.Linear(hid_n, 1), nn.ReLU())
        for m in (self.lr0, self.sigma, self.color):
            m.apply(_kaiming)

    def raw(self) -> NerfRaw:
        r = NerfRaw()
        pairs = [("glob", self.agg.global_fc[0]), ("aggw", self.agg.agg_w_fc[0]), ("fc", self.agg.fc[0]),
                 ("lr0", self.lr0[0]), ("sigma", self.sigma[0]), ("col0", self.color[0]), ("col2", self.color[2])]
        if self.viewdir_agg:
            pairs.append(("view", self.agg.view_fc[0]))
        for name, lin in pairs:
            _require_f32c(lin.weight), _require_f32c(lin.bias)
            setattr(r, name + "_w", lin.weight.data_ptr())
            setattr(r, name + "_b", lin.bias.data_ptr())
        return r


def _require_f32c(t: torch.Tensor):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("parameters must be contiguous float32")


# --------------------------------------------------------------------------------------------------
# the network
# --------------------------------------------------------------------------------------------------
class Network(nn.Module):
    """MI355X-native ENeRF renderer with the reference's call surface (network.py:11-113).

    ``human=True`` reproduces ``network_human.py`` (boolean ``mask_at_box`` ray compaction at the
    last level, :90-107).  ``cfg`` is an :class:`EnerfConfig` (``EnerfConfig.from_yacs(lib.config.cfg)``
    inside the reference tree).
    """

    def __init__(self, cfg: Optional[EnerfConfig] = None, human: bool = False, lib: Optional[EnerfLib] = None,
                 check_nan: bool = False, feature_backend: str = "hip", overlap: bool = False):
        super().__init__()
        if feature_backend not in ("hip", "torch"):
            raise ValueError("feature_backend must be 'hip' or 'torch'")
        # overlap=True (HIP FeatureNet on a GPU only): the level-0 cost volume needs feature level_0 only, so the
        # rest of the FPN (level_1, level_2/texels: 0.19 of the 1.09 ms frame) is enqueued on a second HIP stream
        # next to the level-0 cost regularisation; events order the consumers.  Same kernels, bit-identical
        # results.  Measured on MI355X: the kernels do overlap in time but each slows down by what the other
        # takes (the fused smooth0 blocks hold 141 of the 160 KB of LDS per CU, so the conv3d blocks queue for
        # LDS): 927 -> 931 FPS.  Off by default — one stream keeps the frame graph-capturable.
        self.overlap = overlap
        self.fuse_build_rays = True         # forward(): build_rays in the render kernel's prologue (same bits)
        self._side_stream = None
        self._feat_events = {}
        # "torch": FeatureNet in PyTorch-ROCm/MIOpen (north_star's split); "hip": enerf_feature_net on the
        # matrix cores, channels-last outputs (SURVEY.md §8f row 2 — MIOpen was 49 % of the frame).
        self.feature_backend = feature_backend
        self._feat_ws_by_stream = {}        # FeatureNet scratch, one per HIP stream (frames may be in flight on several)
        self.cfg = cfg or EnerfConfig()
        self.cfg.cas.validate()
        self.human = human
        self.check_nan = check_nan          # reference traps NaN with ipdb + host sync (network.py:110)
        self._lib = lib
        self.feature_net = FeatureNet()
        cas = self.cfg.cas
        for i in range(cas.num):
            setattr(self, f"cost_reg_{i}", CostRegParams(int(32 * (2 ** (-i))), full=(i != 0)))   # network.py:15-20
            setattr(self, f"nerf_{i}", NerfParams(cas.nerf_model_feat_ch[i] + 3, self.cfg.viewdir_agg))
        self._packed: Dict[str, tuple] = {}           # name -> (packed image, ready event or None, stream id)
        self.options: Optional[Options] = None        # enerf_options_t for every launch of forward(); None = defaults
        self._tex_cache = None
        self._timer = None                  # optional stage timer (bench.py): .begin()/.mark(name)/.end()

    def _mark(self, name):
        if self._timer is not None:
            self._timer.mark(name)

    # -- weight images ---------------------------------------------------------------------------
    @property
    def lib(self) -> EnerfLib:
        if self._lib is None:
            self._lib = get_lib()
        return self._lib

    def invalidate_packed(self):
        self._packed = {}

    def _apply(self, fn, *a, **k):
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate_packed()
        return super().load_state_dict(*a, **k)

    # train()/eval() do NOT invalidate: the packed images fold the BN *running* statistics whatever the mode, and a
    # captured HIP graph (enerf_amd/graph.py) or a frame in flight may hold their addresses.

    def _packed_weights(self, name: str) -> torch.Tensor:
        """MFMA operand image of a sub-module, packed lazily by a device kernel on the stream of the first caller.
        Another stream (FramePipeline, the FeatureNet side stream) waits on the pack's event before reading it."""
        ent = self._packed.get(name)
        if ent is None:
            m = getattr(self, name)
            dev = next(m.parameters()).device
            if isinstance(m, CostRegParams):
                t = self.lib.cost_reg_pack(m.raw(), dev)
            elif isinstance(m, FeatureNet):
                t = self.lib.feature_net_pack(m.raw(), dev)
            else:
                t = self.lib.nerf_pack(m.raw(), m.feat_ch, m.viewdir_agg, dev)
            if t.is_cuda:
                st = torch.cuda.current_stream(dev)
                ent = (t, st.record_event(), st.cuda_stream)
            else:
                ent = (t, None, 0)
            self._packed[name] = ent
        t, ev, sid = ent
        if ev is not None:
            cur = torch.cuda.current_stream(t.device)
            if cur.cuda_stream != sid and not ev.query():
                cur.wait_event(ev)
        return t

    def prepare(self):
        """Pack every weight image now, on the current stream (e.g. before opening a FramePipeline or capturing a graph)."""
        names = ["feature_net"] if self.feature_backend == "hip" else []
        for i in range(self.cfg.cas.num):
            names += [f"cost_reg_{i}", f"nerf_{i}"]
        for n in names:
            self._packed_weights(n)
        return self

    # -- reference surface -----------------------------------------------------------------------
    def forward_feat(self, x):
        """network.py:58-67."""
        B, S, C, H, W = x.shape
        f2, f1, f0 = self.feature_net(x.view(B * S, C, H, W))
        return {"level_2": f0.reshape(B, S, f0.shape[1], H, W),
                "level_1": f1.reshape(B, S, f1.shape[1], H // 2, W // 2),
                "level_0": f2.reshape(B, S, f2.shape[1], H // 4, W // 4)}

    @property
    def _feat_ws(self):
        return self._feat_ws_by_stream.get(torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0)

    @_feat_ws.setter
    def _feat_ws(self, ws):
        self._feat_ws_by_stream[torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0] = ws

    def _forward_feat_hip(self, x, texel_level2: bool, options=None):
        """HIP FeatureNet: channels-last (B,S,h,w,C) maps tagged ``_enerf_cl``; level_2 optionally comes
        out as ready render texels (tagged ``_enerf_tex``) when it is only used for the full-res render."""
        B, S, C, H, W = x.shape
        lib, packed = self.lib, self._packed_weights("feature_net")
        src = x.reshape(B * S, C, H, W).contiguous()
        stride = 12 if texel_level2 else 8
        self._feat_events = {}
        if self.overlap and x.is_cuda:
            bufs = lib.feature_net_alloc(src, stride, self._feat_ws)
            f0, f1, f2, self._feat_ws = bufs
            lib.feature_net_stage(packed, src, bufs, lib.FEAT_TRUNK, stride, options)          # -> level_0, caller's stream
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=x.device)
            main, side = torch.cuda.current_stream(x.device), self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                lib.feature_net_stage(packed, src, bufs, lib.FEAT_LEVEL1, stride, options)     # -> level_1
                self._feat_events[1] = side.record_event()
                lib.feature_net_stage(packed, src, bufs, lib.FEAT_LEVEL2, stride, options)     # -> level_2 / texels
                self._feat_events[2] = side.record_event()
            # the buffers were allocated on `main`; every consumer (and so every later reuse by the caching
            # allocator) is ordered after these events through _wait_feat
        else:
            f0, f1, f2, self._feat_ws = lib.feature_net(packed, src, stride, self._feat_ws, options)
        feats = {"level_0": f0.view(B, S, H // 4, W // 4, 32), "level_1": f1.view(B, S, H // 2, W // 2, 16),
                 "level_2": f2.view(B, S, H, W, f2.shape[-1])}
        for k, v in feats.items():
            v._enerf_cl = True
        if texel_level2:
            feats["level_2"]._enerf_tex = True
        return feats

    def _wait_feat(self, feat_level: int):
        """Order the current stream after the side-stream stage that produces feature ``level_{feat_level}``."""
        for lv, ev in list(self._feat_events.items()):
            if lv <= feat_level:
                torch.cuda.current_stream().wait_event(ev)
                del self._feat_events[lv]

    def _texels(self, level, batch, im_feat):
        """unpreprocess + cat as the channels-last gather source (network.py:28-34); cached per frame."""
        self._wait_feat(2)
        if getattr(im_feat, "_enerf_tex", False):
            return im_feat
        key = (level, im_feat.data_ptr(), batch["src_inps"].data_ptr())
        if self._tex_cache is not None and self._tex_cache[0] == key:
            return self._tex_cache[1]
        cas = self.cfg.cas
        H, W = batch["src_inps"].shape[-2:]
        Hr, Wr = int(H * cas.render_scale[level]), int(W * cas.render_scale[level])
        if getattr(im_feat, "_enerf_cl", False):
            B, S, hf, wf, Cf = im_feat.shape
            tex = self.lib.pack_texels_cl(im_feat.reshape(B * S, hf, wf, Cf),
                                          batch["src_inps"].reshape(B * S, 3, H, W).contiguous(), Hr, Wr)
            tex = tex.view(B, S, Hr, Wr, tex.shape[-1])
            self._tex_cache = (key, tex)
            return tex
        B, S, Cf, Hf, Wf = im_feat.shape
        up = cas.render_scale[level] / cas.im_ibr_scale[level]
        if (int(Hf * up), int(Wf * up)) != (Hr, Wr):
            raise RuntimeError("im_feat resolution inconsistent with render_scale / im_ibr_scale")
        tex = self.lib.pack_img_feat_rgb(im_feat.reshape(B * S, Cf, Hf, Wf).contiguous(),
                                         batch["src_inps"].reshape(B * S, 3, H, W).contiguous(), Hr, Wr)
        tex = tex.view(B, S, Hr, Wr, tex.shape[-1])
        self._tex_cache = (key, tex)
        return tex

    def render_rays(self, rays, **kwargs):
        """network.py:24-43.  ``rays`` (B,N,12); kwargs: level, batch, im_feat, feature_volume, nerf_model."""
        level, batch, im_feat, vol = kwargs["level"], kwargs["batch"], kwargs["im_feat"], kwargs["feature_volume"]
        cas = self.cfg.cas
        if not getattr(vol, "_enerf_channels_last", False):                     # reference layout (B,8,D,h,w)
            if vol.dim() != 5 or vol.shape[1] != 8:
                raise RuntimeError("feature_volume must be (B,8,D,h,w) or a channels-last volume from forward()")
            B, _, D, h, w = vol.shape
            vol = self.lib.channels_last(vol.contiguous(), B, 8, D * h * w).view(B, D, h, w, 8)
        nerf = kwargs.get("nerf_model", None)
        name = f"nerf_{level}"
        if nerf is not None and nerf is not getattr(self, name):
            raise RuntimeError("render_rays: nerf_model must be this network's nerf_{level}")
        tex = self._texels(level, batch, im_feat)
        self._mark(f"texels_{level}")
        # internal fast path of forward(): 8-float rays + the level's (depth, std, near_far) maps — build_rays
        # (utils.py:390-420) then runs in the render kernel's prologue instead of as its own launch
        maps = kwargs.get("_build_rays_maps", None)
        rgb, depth, weights = self.lib.render_rays(
            rays.contiguous(), tex, vol, batch["src_exts"].contiguous(), batch["src_ixts"].contiguous(),
            batch["tar_ext"].contiguous(), self._packed_weights(name), n_samples=cas.num_samples[level],
            depth_inv=cas.depth_inv[level], F=cas.nerf_model_feat_ch[level] + 3,
            render_scale=cas.render_scale[level], white_bkgd=self.cfg.white_bkgd, maps=maps,
            options=kwargs.get("_options", self.options))
        self._mark(f"render_{level}")
        return {"rgb": rgb, "depth": depth, "weights": weights}

    def batchify_rays(self, rays, **kwargs):
        """network.py:45-55."""
        chunk = int(self.cfg.chunk_size)
        if rays.shape[1] <= chunk:
            return self.render_rays(rays, **kwargs)
        parts = [self.render_rays(rays[:, i:i + chunk], **kwargs) for i in range(0, rays.shape[1], chunk)]
        return {k: torch.cat([p[k] for p in parts], dim=1) for k in parts[0]}

    def forward(self, batch):
        """network.py:76-113 / network_human.py:69-119."""
        return self._forward(batch, self.options)

    def _forward(self, batch, options):
        if self.training:
            raise NotImplementedError("enerf_amd.Network: the HIP path is inference-only for now "
                                      "(training backward is SURVEY.md §8f row 1); call .eval()")
        cas, lib = self.cfg.cas, self.lib
        self._tex_cache = None
        src = batch["src_inps"]
        B, S, _, H, W = src.shape
        if self._timer is not None:
            self._timer.begin()
        with torch.no_grad():
            hip_feats = self.feature_backend == "hip"
            if hip_feats:
                # level_2 is only ever the im_feat of a full-resolution render: emit it as texels directly
                uses = [i for i in range(cas.num) if cas.render_if[i] and cas.render_im_feat_level[i] == 2]
                tex2 = bool(uses) and all(cas.render_scale[i] == 1.0 and cas.im_ibr_scale[i] == 1.0 for i in uses) \
                    and all(cas.nerf_model_feat_ch[i] == 8 for i in uses) and cas.num <= 2   # level_2 never feeds a cost volume
                feats = self._forward_feat_hip(src, tex2, options)
            else:
                feats = self.forward_feat(src)
            self._mark("feature_net")
            ret = {}
            prev = None
            for i in range(cas.num):
                D = cas.volume_planes[i]
                h, w = int(H * cas.volume_scale[i]), int(W * cas.volume_scale[i])
                f = feats[f"level_{i}"]
                if hip_feats:
                    self._wait_feat(i)
                    feat_cl = f                                   # already (B,S,Hs,Ws,C)
                    Hs, Ws, C = f.shape[2:]
                    if i == 2 and C != 8:
                        raise RuntimeError("level_2 texels cannot feed a cost volume")
                else:
                    C, Hs, Ws = f.shape[2:]
                    feat_cl = lib.channels_last(f.reshape(B * S, C, Hs * Ws), B * S, C, Hs * Ws).view(B, S, Hs, Ws, C)
                if prev is not None and not cas.depth_inv[i - 1]:
                    raise RuntimeError("cascade levels after a depth-space level are undefined in the "
                                       "reference (utils.py:130)")
                # get_proj_mats + get_depth_values of the level, one launch
                proj, dv, near_far = lib.level_prep(batch["src_ixts"].contiguous(), batch["src_exts"].contiguous(),
                                                    batch["tar_ixt"].contiguous(), batch["tar_ext"].contiguous(),
                                                    cas.im_feat_scale[i], cas.volume_scale[i],
                                                    batch["near_far"].contiguous(), prev, D, h, w, cas.depth_inv[i])
                self._mark(f"prep_{i}")
                vol = lib.build_feature_volume(feat_cl, proj, dv, C)
                self._mark(f"volume_{i}")
                name = f"cost_reg_{i}"
                m = getattr(self, name)
                feat3d, prob = lib.cost_reg(self._packed_weights(name), m.in_channels, m.full, vol, options=options)
                feat3d._enerf_channels_last = True      # (B,D,h,w,8); render_rays also accepts (B,8,D,h,w)
                self._mark(f"cost_reg_{i}")
                depth, std = lib.depth_regression(prob, dv, cas.depth_inv[i])
                self._mark(f"depth_reg_{i}")
                prev = (depth, std, near_far)
                if not cas.render_if[i]:
                    continue
                Hr, Wr = int(H * cas.render_scale[i]), int(W * cas.render_scale[i])
                masked = self.human and "mask_at_box" in batch and i == cas.num - 1
                rays8 = batch[f"rays_{i}"].contiguous()
                extra = {}
                if self.fuse_build_rays and not masked and rays8.shape[1] <= int(self.cfg.chunk_size):
                    rays = rays8                                     # build_rays runs inside the render launch
                    extra["_build_rays_maps"] = (depth, std, near_far)
                else:
                    rays = lib.build_rays(rays8, depth, std, near_far, Hr, Wr, cas.depth_inv[i])
                    self._mark(f"build_rays_{i}")
                if masked:
                    mask = batch["mask_at_box"].bool().reshape(1, -1)
                    rays = rays[mask][None]
                ret_i = self.batchify_rays(rays=rays, feature_volume=feat3d, batch=batch,
                                           im_feat=feats[f"level_{cas.render_im_feat_level[i]}"],
                                           nerf_model=getattr(self, f"nerf_{i}"), level=i, _options=options, **extra)
                if masked:
                    rgb = torch.zeros((1, mask.shape[1], 3), dtype=torch.float32, device=src.device)
                    if int(mask.sum()) > 1:
                        rgb[mask] = ret_i["rgb"][0]
                    ret_i["rgb"] = rgb
                ret_i["depth_mvs"] = 1.0 / depth if cas.depth_inv[i] else depth
                ret_i["std"] = std
                if self.check_nan and bool(ret_i["rgb"].isnan().any()):
                    raise RuntimeError(f"NaN in rgb_level{i}")
                ret.update({f"{k}_level{i}": v for k, v in ret_i.items()})
            self._wait_feat(2)              # never leave side-stream work un-joined (e.g. no level rendered level_2)
        if self._timer is not None:
            self._timer.end()
        return ret


class NetworkHuman(Network):
    """``lib/networks/enerf/network_human.py`` drop-in."""

    def __init__(self, cfg: Optional[EnerfConfig] = None, **kw):
        super().__init__(cfg, human=True, **kw)
