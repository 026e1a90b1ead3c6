"""Drop-in ``Network`` for the reference's ``lib/networks/enerf/network.py`` (and ``network_human.py``).

Same constructor role, same ``forward(batch) -> dict`` / ``render_rays(rays, **kwargs)`` /
``batchify_rays`` / ``forward_feat`` surface and the same ``state_dict`` names and shapes
(SURVEY.md §8b), so ``run.py`` / ``gui_human.py`` / ``net_utils.load_network(strict=True)`` work
unchanged (INTEGRATION.md shows the 6-line module a maintainer drops into ``lib/networks/``).

What runs where:
  * ``feature_net`` (2-D FPN, feature_net.py:4-36)  — ``feature_backend="hip"`` (default): the library's MFMA convolution
    kernels (csrc/conv2d.hip) inside the same ``enerf_forward`` call; ``feature_backend="torch"``: PyTorch-ROCm/MIOpen,
    the split BASELINE.json's north_star describes, NCHW maps handed over through the C ABI (kept for A/B);
  * everything else (network.py:80-112)            — hand-written HIP kernels through the C ABI
    (``enerf_amd/lib.py`` -> ``libenerf_hip.so``).  The ``cost_reg_*`` / ``nerf_*`` sub-modules here only
    OWN the parameters (so checkpoints, ``.cuda()``, ``SyncBatchNorm.convert_sync_batchnorm`` keep
    working); their weights are re-laid-out once per load into MFMA operand images by device kernels.
There is no eager fallback for inference: without the built library ``forward`` raises in ``.eval()`` mode.
In ``.train()`` mode ``forward`` is the differentiable path of ``enerf_amd/train_path.py``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import EnerfConfig
from .lib import (STAGE_COUNT, STAGE_NAMES, ConvBn, CostRegRaw, EnerfLib, FeatNetRaw, FrameArgs, NerfRaw, Options,
                  cascade_struct, get_lib)


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
                 check_nan: bool = False, feature_backend: str = "hip", static_shapes: bool = False):
        super().__init__()
        if feature_backend not in ("hip", "torch"):
            raise ValueError("feature_backend must be 'hip' or 'torch'")
        # "torch": FeatureNet in PyTorch-ROCm/MIOpen (north_star's split); "hip": enerf_feature_net on the
        # matrix cores, channels-last outputs (SURVEY.md §8f row 2 — MIOpen was 49 % of the frame).
        self.feature_backend = feature_backend
        # human variant only: the reference returns depth/weights with mask_at_box.sum() rows (network_human.py:93), a
        # data-dependent shape the host can only build after reading the count back (a wait for three tiny kernels).
        # static_shapes=True returns the full-size buffers (selected rays first) plus the device-side count
        # ``num_rays_level{i}`` instead: no host sync at all, so the frame can be graph-captured / pipelined.
        self.static_shapes = static_shapes
        self._frames = {}                   # per HIP stream: frame args struct, workspace, plan signature
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
        self._packed_gen = 0
        self.options: Optional[Options] = None        # enerf_options_t for every launch of forward(); None = defaults
        self._tex_cache = None
        self._timer = None                  # optional stage timer (bench.py StageTimer): .new_events(n) / .frame(list)

    # -- weight images ---------------------------------------------------------------------------
    @property
    def lib(self) -> EnerfLib:
        if self._lib is None:
            self._lib = get_lib()
        return self._lib

    def invalidate_packed(self):
        self._packed = {}
        self._packed_gen = getattr(self, "_packed_gen", 0) + 1      # per-stream pointer caches in _frames notice this

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
        names = ["feature_net"] if self.feature_backend == "hip" else []      # torch backend: MIOpen owns those weights
        for i in range(self.cfg.cas.num):
            names += [f"cost_reg_{i}", f"nerf_{i}"]
        for n in names:
            self._packed_weights(n)
        return self

    # -- reference surface -----------------------------------------------------------------------
    def forward_feat(self, x):
        """network.py:58-67 (FeatureNet in PyTorch-ROCm: ``feature_backend="torch"``, north_star's split)."""
        B, S, C, H, W = x.shape
        f2, f1, f0 = self.feature_net(x.view(B * S, C, H, W))
        return {"level_2": f0.reshape(B, S, f0.shape[1], H, W),
                "level_1": f1.reshape(B, S, f1.shape[1], H // 2, W // 2),
                "level_0": f2.reshape(B, S, f2.shape[1], H // 4, W // 4)}

    @staticmethod
    def _stream_key(t: torch.Tensor) -> int:
        return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0

    def _texels(self, level, batch, im_feat):
        """unpreprocess + cat as the channels-last gather source (network.py:28-34); cached per frame."""
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
        # internal fast path of forward(): 8-float rays + the level's (depth, std, near_far) maps — build_rays
        # (utils.py:390-420) then runs in the render kernel's prologue instead of as its own launch
        maps = kwargs.get("_build_rays_maps", None)
        rgb, depth, weights = self.lib.render_rays(
            rays.contiguous(), tex, vol, batch["src_exts"].contiguous(), batch["src_ixts"].contiguous(),
            batch["tar_ext"].contiguous(), self._packed_weights(name), n_samples=cas.num_samples[level],
            depth_inv=cas.depth_inv[level], F=cas.nerf_model_feat_ch[level] + 3,
            render_scale=cas.render_scale[level], white_bkgd=self.cfg.white_bkgd, maps=maps,
            options=kwargs.get("_options", self.options))
        return {"rgb": rgb, "depth": depth, "weights": weights}

    def batchify_rays(self, rays, **kwargs):
        """network.py:45-55."""
        chunk = int(self.cfg.chunk_size)
        if rays.shape[1] <= chunk:
            return self.render_rays(rays, **kwargs)
        parts = [self.render_rays(rays[:, i:i + chunk], **kwargs) for i in range(0, rays.shape[1], chunk)]
        return {k: torch.cat([p[k] for p in parts], dim=1) for k in parts[0]}

    def forward(self, batch):
        """network.py:76-113 / network_human.py:69-119 — one ``enerf_forward`` C call (enerf_amd/csrc/frame.hip)."""
        return self._forward(batch, self.options)

    def _alloc_outputs(self, B, H, W, n_rays, dev):
        """Fresh output tensors of one frame (network.py:93-108 keys) + their addresses per rendered level."""
        cas = self.cfg.cas
        ret, ptrs = {}, {}
        for i in range(cas.num):
            if not cas.render_if[i]:
                continue
            N, h, w = n_rays[i], int(H * cas.volume_scale[i]), int(W * cas.volume_scale[i])
            rgb = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
            depth = torch.empty((B, N), dtype=torch.float32, device=dev)
            weights = torch.empty((B, N, cas.num_samples[i]), dtype=torch.float32, device=dev)
            dmvs = torch.empty((B, h, w), dtype=torch.float32, device=dev)
            std = torch.empty((B, h, w), dtype=torch.float32, device=dev)
            ret.update({f"rgb_level{i}": rgb, f"depth_level{i}": depth, f"weights_level{i}": weights,
                        f"depth_mvs_level{i}": dmvs, f"std_level{i}": std})
            ptrs[i] = (rgb.data_ptr(), depth.data_ptr(), weights.data_ptr(), dmvs.data_ptr(), std.data_ptr())
        return ret, ptrs

    def _frame_state(self, key):
        st = self._frames.get(key)
        if st is None:
            st = {"args": FrameArgs(), "ws": None, "need": None, "sig": None, "pin": None}
            st["args"].cas = cascade_struct(self.cfg)
            self._frames[key] = st
        return st

    def _forward(self, batch, options):
        if self.training:
            # trainer.py:56-63 / losses/enerf.py:16-56: the differentiable path on this network's own parameter modules
            # (BatchNorm batch statistics, autograd, DDP-ready) — enerf_amd/train_path.py
            from .train_path import forward_train
            self.invalidate_packed()                   # the optimizer is about to change what the packed images hold
            return forward_train(self, batch)
        cas, lib = self.cfg.cas, self.lib
        if self._tex_cache is not None:
            self._tex_cache = None                     # (nn.Module.__setattr__ costs 6 us: only when there is something to drop)
        src = batch["src_inps"]
        B, S, _, H, W = src.shape
        dev = src.device
        # Host time before the C call is GPU idle time under the reference's sync-per-frame protocol (run.py:62-76: measured
        # 30 us of a 0.80 ms dtu frame), so the per-frame Python work is kept minimal: one stream lookup; everything that depends
        # only on the frame's SHAPES, the weights' generation and the options (packed-weight pointers, ray counts, workspace
        # plan: ~35 ctypes field stores) is written when that key changes, not per frame; per frame only the batch's and the
        # outputs' addresses are stored, and the OUTPUT tensors of this frame were allocated right after the previous frame's
        # launch (below) — every frame still returns fresh, never-aliased tensors.
        sid = torch.cuda.current_stream(dev).cuda_stream if src.is_cuda else 0
        st = self._frame_state(sid)
        a = st["args"]
        keep = []                                             # tensors whose addresses the call uses

        def ptr(t):
            if not t.is_contiguous():
                t = t.contiguous()
            if t.dtype != torch.float32:
                raise RuntimeError(f"expected float32, got {t.dtype}")
            keep.append(t)
            return t.data_ptr()

        with torch.no_grad():
            a.src_inps, a.src_exts, a.src_ixts = ptr(src), ptr(batch["src_exts"]), ptr(batch["src_ixts"])
            a.tar_ext, a.tar_ixt, a.near_far = ptr(batch["tar_ext"]), ptr(batch["tar_ixt"]), ptr(batch["near_far"])
            masked = self.human and "mask_at_box" in batch
            hip_feats = self.feature_backend == "hip"
            rays_of = [batch.get(f"rays_{i}") if cas.render_if[i] else None for i in range(cas.num)]
            sig = (B, S, H, W, hip_feats, masked) + tuple(
                -1 if not cas.render_if[i] else (-2 if rays_of[i] is None else rays_of[i].shape[1]) for i in range(cas.num))
            # (ctypes structures compare by identity: key on the option VALUES, so that a fresh-but-equal Options does not
            # refresh every frame and an Options mutated in place does)
            okey = None if options is None else tuple(getattr(options, f[0]) for f in options._fields_)
            static_key = (sig, self._packed_gen, okey, dev)
            if st.get("static_key") != static_key:            # shapes / weights / options changed: the shape-only fields
                a.B, a.S, a.H, a.W = B, S, H, W
                pk = st.get("packed")
                if pk is None or pk[0] != self._packed_gen:   # first frame on this stream / weights changed: (re)pack, wait
                    names = (["feature_net"] if hip_feats else []) + \
                            [f"cost_reg_{i}" for i in range(cas.num)] + [f"nerf_{i}" for i in range(cas.num) if cas.render_if[i]]
                    pk = (self._packed_gen, {n: self._packed_weights(n).data_ptr() for n in names})
                    st["packed"] = pk
                pp = pk[1]
                a.feature_net_packed = pp["feature_net"] if hip_feats else None
                if hip_feats:
                    for l in range(3):
                        a.feats_nchw[l] = None
                n_rays = []
                for i in range(cas.num):
                    a.cost_reg_packed[i] = pp[f"cost_reg_{i}"]
                    if not cas.render_if[i]:
                        a.rays[i], a.n_rays[i], a.nerf_packed[i] = None, 0, None
                        n_rays.append(0)
                        continue
                    a.nerf_packed[i] = pp[f"nerf_{i}"]
                    # no rays in the batch: the full image, generated on the device (enerf_utils.py:61-71)
                    N = rays_of[i].shape[1] if rays_of[i] is not None else int(H * cas.render_scale[i]) * int(W * cas.render_scale[i])
                    a.n_rays[i] = N
                    n_rays.append(N)
                    if rays_of[i] is None:
                        a.rays[i] = None
                a.options = None if options is None else C.pointer(options)
                if not (masked and cas.render_if[cas.num - 1]):
                    a.mask_at_box, a.ray_index, a.ray_count, a.ray_index_ready = None, None, None, 0
                a.stage_events = None
                st["static_key"], st["n_rays"], st["plan"] = static_key, n_rays, True
            n_rays = st["n_rays"]
            if not hip_feats:
                feats = self.forward_feat(src)
                for l in range(3):
                    a.feats_nchw[l] = ptr(feats[f"level_{l}"])
            for i in range(cas.num):
                if rays_of[i] is not None:
                    a.rays[i] = ptr(rays_of[i])
            nxt = st.get("next_out")
            if nxt is not None and nxt[0] == (sig, dev):
                ret, optrs = nxt[1], nxt[2]
            else:
                ret, optrs = self._alloc_outputs(B, H, W, n_rays, dev)
            st["next_out"] = None
            for i, p5 in optrs.items():
                a.rgb[i], a.depth[i], a.weights[i], a.depth_mvs[i], a.std[i] = p5
            count_ready = None
            last = cas.num - 1
            if masked and cas.render_if[last]:
                mask = batch["mask_at_box"].contiguous()
                keep.append(mask)
                a.mask_at_box, a.mask_elem_bytes = mask.data_ptr(), mask.element_size()
                # compaction first, on its own: its count can then reach the host (for the reference's data-dependent
                # output shapes) after waiting for these three tiny kernels only, not for the frame
                index, count = lib.mask_compact(mask.reshape(-1))
                keep += [index, count]
                a.ray_index, a.ray_count, a.ray_index_ready = index.data_ptr(), count.data_ptr(), 1
                if not self.static_shapes:
                    if count.is_cuda:
                        if st["pin"] is None:
                            st["pin"] = torch.empty((1,), dtype=torch.int32, pin_memory=True)
                        st["pin"].copy_(count, non_blocking=True)
                        count_ready = torch.cuda.current_stream(dev).record_event()
                    else:
                        st["pin"] = count
            timer = self._timer
            if timer is not None and src.is_cuda:
                evs = timer.new_events(STAGE_COUNT)
                arr = (C.c_void_p * STAGE_COUNT)(*[e.cuda_event for e in evs])
                a.stage_events = C.cast(arr, C.POINTER(C.c_void_p))
                st["static_key"] = None                          # (the next untimed frame clears the event slots again)
            if st.get("plan"):                                   # after a static refresh: (re)plan the workspace for these shapes
                if st["sig"] != (sig, okey):                          # some options change the plan's workspace needs
                    a.workspace, a.workspace_bytes = None, 0
                    st["need"], st["sig"] = lib.forward_workspace_bytes(a), (sig, okey)
                if st["ws"] is None or st["ws"].numel() * 4 < st["need"] or st["ws"].device != dev:
                    st["ws"] = torch.empty(((st["need"] + 3) // 4,), dtype=torch.float32, device=dev)
                a.workspace, a.workspace_bytes = st["ws"].data_ptr(), st["ws"].numel() * 4
                st["plan"] = False
            lib.forward(a, sid if src.is_cuda else None)
            # the NEXT frame's outputs, allocated while the GPU works on this one
            if not (src.is_cuda and torch.cuda.is_current_stream_capturing()):   # (a capture keeps its allocations private)
                st["next_out"] = ((sig, dev),) + self._alloc_outputs(B, H, W, n_rays, dev)
            if timer is not None and src.is_cuda:
                used = [0, 1]
                for i in range(cas.num):
                    used += [2 + 6 * i + k for k in range(6 if cas.render_if[i] else 4)]
                timer.frame([(STAGE_NAMES[k], evs[k]) for k in used])
            if masked and cas.render_if[last]:
                if self.static_shapes:                           # no host sync: full-size buffers + the count on the device
                    ret[f"num_rays_level{last}"] = count
                else:                                            # network_human.py:93: depth / weights have mask.sum() rows
                    if count_ready is not None:
                        count_ready.synchronize()
                    m = int(st["pin"][0])
                    ret[f"depth_level{last}"] = ret[f"depth_level{last}"][:, :m]
                    ret[f"weights_level{last}"] = ret[f"weights_level{last}"][:, :m]
            if self.check_nan:
                for i in range(cas.num):
                    if cas.render_if[i] and bool(ret[f"rgb_level{i}"].isnan().any()):
                        raise RuntimeError(f"NaN in rgb_level{i}")
        return ret


class NetworkHuman(Network):
    """``lib/networks/enerf/network_human.py`` drop-in."""

    def __init__(self, cfg: Optional[EnerfConfig] = None, **kw):
        super().__init__(cfg, human=True, **kw)
