"""ctypes binding of the C ABI in include/enerf_hip.h (libenerf_hip.so).

PyTorch is used only as the owner of device memory and streams: every call passes raw
``tensor.data_ptr()`` addresses, sizes and the current HIP stream handle.  There is NO CPU or eager
fallback: if the shared library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from typing import Optional

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libenerf_hip.so")
ABI_VERSION = 11

_f = C.c_void_p     # device float*
_i = C.c_int
_ll = C.c_longlong
_fl = C.c_float


class ConvBn(C.Structure):
    _fields_ = [("w", _f), ("bn_weight", _f), ("bn_bias", _f), ("bn_mean", _f), ("bn_var", _f)]


class CostRegRaw(C.Structure):
    _fields_ = [("conv", ConvBn * 12), ("feat_conv_w", _f), ("depth_conv_w", _f), ("in_channels", _i), ("full", _i)]


class FeatNetRaw(C.Structure):
    _fields_ = [("conv", ConvBn * 6)] + [(n, _f) for n in ("toplayer_w", "toplayer_b", "lat1_w", "lat1_b", "lat0_w",
                                                            "lat0_b", "smooth1_w", "smooth1_b", "smooth0_w", "smooth0_b")]


class NerfRaw(C.Structure):
    _fields_ = [(n, _f) for n in ("view_w", "view_b", "glob_w", "glob_b", "aggw_w", "aggw_b", "fc_w", "fc_b",
                                  "lr0_w", "lr0_b", "sigma_w", "sigma_b", "col0_w", "col0_b", "col2_w", "col2_b")]


class Options(C.Structure):
    """``enerf_options_t``: explicit kernel-variant choices, passed per call (all zero = defaults)."""
    _fields_ = [("conv3d_global_only", _i), ("conv3d_lds_min_voxels", _ll), ("conv3d_pk8", _i),
                ("featnet_unfused", _i), ("featnet_smooth0_plain", _i), ("conv3d_b4", _i), ("single_stream", _i),
                ("side_gate", _i), ("fuse_depth_prep", _i), ("conv3d_t2_variant", _i), ("conv3d_small_variant", _i),
                ("render_precision", _i)]

    def __repr__(self):
        return "Options(" + ", ".join(f"{n}={getattr(self, n)}" for n, _ in self._fields_ if getattr(self, n)) + ")"


# what "most frames per second" prefers over "one frame as fast as possible" (enerf_amd/pipeline.py)
def throughput_options() -> "Options":
    # frames in flight already fill the chip: the in-frame side lanes add nothing there (measured equal) and triple the
    # number of streams the runtime has to map onto its few hardware queues
    return Options(single_stream=1)


def _opt(o):
    return None if o is None else C.byref(o)


class MlpBwdArgs(C.Structure):
    """``enerf_mlp_bwd_args_t``."""
    _fields_ = ([(n, _f) for n in ("vox", "x", "g_raw", "packed", "bimg", "g_vox", "g_x")] + [("save", C.c_void_p * 16),
                ("P", _ll), ("F", _i), ("S", _i), ("image_offsets", _i * 8)])


class GatherArgs(C.Structure):
    """``enerf_gather_args_t``."""
    _fields_ = ([(n, _f) for n in ("xyz", "dn", "uv", "tex", "vol", "cam", "tcen", "x", "vox", "g_x", "g_vox", "g_tex",
                                   "g_vol", "g_xyz", "g_dn")]
                + [("P", _ll)] + [(n, _i) for n in ("B", "S", "F", "Hr", "Wr", "D", "h", "w", "ray_w", "n_samples")])


class GemmWgradDesc(C.Structure):
    _fields_ = [("a", _f), ("lda", _i), ("Ca", _i), ("b", _f), ("ldb", _i), ("Cb", _i), ("P", _ll), ("grad_w", _f), ("ldw", _i),
                ("grad_bias", _f), ("partials", _f), ("partial_chunks", _i)]


class RenderArgs(C.Structure):
    _fields_ = ([(n, _f) for n in ("rays12", "tex", "vol", "src_exts", "src_ixts", "tar_ext", "packed", "rgb",
                                   "depth", "weights")]
                + [(n, _i) for n in ("B", "N", "S", "n_samples", "depth_inv", "Hr", "Wr", "F", "D", "h", "w",
                                     "white_bkgd")]
                + [("render_scale", _fl)]
                + [(n, _f) for n in ("rays8", "depth_map", "std_map", "nf_map")] + [("map_h", _i), ("map_w", _i)]
                + [("options", C.POINTER(Options)), ("ray_index", C.c_void_p), ("ray_count", C.c_void_p),
                   ("scatter_rgb", _i), ("max_blocks", _i)])


MAX_LEVELS = 3
STAGE_COUNT = 2 + 6 * MAX_LEVELS
STAGE_NAMES = ["begin", "feature_net"] + [f"{n}_{i}" for i in range(MAX_LEVELS)
                                          for n in ("prep", "volume", "cost_reg", "depth_reg", "texels", "render")]
_dL, _iL, _fL = C.c_double * MAX_LEVELS, C.c_int * MAX_LEVELS, C.c_void_p * MAX_LEVELS


class Cascade(C.Structure):
    """``enerf_cascade_t`` (cfg.enerf.cas_config)."""
    _fields_ = [("num", _i), ("depth_inv", _iL), ("volume_scale", _dL), ("volume_planes", _iL), ("im_feat_scale", _dL),
                ("im_ibr_scale", _dL), ("render_scale", _dL), ("render_im_feat_level", _iL), ("nerf_model_feat_ch", _iL),
                ("render_if", _iL), ("num_samples", _iL), ("white_bkgd", _i)]


class FrameArgs(C.Structure):
    """``enerf_frame_args_t``."""
    _fields_ = ([(n, _f) for n in ("src_inps", "src_exts", "src_ixts", "tar_ext", "tar_ixt", "near_far")]
                + [("rays", _fL), ("n_rays", _iL), ("mask_at_box", C.c_void_p), ("mask_elem_bytes", _i)]
                + [(n, _i) for n in ("B", "S", "H", "W")] + [("cas", Cascade)]
                + [("feature_net_packed", _f), ("cost_reg_packed", _fL), ("nerf_packed", _fL), ("feats_nchw", C.c_void_p * 3)]
                + [(n, _fL) for n in ("rgb", "depth", "weights", "depth_mvs", "std")]
                + [("ray_index", C.c_void_p), ("ray_count", C.c_void_p), ("ray_index_ready", _i),
                   ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("options", C.POINTER(Options)),
                   ("stage_events", C.POINTER(C.c_void_p))])


def cascade_struct(cfg) -> Cascade:
    """EnerfConfig -> enerf_cascade_t."""
    cas = cfg.cas
    if cas.num > MAX_LEVELS:
        raise EnerfError(f"cas_config.num={cas.num} > {MAX_LEVELS}")
    c = Cascade(num=cas.num, white_bkgd=int(cfg.white_bkgd))
    for i in range(cas.num):
        c.depth_inv[i] = int(cas.depth_inv[i]); c.volume_scale[i] = float(cas.volume_scale[i])
        c.volume_planes[i] = int(cas.volume_planes[i]); c.im_feat_scale[i] = float(cas.im_feat_scale[i])
        c.im_ibr_scale[i] = float(cas.im_ibr_scale[i]); c.render_scale[i] = float(cas.render_scale[i])
        c.render_im_feat_level[i] = int(cas.render_im_feat_level[i])
        c.nerf_model_feat_ch[i] = int(cas.nerf_model_feat_ch[i]); c.render_if[i] = int(cas.render_if[i])
        c.num_samples[i] = int(cas.num_samples[i])
    return c


_SIGNATURES = {
    "enerf_abi_version": (_i, []),
    "enerf_last_error": (C.c_char_p, []),
    "enerf_channels_last": (_i, [_f, _f, _i, _i, _ll, _i, _f]),
    "enerf_channels_first": (_i, [_f, _f, _i, _i, _ll, _i, _f]),
    "enerf_pack_img_feat_rgb": (_i, [_f, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _f, _f]),
    "enerf_feature_net_packed_floats": (_ll, []),
    "enerf_feature_net_pack": (_i, [C.POINTER(FeatNetRaw), _f, _f]),
    "enerf_feature_net_workspace_bytes": (C.c_size_t, [_i, _i, _i]),
    "enerf_feature_net": (_i, [_f, _f, _i, _i, _i, _f, _f, _f, _i, _f, C.c_size_t, C.POINTER(Options), _f]),
    "enerf_feature_net_stage": (_i, [_f, _f, _i, _i, _i, _f, _f, _f, _i, _f, C.c_size_t, _i, C.POINTER(Options), _f]),
    "enerf_pack_texels_cl": (_i, [_f, _i, _f, _i, _i, _i, _i, _i, _i, _f, _f]),
    "enerf_get_proj_mats": (_i, [_f, _f, _f, _f, _i, _i, _fl, _fl, _f, _f]),
    "enerf_get_depth_values": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f]),
    "enerf_level_prep": (_i, [_f, _f, _f, _f, _i, _i, _fl, _fl, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f, _f, _f]),
    "enerf_build_feature_volume": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f]),
    "enerf_cost_reg_packed_floats": (_ll, [_i, _i]),
    "enerf_cost_reg_pack": (_i, [C.POINTER(CostRegRaw), _f, _f]),
    "enerf_cost_reg_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "enerf_cost_reg": (_i, [_f, _i, _i, _f, _i, _i, _i, _i, _f, _f, _f, C.c_size_t, C.POINTER(Options), _f]),
    "enerf_depth_regression": (_i, [_f, _f, _i, _i, _i, _i, _i, _f, _f, _f]),
    "enerf_build_rays": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f]),
    "enerf_nerf_packed_floats": (_ll, [_i]),
    "enerf_nerf_pack": (_i, [C.POINTER(NerfRaw), _i, _i, _f, _f]),
    "enerf_render_rays": (_i, [C.POINTER(RenderArgs), _f]),
    "enerf_mask_compact_workspace_bytes": (C.c_size_t, [_ll]),
    "enerf_mask_compact": (_i, [C.c_void_p, _i, _ll, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, _f]),
    "enerf_forward_workspace_bytes": (C.c_size_t, [C.POINTER(FrameArgs)]),
    "enerf_forward": (_i, [C.POINTER(FrameArgs), _f]),
    "enerf_build_feature_volume_bwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f]),
    "enerf_depth_regression_bwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f]),
    "enerf_conv_wgrad_workspace_bytes": (C.c_size_t, [_ll, _i, _i, _i, _i, _i]),
    "enerf_conv_wgrad": (_i, [_f, _f] + [_i] * 16 + [_f, C.c_void_p, C.c_size_t, _f]),
    "enerf_gemm_wgrad_workspace_bytes": (C.c_size_t, [_ll, _i, _i, _i]),
    "enerf_nerf_mlp_bwd": (_i, [C.POINTER(MlpBwdArgs), _f]),
    "enerf_nerf_mlp_bwd_chunks": (_ll, [_ll]),
    "enerf_nerf_mlp_bwd_partials": (_i, [C.POINTER(MlpBwdArgs), _i, _f, _f, _f, _f, _f]),
    "enerf_colsum": (_i, [_f, _i, _i, _f, _f]),
    "enerf_gemm_wgrad": (_i, [_f, _i, _i, _f, _i, _i, _ll, _f, _f, C.c_void_p, C.c_size_t, _f]),
    "enerf_bn_train_apply": (_i, [_f, _ll, _i, C.c_void_p, C.c_size_t, _f, _f, C.c_double, C.c_double, _f, _f, C.c_void_p, _i, C.c_void_p, _f, _f, _i, _f, _f]),
    "enerf_bn_train_bwd_apply": (_i, [_f, _f, _i, _ll, _i, C.c_void_p, C.c_size_t, C.c_void_p, _f, _f, _f, _f]),
    "enerf_wgrad_reduce_begin": (_i, []),
    "enerf_wgrad_reduce_flush": (_i, [_f]),
    "enerf_selftest_checks": (_i, []),
    "enerf_selftest_primitives": (_i, [_f, _i, _f, _i, C.c_void_p, _f]),
    "enerf_gemm_wgrad_group_workspace_bytes": (C.c_size_t, [C.POINTER(GemmWgradDesc), _i]),
    "enerf_gemm_wgrad_group": (_i, [C.POINTER(GemmWgradDesc), _i, C.c_void_p, C.c_size_t, _f]),
    "enerf_nerf_mlp_fwd": (_i, [_f, _f, _f, _ll, _i, _i, _f, _f]),
    "enerf_gather_fwd": (_i, [C.POINTER(GatherArgs), _f]),
    "enerf_gather_bwd": (_i, [C.POINTER(GatherArgs), _f]),
    "enerf_conv3d_layer_packed_floats": (_ll, [_i, _i, _i]),
    "enerf_conv3d_layer_pack": (_i, [_f, _i, _i, _i, _f, _f]),
    "enerf_conv3d_layer": (_i, [_f, _i, _i, _i, _f, _f, _f, _i, _i, _i, _i, C.POINTER(Options), _f]),
    "enerf_up2_adjoint": (_i, [_f, _f, _i, _i, _i, _i, _f, _f]),
    "enerf_conv2d_layer_packed_floats": (_ll, [_i, _i, _i]),
    "enerf_conv2d_layer_pack": (_i, [_f, _f, _i, _i, _i, _f, _f]),
    "enerf_conv2d_layer": (_i, [_f, _i, _i, _i, _i, _f, _f, _f, _i, _i, _i, _f]),
    "enerf_channel_sums": (_i, [_f, _f, _f, _f, _f, _ll, _i, C.c_void_p, _f]),
    "enerf_channel_sums_workspace_bytes": (C.c_size_t, [_ll, _i]),
    "enerf_channel_sums_ws": (_i, [_f, _f, _f, _f, _f, _ll, _i, C.c_void_p, C.c_void_p, C.c_size_t, _f]),
    "enerf_bn_train_coeffs": (_i, [C.c_void_p, C.c_void_p, C.c_double, _f, _f, C.c_double, C.c_double, _f, _f, C.c_void_p, _i, _i,
                                   C.c_void_p, _f, _f]),
    "enerf_bn_train_bwd_coeffs": (_i, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, _f, _i, _f, _f, _f]),
    "enerf_bn_train_stats": (_i, [_f, _ll, _i, C.c_void_p, C.c_size_t, _f, _f, C.c_double, C.c_double, _f, _f, C.c_void_p, _i, C.c_void_p,
                                  C.c_void_p, _f, C.c_void_p]),
    "enerf_bn_train_bwd_stats": (_i, [_f, _f, _f, _f, _f, _ll, _i, C.c_void_p, C.c_size_t, C.c_void_p, _f, _f, _f, C.c_void_p]),
    "enerf_channel_affine": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _ll, _i, _f, _f]),
    "enerf_conv2d_s2k5_dgrad_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "enerf_conv2d_s2k5_dgrad": (_i, [_f, _i, _i, _f, _f, _f, _i, _i, _i, C.c_void_p, C.c_size_t, _f]),
    "enerf_conv2d_s2k5_dgrad_packed_floats": (_ll, [_i, _i]),
    "enerf_conv2d_s2k5_dgrad_pack": (_i, [_f, _i, _i, _f, _f, _f]),
    "enerf_conv2d_s2k5_dgrad_packed": (_i, [_f, _i, _i, _f, _f, _f, _i, _i, _i, C.c_void_p, C.c_size_t, _f]),
    "enerf_resize_ac_adjoint": (_i, [_f, _f, _i, _i, _i, _i, _i, _f, _f]),
    "enerf_get_depth_values_bwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f]),
    "enerf_ray_samples_fwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f]),
    "enerf_ray_samples_bwd": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f]),
    "enerf_camera_tables": (_i, [_f, _f, _f, _i, _i, _fl, _f, _f, _f]),
    "enerf_weights_flip_transpose": (_i, [_f, _i, _i, _i, _f, _f]),
    "enerf_concat2_pad": (_i, [_f, _ll, _f, _ll, _ll, _f, _f]),
    "enerf_pack_texels_train": (_i, [_f, _i, _f, _i, _i, _i, _i, _i, _f, _f]),
    "enerf_slice_channels": (_i, [_f, _ll, _i, _i, _i, _f, _f]),
    "enerf_concat_channels": (_i, [_f, _i, _f, _i, _ll, _i, _f, _f]),
    "enerf_gather_images": (_i, [C.c_void_p, _i, C.c_void_p, C.c_void_p, _ll, _f, _f]),
    "enerf_add": (_i, [_f, _f, _ll, _f, _f]),
    "enerf_cast_f64_f32": (_i, [C.c_void_p, _ll, _f, _f]),
    "enerf_reciprocal": (_i, [_f, _ll, _f, _f]),
    "enerf_composite": (_i, [_f, _f, _ll, _i, _i, _f, _f, _f, _f]),
    "enerf_composite_bwd": (_i, [_f, _f, _f, _f, _f, _ll, _i, _f, _f, _f]),
    "enerf_gen_rays": (_i, [_f, _f, _i, _i, _i, _fl, _f, _f]),
    "enerf_pack_rgb8": (_i, [_f, _i, _i, _i, _f, _f]),
    "enerf_eval_stats": (_i, [_f, _f, C.c_void_p, _i, _ll, _i, _i, _i, _i, _f, _f, _ll, _f, _f]),
    "enerf_gen_rays_at": (_i, [_f, _f, C.c_void_p, _i, _i, _fl, _f, _f]),
    "enerf_rays_bbox_mask": (_i, [_f, _f, _ll, C.c_void_p, _f]),
    "enerf_select_views": (_i, [_f, _i, _f, _i, C.c_void_p, _f]),
    "enerf_gather_views": (_i, [_f, _f, _f, C.c_void_p, _i, _i, _i, _f, _f, _f, _f]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class EnerfError(RuntimeError):
    pass


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise EnerfError(f"expected contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t.data_ptr()


class EnerfLib:
    """Thin typed wrapper over the shared library.  ``path`` is only overridden by tests (CPU-emulated
    twin built from the same kernel sources); the product always loads :data:`LIB_PATH`."""

    def __init__(self, path: str = LIB_PATH):
        if not os.path.exists(path):
            raise EnerfError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950); there is no fallback path")
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(self.dll, name)          # raises AttributeError if a symbol is missing
            fn.restype, fn.argtypes = res, args
        v = self.dll.enerf_abi_version()
        if v != ABI_VERSION:
            raise EnerfError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")

    # -- plumbing --------------------------------------------------------------------------------
    @staticmethod
    def stream_of(t: torch.Tensor):
        return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None

    _tls = threading.local()           # .keep: a list while wgrad_reduce_batch() is open ON THIS THREAD (the C side records per thread too):
                                       # the partial sums must outlive their wrapper calls

    @property
    def _deferred_scratch(self):
        return getattr(self._tls, "keep", None)

    @_deferred_scratch.setter
    def _deferred_scratch(self, value):
        self._tls.keep = value

    def _scratch(self, nbytes: int, device):
        """Per-call scratch from torch's caching allocator (stream-ordered, graph-capture safe)."""
        t = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=device)
        keep = self._deferred_scratch
        if keep is not None:
            keep.append(t)
        return t

    @contextlib.contextmanager
    def wgrad_reduce_batch(self, ref):
        """ABI v11: inside the block the second stage of every enerf_conv_wgrad / enerf_gemm_wgrad call (the reduction of the blocks'
        partial sums into grad_w) is recorded instead of launched, and ONE kernel runs them all on leaving it (on ``ref``'s stream):
        the gradients returned inside the block are complete only after it.  Same arithmetic, same bits."""
        if os.environ.get("ENERF_WGRAD_DEFER", "1") == "0":       # A/B only (tools/gpu_r06_defer_ab.sh): immediate second stages
            yield
            return
        self._check(self.dll.enerf_wgrad_reduce_begin(), "wgrad_reduce_begin")
        self._deferred_scratch = []
        try:
            yield
        finally:
            try:
                self._check(self.dll.enerf_wgrad_reduce_flush(self.stream_of(ref)), "wgrad_reduce_flush")
            finally:
                self._deferred_scratch = None

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise EnerfError(f"{what} failed ({rc}): {self.dll.enerf_last_error().decode()}")

    # -- entry points -----------------------------------------------------------------------------
    def channels_last(self, src, n, C_, P, Cpad=None):
        Cpad = Cpad or C_
        dst = torch.empty((n, P, Cpad), dtype=torch.float32, device=src.device)
        self._check(self.dll.enerf_channels_last(_ptr(src), _ptr(dst), n, C_, P, Cpad, self.stream_of(src)),
                    "channels_last")
        return dst

    def channels_first(self, src, n, C_, P, Cpad=None):
        Cpad = Cpad or C_
        dst = torch.empty((n, C_, P), dtype=torch.float32, device=src.device)
        self._check(self.dll.enerf_channels_first(_ptr(src), _ptr(dst), n, C_, P, Cpad, self.stream_of(src)),
                    "channels_first")
        return dst

    def pack_img_feat_rgb(self, im_feat, src_inps, Hr, Wr):
        n_img, Cf, Hf, Wf = im_feat.shape
        H, W = src_inps.shape[-2:]
        tex = 4 * ((Cf + 3 + 3) // 4)
        out = torch.empty((n_img, Hr, Wr, tex), dtype=torch.float32, device=im_feat.device)
        self._check(self.dll.enerf_pack_img_feat_rgb(_ptr(im_feat), Cf, Hf, Wf, _ptr(src_inps), H, W, Hr, Wr, tex,
                                                     n_img, _ptr(out), self.stream_of(out)), "pack_img_feat_rgb")
        return out

    def feature_net_pack(self, raw: FeatNetRaw, device):
        n = self.dll.enerf_feature_net_packed_floats()
        packed = torch.empty((n,), dtype=torch.float32, device=device)
        self._check(self.dll.enerf_feature_net_pack(C.byref(raw), _ptr(packed), self.stream_of(packed)),
                    "feature_net_pack")
        return packed

    FEAT_ALL, FEAT_TRUNK, FEAT_LEVEL1, FEAT_LEVEL2 = 0, 1, 2, 3

    def feature_net_alloc(self, src_inps, l2_stride=8, workspace=None):
        """Output / workspace tensors of the FeatureNet for ``src_inps`` (n,3,H,W)."""
        n, _, H, W = src_inps.shape
        dev = src_inps.device
        need = self.dll.enerf_feature_net_workspace_bytes(n, H, W)
        if workspace is None or workspace.numel() * 4 < need:
            workspace = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=dev)
        f0 = torch.empty((n, H // 4, W // 4, 32), dtype=torch.float32, device=dev)
        f1 = torch.empty((n, H // 2, W // 2, 16), dtype=torch.float32, device=dev)
        f2 = torch.empty((n, H, W, l2_stride), dtype=torch.float32, device=dev)
        return f0, f1, f2, workspace

    def feature_net_stage(self, packed, src_inps, bufs, stage, l2_stride=8, options=None):
        """One stage (FEAT_TRUNK / FEAT_LEVEL1 / FEAT_LEVEL2, or FEAT_ALL) on the current stream."""
        n, _, H, W = src_inps.shape
        f0, f1, f2, workspace = bufs
        self._check(self.dll.enerf_feature_net_stage(_ptr(packed), _ptr(src_inps), n, H, W, _ptr(f0), _ptr(f1),
                                                     _ptr(f2), l2_stride, _ptr(workspace), workspace.numel() * 4,
                                                     int(stage), _opt(options), self.stream_of(src_inps)),
                    "feature_net_stage")

    def feature_net(self, packed, src_inps, l2_stride=8, workspace=None, options=None):
        """src_inps (n,3,H,W) -> channels-last (n,H/4,W/4,32), (n,H/2,W/2,16), (n,H,W,l2_stride)."""
        n, _, H, W = src_inps.shape
        f0, f1, f2, workspace = self.feature_net_alloc(src_inps, l2_stride, workspace)
        self._check(self.dll.enerf_feature_net(_ptr(packed), _ptr(src_inps), n, H, W, _ptr(f0), _ptr(f1), _ptr(f2),
                                               l2_stride, _ptr(workspace), workspace.numel() * 4, _opt(options),
                                               self.stream_of(src_inps)), "feature_net")
        return f0, f1, f2, workspace

    def pack_texels_cl(self, feat_cl, src_inps, Hr, Wr):
        n_img, hf, wf, Cf = feat_cl.shape
        if (hf, wf) != (Hr, Wr):
            raise EnerfError("pack_texels_cl: features must already be at the render resolution")
        H, W = src_inps.shape[-2:]
        tex = 4 * ((Cf + 3 + 3) // 4)
        out = torch.empty((n_img, Hr, Wr, tex), dtype=torch.float32, device=feat_cl.device)
        self._check(self.dll.enerf_pack_texels_cl(_ptr(feat_cl), Cf, _ptr(src_inps), H, W, Hr, Wr, tex, n_img, _ptr(out),
                                                  self.stream_of(out)), "pack_texels_cl")
        return out

    def get_proj_mats(self, src_ixts, src_exts, tar_ixt, tar_ext, src_scale, tar_scale):
        B, S = src_ixts.shape[:2]
        proj = torch.empty((B, S, 3, 4), dtype=torch.float32, device=src_ixts.device)
        self._check(self.dll.enerf_get_proj_mats(_ptr(src_ixts), _ptr(src_exts), _ptr(tar_ixt), _ptr(tar_ext), B, S,
                                                 float(src_scale), float(tar_scale), _ptr(proj),
                                                 self.stream_of(proj)), "get_proj_mats")
        return proj

    def get_depth_values(self, near_far, prev, B, D, h, w, depth_inv):
        dev = near_far.device
        dv = torch.empty((B, D, h, w), dtype=torch.float32, device=dev)
        nf = torch.empty((B, 2, h, w), dtype=torch.float32, device=dev)
        if prev is None:
            pd = ps = pn = None
            hp = wp = 0
        else:
            pd, ps, pn = prev
            hp, wp = pd.shape[-2:]
        self._check(self.dll.enerf_get_depth_values(_ptr(near_far), _ptr(pd), _ptr(ps), _ptr(pn), B, D, h, w, hp, wp,
                                                    int(depth_inv), _ptr(dv), _ptr(nf), self.stream_of(dv)),
                    "get_depth_values")
        return dv, nf

    def level_prep(self, src_ixts, src_exts, tar_ixt, tar_ext, src_scale, tar_scale, near_far, prev, D, h, w, depth_inv):
        """get_proj_mats + get_depth_values of one level in one launch -> (proj, depth_values, near_far)."""
        B, S = src_ixts.shape[:2]
        dev = src_ixts.device
        proj = torch.empty((B, S, 3, 4), dtype=torch.float32, device=dev)
        dv = torch.empty((B, D, h, w), dtype=torch.float32, device=dev)
        nf = torch.empty((B, 2, h, w), dtype=torch.float32, device=dev)
        if prev is None:
            pd = ps = pn = None
            hp = wp = 0
        else:
            pd, ps, pn = prev
            hp, wp = pd.shape[-2:]
        self._check(self.dll.enerf_level_prep(_ptr(src_ixts), _ptr(src_exts), _ptr(tar_ixt), _ptr(tar_ext), B, S,
                                              float(src_scale), float(tar_scale), _ptr(proj), _ptr(near_far), _ptr(pd),
                                              _ptr(ps), _ptr(pn), D, h, w, hp, wp, int(depth_inv), _ptr(dv), _ptr(nf),
                                              self.stream_of(dv)), "level_prep")
        return proj, dv, nf

    def build_feature_volume(self, feat_cl, proj, dv, Cc):
        B, S, Hs, Ws = feat_cl.shape[:4]
        _, D, h, w = dv.shape
        vol = torch.empty((B, D, h, w, Cc), dtype=torch.float32, device=dv.device)
        self._check(self.dll.enerf_build_feature_volume(_ptr(feat_cl), _ptr(proj), _ptr(dv), B, S, Cc, Hs, Ws, D, h, w,
                                                        _ptr(vol), self.stream_of(vol)), "build_feature_volume")
        return vol

    def cost_reg_pack(self, raw: CostRegRaw, device):
        n = self.dll.enerf_cost_reg_packed_floats(raw.in_channels, raw.full)
        packed = torch.empty((n,), dtype=torch.float32, device=device)
        self._check(self.dll.enerf_cost_reg_pack(C.byref(raw), _ptr(packed), self.stream_of(packed)), "cost_reg_pack")
        return packed

    def cost_reg(self, packed, in_channels, full, vol, workspace=None, options=None):
        B, D, h, w, _ = vol.shape
        need = self.dll.enerf_cost_reg_workspace_bytes(int(full), B, D, h, w)
        if workspace is None or workspace.numel() * 4 < need:
            workspace = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=vol.device)
        feat = torch.empty((B, D, h, w, 8), dtype=torch.float32, device=vol.device)
        prob = torch.empty((B, D, h, w), dtype=torch.float32, device=vol.device)
        self._check(self.dll.enerf_cost_reg(_ptr(packed), in_channels, int(full), _ptr(vol), B, D, h, w, _ptr(feat),
                                            _ptr(prob), _ptr(workspace), workspace.numel() * 4, _opt(options),
                                            self.stream_of(vol)), "cost_reg")
        return feat, prob

    def depth_regression(self, prob, dv, depth_inv):
        B, D, h, w = prob.shape
        depth = torch.empty((B, h, w), dtype=torch.float32, device=prob.device)
        std = torch.empty_like(depth)
        self._check(self.dll.enerf_depth_regression(_ptr(prob), _ptr(dv), B, D, h, w, int(depth_inv), _ptr(depth),
                                                    _ptr(std), self.stream_of(prob)), "depth_regression")
        return depth, std

    def build_rays(self, rays8, depth, std, near_far, Hr, Wr, depth_inv):
        B, N = rays8.shape[:2]
        h, w = depth.shape[-2:]
        out = torch.empty((B, N, 12), dtype=torch.float32, device=rays8.device)
        self._check(self.dll.enerf_build_rays(_ptr(rays8), _ptr(depth), _ptr(std), _ptr(near_far), B, N, h, w, Hr, Wr,
                                              int(depth_inv), _ptr(out), self.stream_of(out)), "build_rays")
        return out

    def nerf_pack(self, raw: NerfRaw, F: int, viewdir_agg: bool, device):
        n = self.dll.enerf_nerf_packed_floats(F)
        packed = torch.empty((n,), dtype=torch.float32, device=device)
        self._check(self.dll.enerf_nerf_pack(C.byref(raw), F, int(viewdir_agg), _ptr(packed),
                                             self.stream_of(packed)), "nerf_pack")
        return packed

    def render_rays(self, rays12, tex, vol, src_exts, src_ixts, tar_ext, packed, *, n_samples, depth_inv, F,
                    render_scale, white_bkgd=False, maps=None, options=None):
        """``rays12`` (B,N,12) from build_rays — or, with ``maps=(depth, std, near_far)`` of the level, the
        8-float rays (B,N,8): build_rays then runs inside the render kernel."""
        B, N = rays12.shape[:2]
        if (rays12.shape[-1] != 12) != (maps is not None):
            raise EnerfError("render_rays: pass (B,N,12) rays, or (B,N,8) rays together with maps=(depth,std,near_far)")
        S, Hr, Wr = tex.shape[1:4]
        _, D, h, w, _ = vol.shape
        dev = rays12.device
        rgb = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        depth = torch.empty((B, N), dtype=torch.float32, device=dev)
        weights = torch.empty((B, N, n_samples), dtype=torch.float32, device=dev)
        if maps is None:
            fused = (None, None, None, None, 0, 0)
            r12 = _ptr(rays12)
        else:
            md, ms, mn = maps
            fused = (_ptr(rays12), _ptr(md), _ptr(ms), _ptr(mn), int(md.shape[-2]), int(md.shape[-1]))
            r12 = None
        a = RenderArgs(r12, _ptr(tex), _ptr(vol), _ptr(src_exts), _ptr(src_ixts), _ptr(tar_ext),
                       _ptr(packed), _ptr(rgb), _ptr(depth), _ptr(weights), B, N, S, n_samples, int(depth_inv), Hr,
                       Wr, F, D, h, w, int(white_bkgd), float(render_scale), *fused,
                       None if options is None else C.pointer(options))
        self._check(self.dll.enerf_render_rays(C.byref(a), self.stream_of(rays12)), "render_rays")
        return rgb, depth, weights


    # -- backward kernels (training path; wrapped by enerf_amd/autograd.py) -----------------------------
    def build_feature_volume_bwd(self, feat_cl, proj, dv, grad_vol):
        B, S, Hs, Ws, Cc = feat_cl.shape
        _, D, h, w = dv.shape
        gfeat = torch.empty_like(feat_cl)
        gdv = torch.empty_like(dv)
        self._check(self.dll.enerf_build_feature_volume_bwd(_ptr(feat_cl), _ptr(proj), _ptr(dv), _ptr(grad_vol), B, S, Cc, Hs, Ws,
                                                            D, h, w, _ptr(gfeat), _ptr(gdv), self.stream_of(dv)),
                    "build_feature_volume_bwd")
        return gfeat, gdv

    def depth_regression_bwd(self, prob, dv, g_depth, g_std, depth_inv):
        B, D, h, w = prob.shape
        gp, gdv = torch.empty_like(prob), torch.empty_like(dv)
        self._check(self.dll.enerf_depth_regression_bwd(_ptr(prob), _ptr(dv), _ptr(g_depth), _ptr(g_std), B, D, h, w,
                                                        int(depth_inv), _ptr(gp), _ptr(gdv), self.stream_of(prob)),
                    "depth_regression_bwd")
        return gp, gdv

    # training-mode 3-D conv layers / BatchNorm pieces (train.hip); kind: 0 stride 1, 1 stride 2, 2 transposed stride 2
    def conv3d_layer_pack(self, w, cin, cout, kind):
        packed = torch.empty((self.dll.enerf_conv3d_layer_packed_floats(cin, cout, kind),), dtype=torch.float32, device=w.device)
        self._check(self.dll.enerf_conv3d_layer_pack(_ptr(w), cin, cout, kind, _ptr(packed), self.stream_of(w)), "conv3d_layer_pack")
        return packed

    def conv3d_layer(self, packed, cin, cout, kind, x_cl, residual=None, options=None):
        """x_cl (B,D,h,w,cin) channels-last -> (B,D',h',w',cout); D' = D (kind 0), ceil(D/2) (kind 1), 2D (kind 2)."""
        B, D, h, w, _ = x_cl.shape
        if kind == 1:
            Do, Ho, Wo = (D - 1) // 2 + 1, (h - 1) // 2 + 1, (w - 1) // 2 + 1
        elif kind == 2:
            Do, Ho, Wo = 2 * D, 2 * h, 2 * w
        else:
            Do, Ho, Wo = D, h, w
        out = torch.empty((B, Do, Ho, Wo, cout), dtype=torch.float32, device=x_cl.device)
        self._check(self.dll.enerf_conv3d_layer(_ptr(packed), cin, cout, kind, _ptr(x_cl), _ptr(residual), _ptr(out), B, D, h, w,
                                                _opt(options), self.stream_of(x_cl)), "conv3d_layer")
        return out

    # training-mode FeatureNet convolutions (train.hip: enerf_conv2d_layer): weights (cout,cin,k,k), padding (k-1)/2
    def conv2d_layer_pack(self, w, bias, cin, cout, k):
        packed = torch.empty((self.dll.enerf_conv2d_layer_packed_floats(cin, cout, k),), dtype=torch.float32, device=w.device)
        self._check(self.dll.enerf_conv2d_layer_pack(_ptr(w), _ptr(bias), cin, cout, k, _ptr(packed), self.stream_of(w)), "conv2d_layer_pack")
        return packed

    def conv2d_layer(self, packed, cin, cout, k, stride, x, up=None):
        """x channels-last (N,H,W,cin) — for cin = 3 the NCHW image batch (N,3,H,W) — -> channels-last (N,Ho,Wo,cout);
        ``up`` (N,Ho/2,Wo/2,cout) is upsampled 2x (bilinear, align_corners) and added."""
        if cin == 3:
            N, _, H, W = x.shape
        else:
            N, H, W, _ = x.shape
        P = (k - 1) // 2
        Ho, Wo = (H + 2 * P - k) // stride + 1, (W + 2 * P - k) // stride + 1
        out = torch.empty((N, Ho, Wo, cout), dtype=torch.float32, device=x.device)
        self._check(self.dll.enerf_conv2d_layer(_ptr(packed), cin, cout, k, stride, _ptr(x), _ptr(up), _ptr(out), N, H, W,
                                                self.stream_of(x)), "conv2d_layer")
        return out

    def channel_sums(self, a, b, z_mask=None, mask_scale=None, mask_shift=None):
        """(sum_p a*m, sum_p a*m*b) per channel in fp64; tensors channels-last (..., C)."""
        Cc = a.shape[-1]
        sums = torch.empty((2, Cc), dtype=torch.float64, device=a.device)
        n = a.numel() // Cc
        nb = self.dll.enerf_channel_sums_workspace_bytes(n, Cc)
        ws = self._scratch(nb, a.device)
        self._check(self.dll.enerf_channel_sums_ws(_ptr(a), _ptr(b), _ptr(z_mask), _ptr(mask_scale), _ptr(mask_shift),
                                                   n, Cc, sums.data_ptr(), ws.data_ptr(), nb, self.stream_of(a)), "channel_sums")
        return sums[0], sums[1]

    def up2_adjoint(self, g_fine, add=None):
        """Adjoint of the 2x align-corners bilinear upsampling on a channels-last gradient (N,2h,2w,C) -> (N,h,w,C) (+ add)."""
        N, Hf, Wf, Cc = g_fine.shape
        out = torch.empty((N, Hf // 2, Wf // 2, Cc), dtype=torch.float32, device=g_fine.device)
        self._check(self.dll.enerf_up2_adjoint(_ptr(g_fine), _ptr(add), N, Hf // 2, Wf // 2, Cc, _ptr(out), self.stream_of(g_fine)),
                    "up2_adjoint")
        return out

    def channel_sums_raw(self, a, b, z_mask=None, mask_scale=None, mask_shift=None):
        """channel_sums as ONE (2, C) fp64 tensor [sum a*m ; sum a*m*b] (what the all-reduce and the coefficient kernels take)."""
        Cc = a.shape[-1]
        sums = torch.empty((2, Cc), dtype=torch.float64, device=a.device)
        n = a.numel() // Cc
        nb = self.dll.enerf_channel_sums_workspace_bytes(n, Cc)
        ws = self._scratch(nb, a.device)
        self._check(self.dll.enerf_channel_sums_ws(_ptr(a), _ptr(b), _ptr(z_mask), _ptr(mask_scale), _ptr(mask_shift),
                                                   n, Cc, sums.data_ptr(), ws.data_ptr(), nb, self.stream_of(a)), "channel_sums")
        return sums

    def bn_train_coeffs(self, sums, count, bn):
        """(2,C) fp64 sums + position count (python number, or a 1-element fp64 device tensor) + the BatchNorm module ->
        mean_invstd (2,C) fp64, scale_shift (2,C) fp32; updates the module's running statistics in place."""
        Cc = sums.shape[1]
        mi = torch.empty((2, Cc), dtype=torch.float64, device=sums.device)
        ss = torch.empty((2, Cc), dtype=torch.float32, device=sums.device)
        track = bn.track_running_stats and bn.running_mean is not None      # (num_batches_tracked += 1 happens in the kernel)
        cd, ch = (count.data_ptr(), 0.0) if torch.is_tensor(count) else (None, float(count))
        self._check(self.dll.enerf_bn_train_coeffs(sums.data_ptr(), cd, ch, _ptr(bn.weight.detach()), _ptr(bn.bias.detach()), float(bn.eps),
                                                   -1.0 if bn.momentum is None else float(bn.momentum),
                                                   _ptr(bn.running_mean) if track else None, _ptr(bn.running_var) if track else None,
                                                   bn.num_batches_tracked.data_ptr() if track else None, 1, Cc, mi.data_ptr(), _ptr(ss),
                                                   self.stream_of(sums)), "bn_train_coeffs")
        return mi, ss

    def bn_stats_fit(self, z) -> bool:
        """the channel widths enerf_channel_sums / enerf_bn_train_stats handle (C in 4..64, C/4 dividing 256)"""
        Cc = z.shape[-1]
        return 4 <= Cc <= 64 and Cc % 4 == 0 and 256 % (Cc // 4) == 0

    def bn_train_stats(self, z, bn):
        """ABI v8: batch statistics of z (..., C) AND the forward coefficients of the BatchNorm module in two launches (no
        SyncBatchNorm): -> mean_invstd (2,C) fp64, scale_shift (2,C) fp32; updates the running statistics in place."""
        Cc = z.shape[-1]
        n = z.numel() // Cc
        nb = self.dll.enerf_channel_sums_workspace_bytes(n, Cc)
        ws = self._scratch(nb, z.device)
        mi = torch.empty((2, Cc), dtype=torch.float64, device=z.device)
        ss = torch.empty((2, Cc), dtype=torch.float32, device=z.device)
        track = bn.track_running_stats and bn.running_mean is not None
        self._check(self.dll.enerf_bn_train_stats(_ptr(z), n, Cc, ws.data_ptr(), nb, _ptr(bn.weight.detach()), _ptr(bn.bias.detach()),
                                                  float(bn.eps), -1.0 if bn.momentum is None else float(bn.momentum),
                                                  _ptr(bn.running_mean) if track else None, _ptr(bn.running_var) if track else None,
                                                  bn.num_batches_tracked.data_ptr() if track else None, 1, None, mi.data_ptr(), _ptr(ss),
                                                  self.stream_of(z)), "bn_train_stats")
        return mi, ss, n

    def bn_train_bwd_stats(self, g, z, mean_invstd, scale, z_mask=None, mask_scale=None, mask_shift=None):
        """ABI v8: sums of [g*m, g*m*z] AND the backward coefficients in two launches -> dgamma_dbeta (2,C), k2k3 (2,C)."""
        Cc = z.shape[-1]
        n = z.numel() // Cc
        nb = self.dll.enerf_channel_sums_workspace_bytes(n, Cc)
        ws = self._scratch(nb, z.device)
        dgb = torch.empty((2, Cc), dtype=torch.float32, device=z.device)
        k23 = torch.empty((2, Cc), dtype=torch.float32, device=z.device)
        self._check(self.dll.enerf_bn_train_bwd_stats(_ptr(g), _ptr(z), _ptr(z_mask), _ptr(mask_scale), _ptr(mask_shift), n, Cc, ws.data_ptr(), nb,
                                                      mean_invstd.data_ptr(), _ptr(scale), _ptr(dgb), _ptr(k23), self.stream_of(z)),
                    "bn_train_bwd_stats")
        return dgb, k23

    def bn_train_apply(self, z, bn, residual=None, relu=False):
        """ABI v11: training-mode BatchNorm of z (..., C) (+ ReLU) (+ residual) in two launches for small / mid layers, three otherwise
        (enerf_bn_train_apply) -> (out, mean_invstd (2,C) fp64, scale_shift (2,C) fp32, positions); updates the running statistics."""
        Cc = z.shape[-1]
        n = z.numel() // Cc
        nb = self.dll.enerf_channel_sums_workspace_bytes(n, Cc)
        ws = self._scratch(nb, z.device)
        mi = torch.empty((2, Cc), dtype=torch.float64, device=z.device)
        ss = torch.empty((2, Cc), dtype=torch.float32, device=z.device)
        out = torch.empty_like(z)
        track = bn.track_running_stats and bn.running_mean is not None
        self._check(self.dll.enerf_bn_train_apply(_ptr(z), n, Cc, ws.data_ptr(), nb, _ptr(bn.weight.detach()), _ptr(bn.bias.detach()),
                                                  float(bn.eps), -1.0 if bn.momentum is None else float(bn.momentum),
                                                  _ptr(bn.running_mean) if track else None, _ptr(bn.running_var) if track else None,
                                                  bn.num_batches_tracked.data_ptr() if track else None, 1, mi.data_ptr(), _ptr(ss),
                                                  _ptr(residual), int(relu), _ptr(out), self.stream_of(z)), "bn_train_apply")
        return out, mi, ss, n

    def bn_train_bwd_apply(self, g, z, mean_invstd, scale_shift, relu):
        """ABI v11: (grad_z, dgamma_dbeta (2,C)) of the BatchNorm (+ ReLU) whose forward was bn_train_apply (enerf_bn_train_bwd_apply)."""
        Cc = z.shape[-1]
        n = z.numel() // Cc
        nb = self.dll.enerf_channel_sums_workspace_bytes(n, Cc) + 2 * Cc * 4
        ws = self._scratch(nb, z.device)
        dgb = torch.empty((2, Cc), dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        self._check(self.dll.enerf_bn_train_bwd_apply(_ptr(g), _ptr(z), int(relu), n, Cc, ws.data_ptr(), nb, mean_invstd.data_ptr(),
                                                      _ptr(scale_shift), _ptr(dgb), _ptr(dz), self.stream_of(z)), "bn_train_bwd_apply")
        return dz, dgb

    def bn_train_bwd_coeffs(self, local, glob, count, mean_invstd, scale):
        Cc = local.shape[1]
        dgb = torch.empty((2, Cc), dtype=torch.float32, device=local.device)
        k23 = torch.empty((2, Cc), dtype=torch.float32, device=local.device)
        cd, ch = (count.data_ptr(), 0.0) if torch.is_tensor(count) else (None, float(count))
        self._check(self.dll.enerf_bn_train_bwd_coeffs(local.data_ptr(), glob.data_ptr(), cd, ch, mean_invstd.data_ptr(), _ptr(scale), Cc,
                                                       _ptr(dgb), _ptr(k23), self.stream_of(local)), "bn_train_bwd_coeffs")
        return dgb, k23

    def channel_affine(self, a, p, r, b=None, q=None, z_mask=None, mask_scale=None, mask_shift=None, residual=None, relu=False):
        Cc = a.shape[-1]
        out = torch.empty_like(a)
        self._check(self.dll.enerf_channel_affine(_ptr(a), _ptr(b), _ptr(p), _ptr(q), _ptr(r), _ptr(z_mask), _ptr(mask_scale),
                                                  _ptr(mask_shift), _ptr(residual), int(relu), a.numel() // Cc, Cc, _ptr(out),
                                                  self.stream_of(a)), "channel_affine")
        return out

    def gemm_wgrad(self, a, b, Ca=None, Cb=None, bias=False):
        """grad_w (Ca,Cb) = sum over rows p of a[p,:Ca]^T b[p,:Cb]; a, b 2-D row-major (possibly column slices: a view whose
        rows are wider than Ca is passed by its base pointer + row stride).  bias=True -> (grad_w, grad_bias (Ca) = column
        sums of a) from the same pass."""
        P = a.shape[0]
        Ca, Cb = Ca or a.shape[1], Cb or b.shape[1]
        if a.stride(1) != 1 or b.stride(1) != 1:
            raise EnerfError("gemm_wgrad: rows must be contiguous")
        gw = torch.empty((Ca, Cb), dtype=torch.float32, device=a.device)
        gb = torch.empty((Ca,), dtype=torch.float32, device=a.device) if bias else None
        ws = self._scratch(self.dll.enerf_gemm_wgrad_workspace_bytes(P, Ca, Cb, int(bias)), a.device)
        self._check(self.dll.enerf_gemm_wgrad(a.data_ptr(), a.stride(0), Ca, b.data_ptr(), b.stride(0), Cb, P, _ptr(gw),
                                              _ptr(gb), ws.data_ptr(), ws.numel(), self.stream_of(a)), "gemm_wgrad")
        return (gw, gb) if bias else gw

    SELFTEST_NAMES = ("xor16", "xor32", "group_sum4", "group_max4", "row_sum16", "add_xor8", "group_bcast<2>", "group_bcast<4>",
                      "group_bcast<8>", "glds16 + raw barrier", "raw buffer loads", "mfma 16x16x4 layout", "mfma 4x4x1 broadcast-A",
                      "mfma 4x4x1", "relu1 (med3)", "mul24", "lds / global fp32 atomics", "rcp / sqrt / exp", "wave_sync", "xcd_contiguous")

    def selftest_primitives(self, device, blocks: int = 24):
        """{check name: lanes that disagree with the memory-only specification} (enerf_selftest_primitives): all zero on a sound build."""
        n = self.dll.enerf_selftest_checks()
        assert n == len(self.SELFTEST_NAMES)
        g = torch.Generator().manual_seed(5)
        table = torch.randn(4096, generator=g).to(device)
        scratch = torch.empty(blocks * 16, dtype=torch.float32, device=device)
        bad = torch.empty(n, dtype=torch.int32, device=device)
        self._check(self.dll.enerf_selftest_primitives(_ptr(table), table.numel(), _ptr(scratch), blocks, bad.data_ptr(),
                                                       self.stream_of(table)), "selftest_primitives")
        return dict(zip(self.SELFTEST_NAMES, (int(v) for v in bad.cpu())))

    def gemm_wgrad_group(self, members):
        """Several ``gemm_wgrad`` calls as TWO launches (enerf_gemm_wgrad_group; every gradient bit-identical to its single call).
        ``members``: dicts with ``a``, ``b`` (2-D row-major, possibly column slices), optional ``Cb`` (leading columns of b used),
        ``bias`` (also the column sums of a) and ``into = (matrix (Ca, W), first column)`` — the gradient is a column block of a
        wider weight matrix and is written in place.  Returns one (grad_w or the ``into`` matrix, grad_bias or None) per member."""
        n = len(members)
        descs, out = (GemmWgradDesc * n)(), []
        for d, m in zip(descs, members):
            if m.get("partials") is not None:                  # first stage done elsewhere (nerf_mlp_bwd(partials=True)): reduce only
                part, (mat, c0) = m["partials"], m["into"]
                Ca, Cb = m["Ca"], m["Cb"]
                if part.shape[1] * 16 * 16 < ((Ca + 15) // 16) * ((Cb + 15) // 16) * 256 or not part.is_contiguous() or not mat.is_contiguous():
                    raise EnerfError("gemm_wgrad_group: partial rows do not hold the member's tiles")
                d.Ca, d.Cb, d.grad_w, d.ldw = Ca, Cb, mat.data_ptr() + 4 * c0, mat.shape[1]
                d.partials, d.partial_chunks = part.data_ptr(), part.shape[0]
                out.append((mat, None))
                continue
            a, b = m["a"], m["b"]
            if a.stride(1) != 1 or b.stride(1) != 1 or a.shape[0] != b.shape[0]:
                raise EnerfError("gemm_wgrad_group: rows must be contiguous and of equal count")
            Ca, Cb = a.shape[1], m.get("Cb") or b.shape[1]
            gb = torch.empty((Ca,), dtype=torch.float32, device=a.device) if m.get("bias") else None
            if m.get("into") is not None:
                mat, c0 = m["into"]
                if mat.shape[0] != Ca or not mat.is_contiguous() or c0 + Cb > mat.shape[1]:
                    raise EnerfError("gemm_wgrad_group: `into` matrix does not hold the column block")
                gw, ptr, ldw = mat, mat.data_ptr() + 4 * c0, mat.shape[1]
            else:
                gw = torch.empty((Ca, Cb), dtype=torch.float32, device=a.device)
                ptr, ldw = gw.data_ptr(), Cb
            d.a, d.lda, d.Ca, d.b, d.ldb, d.Cb, d.P = a.data_ptr(), a.stride(0), Ca, b.data_ptr(), b.stride(0), Cb, a.shape[0]
            d.grad_w, d.ldw, d.grad_bias = ptr, ldw, _ptr(gb)
            out.append((gw, gb))
        a0 = next(m["a"] for m in members if m.get("partials") is None)
        ws = self._scratch(self.dll.enerf_gemm_wgrad_group_workspace_bytes(descs, n), a0.device)
        self._check(self.dll.enerf_gemm_wgrad_group(descs, n, ws.data_ptr(), ws.numel(), self.stream_of(a0)), "gemm_wgrad_group")
        return out

    def nerf_mlp_fwd(self, vox, x, packed, S, F):
        raw = torch.empty((vox.shape[0], 4), dtype=torch.float32, device=vox.device)
        self._check(self.dll.enerf_nerf_mlp_fwd(_ptr(vox), _ptr(x), _ptr(packed), vox.shape[0], S, F, _ptr(raw),
                                                self.stream_of(vox)), "nerf_mlp_fwd")
        return raw

    def nerf_mlp_bwd(self, vox, x, g_raw, packed, bimg, offsets, S, F, level=0):
        """Fused MLP backward (enerf_nerf_mlp_bwd) -> (g_vox, g_x, saves: list of 16 tensors).  ``level`` 1 / 2 (F = 11; 2: S <= 3): the
        weight gradients of the per-view colour (and aggregation) branch are accumulated in the kernel (enerf_nerf_mlp_bwd_partials): the
        saves they replace are None and a dict of per-wave partial sums follows — "q" (chunks, 4, 256), "rows" (chunks, 128) and, level 2,
        "g" (chunks, 2, 256), "v" (chunks, 1, 256)."""
        P, dev = vox.shape[0], vox.device
        E = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        g_vox, g_x = E(P, 8), E(P, S, F + 4)
        skip = () if level == 0 else ((2, 6, 7) if level == 1 else (2, 6, 7, 3, 4, 12, 13, 15))
        shapes = [(P, 88), (P, 32), (P, S, 64), (P, S, 32), (P, S, F), (P, 2 * F),
                  (P, S), (P, S, 64), (P, 64), (P,), (P, 64), (P, 16), (P, S), (P, S, 32), (P, 32), (P, S, F)]
        saves = [None if i in skip else E(*sh) for i, sh in enumerate(shapes)]
        a = MlpBwdArgs(_ptr(vox), _ptr(x), _ptr(g_raw), _ptr(packed), _ptr(bimg), _ptr(g_vox), _ptr(g_x))
        for i, t in enumerate(saves):
            a.save[i] = _ptr(t)
        a.P, a.F, a.S = P, F, S
        for i, o in enumerate(offsets):
            a.image_offsets[i] = int(o)
        if level == 0:
            self._check(self.dll.enerf_nerf_mlp_bwd(C.byref(a), self.stream_of(vox)), "nerf_mlp_bwd")
            return g_vox, g_x, saves
        chunks = int(self.dll.enerf_nerf_mlp_bwd_chunks(P))
        part = {"q": E(chunks, 4, 256), "rows": E(chunks, 128)}
        if level == 2:
            part["g"], part["v"] = E(chunks, 2, 256), E(chunks, 1, 256)
        self._check(self.dll.enerf_nerf_mlp_bwd_partials(C.byref(a), level, _ptr(part["q"]), _ptr(part.get("g")), _ptr(part.get("v")),
                                                         _ptr(part["rows"]), self.stream_of(vox)), "nerf_mlp_bwd_partials")
        return g_vox, g_x, saves, part

    def colsum(self, part):
        """out[i] = sum_c part[c, i] in a fixed order (enerf_colsum): the second stage of per-wave partial sums."""
        chunks, n = part.shape
        out = torch.empty((n,), dtype=torch.float32, device=part.device)
        self._check(self.dll.enerf_colsum(_ptr(part), chunks, n, _ptr(out), self.stream_of(part)), "colsum")
        return out

    def _gather_args(self, xyz, dn, uv, tex_cl, vol_cl, cam, tcen):
        B, P = xyz.shape[0], xyz.shape[1]
        a = GatherArgs(_ptr(xyz), _ptr(dn), _ptr(uv), _ptr(tex_cl), _ptr(vol_cl), _ptr(cam), _ptr(tcen))
        a.P, a.B, a.S, a.F = P, B, tex_cl.shape[1], tex_cl.shape[4]
        a.Hr, a.Wr = tex_cl.shape[2], tex_cl.shape[3]
        a.D, a.h, a.w = vol_cl.shape[1], vol_cl.shape[2], vol_cl.shape[3]
        assert vol_cl.shape[4] == 8 and cam.shape[-1] == 16 and tcen.shape[-1] == 4
        return a

    def gather_fwd(self, xyz, dn, uv, tex_cl, vol_cl, cam, tcen):
        """Training-path feature fetch (enerf_gather_fwd): xyz (B,P,3), dn (B,P), uv (B,P,2), tex_cl (B,S,Hr,Wr,F),
        vol_cl (B,D,h,w,8), cam (B,S,16), tcen (B,4) -> x (B,P,S,F+4), vox (B,P,8)."""
        a = self._gather_args(xyz, dn, uv, tex_cl, vol_cl, cam, tcen)
        x = torch.empty((a.B, a.P, a.S, a.F + 4), dtype=torch.float32, device=xyz.device)
        vox = torch.empty((a.B, a.P, 8), dtype=torch.float32, device=xyz.device)
        a.x, a.vox = _ptr(x), _ptr(vox)
        self._check(self.dll.enerf_gather_fwd(C.byref(a), self.stream_of(xyz)), "gather_fwd")
        return x, vox

    def gather_bwd(self, xyz, dn, uv, tex_cl, vol_cl, cam, tcen, g_x, g_vox, n_samples=0, ray_w=0):
        """-> (g_tex_cl, g_vol_cl, g_xyz, g_dn).  ``n_samples`` / ``ray_w``: raster hints (enerf_gather_args_t)."""
        a = self._gather_args(xyz, dn, uv, tex_cl, vol_cl, cam, tcen)
        a.n_samples, a.ray_w = int(n_samples), int(ray_w)
        E = lambda ref: torch.empty_like(ref)
        g_xyz, g_dn = E(xyz), E(dn)
        # the two scatter targets back to back: the library zeroes them with one launch (gather.hip zero_async2)
        # (when the first keeps the second 256-byte aligned; otherwise two allocations, two launches)
        if tex_cl.numel() % 64 == 0:
            acc = torch.empty((tex_cl.numel() + vol_cl.numel(),), dtype=torch.float32, device=xyz.device)
            g_tex, g_vol = acc[:tex_cl.numel()].view(tex_cl.shape), acc[tex_cl.numel():].view(vol_cl.shape)
        else:
            g_tex, g_vol = E(tex_cl), E(vol_cl)
        a.g_x, a.g_vox, a.g_tex, a.g_vol, a.g_xyz, a.g_dn = (_ptr(g_x), _ptr(g_vox), _ptr(g_tex), _ptr(g_vol), _ptr(g_xyz),
                                                            _ptr(g_dn))
        self._check(self.dll.enerf_gather_bwd(C.byref(a), self.stream_of(xyz)), "gather_bwd")
        return g_tex, g_vol, g_xyz, g_dn

    def conv_wgrad_cl(self, a_cl, b_cl, stride):
        """3x3x3 weight gradient from channels-last tensors a (n,Da,Ha,Wa,Ca), b (n,Db,Hb,Wb,Cb) -> (Ca,Cb,3,3,3)."""
        n, Da, Ha, Wa, Ca = a_cl.shape
        _, Db, Hb, Wb, Cb = b_cl.shape
        gw = torch.empty((Ca, Cb, 3, 3, 3), dtype=torch.float32, device=a_cl.device)
        ws = self._scratch(self.dll.enerf_conv_wgrad_workspace_bytes(n * Da * Ha * Wa, Ca, Cb, 3, 3, 3), a_cl.device)
        self._check(self.dll.enerf_conv_wgrad(_ptr(a_cl), _ptr(b_cl), n, Da, Ha, Wa, Ca, Db, Hb, Wb, Cb, 3, 3, 3, int(stride),
                                              1, 1, 1, _ptr(gw), ws.data_ptr(), ws.numel(), self.stream_of(a_cl)), "conv_wgrad")
        return gw

    def conv_wgrad_cl2d(self, a_cl, b_cl, k, stride):
        """k x k weight gradient (padding (k-1)/2) from channels-last 2-D tensors a (n,Ha,Wa,Ca), b (n,Hb,Wb,Cb) -> (Ca,Cb,k,k)."""
        n, Ha, Wa, Ca = a_cl.shape
        _, Hb, Wb, Cb = b_cl.shape
        p = (k - 1) // 2
        gw = torch.empty((Ca, Cb, k, k), dtype=torch.float32, device=a_cl.device)
        ws = self._scratch(self.dll.enerf_conv_wgrad_workspace_bytes(n * Ha * Wa, Ca, Cb, 1, k, k), a_cl.device)
        self._check(self.dll.enerf_conv_wgrad(_ptr(a_cl), _ptr(b_cl), n, 1, Ha, Wa, Ca, 1, Hb, Wb, Cb, 1, k, k, int(stride),
                                              0, p, p, _ptr(gw), ws.data_ptr(), ws.numel(), self.stream_of(a_cl)), "conv_wgrad")
        return gw

    def conv_wgrad(self, a, b, kernel, stride, padding, bias=False):
        """Weight gradient on the matrix cores (enerf_conv_wgrad).  ``a`` (n,Ca,*grid_a), ``b`` (n,Cb,*grid_b) in torch's
        channels-first layout (converted to channels-last here by the library's own adapter kernel); 2-D or 3-D.
        Returns (Ca, Cb, *kernel); with ``bias`` also the per-channel sums of ``a`` (the bias gradient when a = d y)."""
        nd = a.dim() - 2
        ga, gb = list(a.shape[2:]), list(b.shape[2:])
        if nd == 2:
            ga, gb, kernel, padding = [1] + ga, [1] + gb, (1,) + tuple(kernel), (0,) + tuple(padding)
        n, Ca, Cb = a.shape[0], a.shape[1], b.shape[1]
        pa, pb = ga[0] * ga[1] * ga[2], gb[0] * gb[1] * gb[2]
        a_cl = self.channels_last(a.contiguous().reshape(n, Ca, pa), n, Ca, pa)
        b_cl = self.channels_last(b.contiguous().reshape(n, Cb, pb), n, Cb, pb)
        gw = torch.empty((Ca, Cb) + tuple(kernel), dtype=torch.float32, device=a.device)
        ws = self._scratch(self.dll.enerf_conv_wgrad_workspace_bytes(n * pa, Ca, Cb, kernel[0], kernel[1], kernel[2]), a.device)
        self._check(self.dll.enerf_conv_wgrad(_ptr(a_cl), _ptr(b_cl), n, ga[0], ga[1], ga[2], Ca, gb[0], gb[1], gb[2], Cb,
                                              kernel[0], kernel[1], kernel[2], int(stride), padding[0], padding[1], padding[2],
                                              _ptr(gw), ws.data_ptr(), ws.numel(), self.stream_of(a)), "conv_wgrad")
        gw = gw if nd == 3 else gw.reshape(Ca, Cb, kernel[1], kernel[2])
        if not bias:
            return gw
        # the layer's bias gradient = per-channel sums of ``a`` (= d y), from the channels-last copy made above
        # enerf_channel_sums handles C <= 64 with C/4 dividing 256 (capi.hip); any other width takes the torch reduction
        fits = Ca % 4 == 0 and Ca <= 64 and 256 % (Ca // 4) == 0
        gb = self.channel_sums(a_cl, a_cl)[0].float() if fits else a.sum(dim=[0] + list(range(2, a.dim())))
        return gw, gb

    # -- ABI v7: the rest of the training step on the device (train_glue.hip) ----------------------------------------------
    def conv2d_s2k5_dgrad(self, w, dz, add=None):
        """Input gradient of Conv2d(cin -> cout, k5, s2, p2): w (cout,cin,5,5), dz (N,Ho,Wo,cout) -> (N,2Ho,2Wo,cin) (+ add)."""
        cout, cin = w.shape[0], w.shape[1]
        N, Ho, Wo, _ = dz.shape
        gx = torch.empty((N, 2 * Ho, 2 * Wo, cin), dtype=torch.float32, device=dz.device)
        ws = self._scratch(self.dll.enerf_conv2d_s2k5_dgrad_workspace_bytes(cin, cout, N, Ho, Wo), dz.device)
        self._check(self.dll.enerf_conv2d_s2k5_dgrad(_ptr(w), cin, cout, _ptr(dz), _ptr(add), _ptr(gx), N, Ho, Wo, ws.data_ptr(),
                                                     ws.numel(), self.stream_of(dz)), "conv2d_s2k5_dgrad")
        return gx

    def conv2d_s2k5_dgrad_pack(self, w):
        """The packed sub-kernel images of conv2d_s2k5_dgrad alone -> (packed (zero slack), parts, floats per part incl. slack,
        floats per part, output channels per part)."""
        cout, cin = w.shape[0], w.shape[1]
        total = self.dll.enerf_conv2d_s2k5_dgrad_packed_floats(cin, cout)
        cout3 = 2 * cin if 4 * cin > 32 else 4 * cin
        parts = 4 * cin // cout3
        packed = torch.zeros((total,), dtype=torch.float32, device=w.device)
        w3 = torch.empty((4 * cin * cout * 9,), dtype=torch.float32, device=w.device)
        self._check(self.dll.enerf_conv2d_s2k5_dgrad_pack(_ptr(w.contiguous()), cin, cout, _ptr(w3), _ptr(packed), self.stream_of(w)),
                    "conv2d_s2k5_dgrad_pack")
        return packed, parts, total // parts, self.dll.enerf_conv2d_layer_packed_floats(cout, cout3, 3), cout3

    def conv2d_s2k5_dgrad_packed(self, packed, cin, cout, dz, add=None):
        """conv2d_s2k5_dgrad on images prepared by conv2d_s2k5_dgrad_pack (or a PackPlan's gather)."""
        N, Ho, Wo, _ = dz.shape
        gx = torch.empty((N, 2 * Ho, 2 * Wo, cin), dtype=torch.float32, device=dz.device)
        ws = self._scratch(self.dll.enerf_conv2d_s2k5_dgrad_workspace_bytes(cin, cout, N, Ho, Wo), dz.device)
        self._check(self.dll.enerf_conv2d_s2k5_dgrad_packed(_ptr(packed), cin, cout, _ptr(dz), _ptr(add), _ptr(gx), N, Ho, Wo, ws.data_ptr(),
                                                            ws.numel(), self.stream_of(dz)), "conv2d_s2k5_dgrad_packed")
        return gx

    def resize_ac_adjoint(self, g_fine, Hc, Wc, add=None):
        """(..., Hf, Wf) planar gradient maps -> (..., Hc, Wc): adjoint of the align-corners bilinear resize."""
        lead, (Hf, Wf) = g_fine.shape[:-2], g_fine.shape[-2:]
        out = torch.empty(tuple(lead) + (Hc, Wc), dtype=torch.float32, device=g_fine.device)
        self._check(self.dll.enerf_resize_ac_adjoint(_ptr(g_fine), _ptr(add), g_fine.numel() // (Hf * Wf), Hf, Wf, Hc, Wc, _ptr(out),
                                                     self.stream_of(g_fine)), "resize_ac_adjoint")
        return out

    def get_depth_values_bwd(self, prev_depth, prev_std, prev_nf, g_dv, depth_inv):
        """-> (g_depth, g_std) (B,hp,wp): views of one (2,B,hp,wp) buffer."""
        B, D, h, w = g_dv.shape
        hp, wp = prev_depth.shape[-2:]
        g = torch.empty((2, B, hp, wp), dtype=torch.float32, device=g_dv.device)
        scratch = torch.empty((2, B, h, w), dtype=torch.float32, device=g_dv.device)
        self._check(self.dll.enerf_get_depth_values_bwd(_ptr(prev_depth), _ptr(prev_std), _ptr(prev_nf), _ptr(g_dv), B, D, h, w, hp, wp,
                                                        int(depth_inv), _ptr(g), g[1].data_ptr(), _ptr(scratch), self.stream_of(g_dv)),
                    "get_depth_values_bwd")
        return g[0], g[1]

    def ray_samples_fwd(self, rays8, depth, std, nf, Ns, Hr, Wr, depth_inv, want_rays12=False):
        """build_rays + sample_along_depth -> z (B,N,Ns), xyz (B,N,Ns,3), dn (B,N,Ns), uv (B,N,Ns,2)[, rays12 (B,N,12)]."""
        B, N = rays8.shape[:2]
        h, w = depth.shape[-2:]
        E = lambda *sh: torch.empty(sh, dtype=torch.float32, device=rays8.device)
        z, xyz, dn, uv = E(B, N, Ns), E(B, N, Ns, 3), E(B, N, Ns), E(B, N, Ns, 2)
        r12 = E(B, N, 12) if want_rays12 else None
        self._check(self.dll.enerf_ray_samples_fwd(_ptr(rays8), _ptr(depth), _ptr(std), _ptr(nf), B, N, Ns, h, w, Hr, Wr, int(depth_inv),
                                                   _ptr(z), _ptr(xyz), _ptr(dn), _ptr(uv), _ptr(r12), self.stream_of(rays8)),
                    "ray_samples_fwd")
        return (z, xyz, dn, uv, r12) if want_rays12 else (z, xyz, dn, uv)

    def ray_samples_bwd(self, rays8, depth, std, nf, g_xyz, g_dn, Ns, Hr, Wr, depth_inv):
        B, N = rays8.shape[:2]
        h, w = depth.shape[-2:]
        g = torch.empty((2, B, h, w), dtype=torch.float32, device=rays8.device)
        self._check(self.dll.enerf_ray_samples_bwd(_ptr(rays8), _ptr(depth), _ptr(std), _ptr(nf), _ptr(g_xyz), _ptr(g_dn), B, N, Ns, h, w,
                                                   Hr, Wr, int(depth_inv), _ptr(g), g[1].data_ptr(), self.stream_of(rays8)),
                    "ray_samples_bwd")
        return g[0], g[1]

    def camera_tables(self, src_ixts, src_exts, tar_ext, render_scale):
        B, S = src_exts.shape[:2]
        cam = torch.empty((B, S, 16), dtype=torch.float32, device=src_exts.device)
        tcen = torch.empty((B, 4), dtype=torch.float32, device=src_exts.device)
        self._check(self.dll.enerf_camera_tables(_ptr(src_ixts), _ptr(src_exts), _ptr(tar_ext), B, S, float(render_scale), _ptr(cam),
                                                 _ptr(tcen), self.stream_of(src_exts)), "camera_tables")
        return cam, tcen

    def weights_flip_transpose(self, w):
        """w (cout,cin,*k) -> (cin,cout,*k) with every spatial axis reversed: the dgrad weights of a stride-1 convolution."""
        cout, cin = w.shape[:2]
        taps = w.numel() // (cout * cin)
        out = torch.empty((cin, cout) + tuple(w.shape[2:]), dtype=torch.float32, device=w.device)
        self._check(self.dll.enerf_weights_flip_transpose(_ptr(w), cout, cin, taps, _ptr(out), self.stream_of(w)), "weights_flip_transpose")
        return out

    def concat2_pad(self, a, b, n):
        out = torch.empty((n,), dtype=torch.float32, device=a.device)
        self._check(self.dll.enerf_concat2_pad(_ptr(a), a.numel(), _ptr(b), 0 if b is None else b.numel(), n, _ptr(out), self.stream_of(a)),
                    "concat2_pad")
        return out

    def pack_texels_train(self, feat_cl, src_inps, Hr, Wr):
        """feat_cl (n,Hr,Wr,C) channels-last, src_inps (n,3,H,W) -> tex (n,Hr,Wr,C+3)."""
        n, _, _, Cc = feat_cl.shape
        H, W = src_inps.shape[-2:]
        tex = torch.empty((n, Hr, Wr, Cc + 3), dtype=torch.float32, device=feat_cl.device)
        self._check(self.dll.enerf_pack_texels_train(_ptr(feat_cl), Cc, _ptr(src_inps), H, W, Hr, Wr, n, _ptr(tex), self.stream_of(tex)),
                    "pack_texels_train")
        return tex

    def slice_channels(self, src, c0, Cc):
        F = src.shape[-1]
        dst = torch.empty(tuple(src.shape[:-1]) + (Cc,), dtype=torch.float32, device=src.device)
        self._check(self.dll.enerf_slice_channels(_ptr(src), src.numel() // F, F, c0, Cc, _ptr(dst), self.stream_of(src)), "slice_channels")
        return dst

    def concat_channels(self, a, b, Cc):
        """a (..., Ca), b (..., Cb) or None -> (..., Cc) = [a | b | 0]."""
        Ca, Cb = a.shape[-1], 0 if b is None else b.shape[-1]
        out = torch.empty(tuple(a.shape[:-1]) + (Cc,), dtype=torch.float32, device=a.device)
        self._check(self.dll.enerf_concat_channels(_ptr(a), Ca, _ptr(b), Cb, a.numel() // Ca, Cc, _ptr(out), self.stream_of(a)),
                    "concat_channels")
        return out

    def gather_images(self, srcs, which, idx):
        """out[i] = idx[i] >= 0 ? srcs[which[i]].flatten()[idx[i]] : 0; srcs: <= 64 contiguous float tensors; which/idx int32."""
        arr = (C.c_void_p * len(srcs))(*[_ptr(t) for t in srcs])
        out = torch.empty((idx.numel(),), dtype=torch.float32, device=idx.device)
        self._check(self.dll.enerf_gather_images(C.cast(arr, C.c_void_p), len(srcs), which.data_ptr(), idx.data_ptr(), idx.numel(), _ptr(out),
                                                 self.stream_of(out)), "gather_images")
        return out

    def cast_f32(self, x64):
        out = torch.empty(x64.shape, dtype=torch.float32, device=x64.device)
        self._check(self.dll.enerf_cast_f64_f32(x64.data_ptr(), x64.numel(), _ptr(out), self.stream_of(out)), "cast_f64_f32")
        return out

    def reciprocal(self, x):
        out = torch.empty_like(x)
        self._check(self.dll.enerf_reciprocal(_ptr(x), x.numel(), _ptr(out), self.stream_of(x)), "reciprocal")
        return out

    def add(self, a, b):
        out = torch.empty_like(a)
        self._check(self.dll.enerf_add(_ptr(a), _ptr(b), a.numel(), _ptr(out), self.stream_of(a)), "add")
        return out

    def composite(self, raw, z, white_bkgd=False):
        n, Ns = z.shape
        rgb = torch.empty((n, 3), dtype=torch.float32, device=raw.device)
        depth = torch.empty((n,), dtype=torch.float32, device=raw.device)
        weights = torch.empty((n, Ns), dtype=torch.float32, device=raw.device)
        self._check(self.dll.enerf_composite(_ptr(raw), _ptr(z), n, Ns, int(white_bkgd), _ptr(rgb), _ptr(depth), _ptr(weights),
                                             self.stream_of(raw)), "composite")
        return rgb, depth, weights

    def composite_bwd(self, raw, z, g_rgb, g_depth, g_weights):
        n, Ns = z.shape
        g_raw, g_z = torch.empty_like(raw), torch.empty_like(z)
        self._check(self.dll.enerf_composite_bwd(_ptr(raw), _ptr(z), _ptr(g_rgb), _ptr(g_depth), _ptr(g_weights), n, Ns,   # None = zeros
                                                 _ptr(g_raw), _ptr(g_z), self.stream_of(raw)), "composite_bwd")
        return g_raw, g_z

    # -- whole frame / mask compaction -------------------------------------------------------------
    def mask_compact(self, mask: torch.Tensor, workspace=None):
        """network_human.py:90-93 on device: (index int32 (n), count int32 (1)) of the non-zero mask elements."""
        if not mask.is_contiguous():
            raise EnerfError("mask must be contiguous")
        n = mask.numel()
        dev = mask.device
        need = self.dll.enerf_mask_compact_workspace_bytes(n)
        if workspace is None or workspace.numel() * 4 < need:
            workspace = torch.empty(((need + 3) // 4,), dtype=torch.int32, device=dev)
        index = torch.empty((n,), dtype=torch.int32, device=dev)
        count = torch.empty((1,), dtype=torch.int32, device=dev)
        self._check(self.dll.enerf_mask_compact(mask.data_ptr(), mask.element_size(), n, index.data_ptr(), count.data_ptr(),
                                                workspace.data_ptr(), workspace.numel() * 4, self.stream_of(mask)),
                    "mask_compact")
        return index, count

    def forward_workspace_bytes(self, args: "FrameArgs") -> int:
        n = self.dll.enerf_forward_workspace_bytes(C.byref(args))
        if n == 0:
            raise EnerfError(f"forward: {self.dll.enerf_last_error().decode()}")
        return n

    def forward(self, args: "FrameArgs", stream):
        self._check(self.dll.enerf_forward(C.byref(args), stream), "forward")

    # -- the steps before / after the path ---------------------------------
    def gen_rays(self, tar_ext, tar_ixt, Hr, Wr, scale):
        """lib/datasets/enerf_utils.py:61-71 (full image) on device -> (B, Hr*Wr, 8)."""
        B = tar_ext.shape[0]
        rays = torch.empty((B, Hr * Wr, 8), dtype=torch.float32, device=tar_ext.device)
        self._check(self.dll.enerf_gen_rays(_ptr(tar_ext), _ptr(tar_ixt), B, Hr, Wr, float(scale), _ptr(rays),
                                            self.stream_of(rays)), "gen_rays")
        return rays

    def pack_rgb8(self, rgb, H, W, flip=True):
        """gui_human.py:88-91: (H*W,3) float in [0,1] -> (H,W,3) uint8, vertically flipped for GL."""
        out = torch.empty((H, W, 3), dtype=torch.uint8, device=rgb.device)
        self._check(self.dll.enerf_pack_rgb8(_ptr(rgb), H, W, int(flip), out.data_ptr(), self.stream_of(rgb)),
                    "pack_rgb8")
        return out

    def gen_rays_at(self, tar_ext, tar_ixt, xy, scale):
        """lib/datasets/enerf_utils.py:45-56 (training branch) for the pixel list ``xy`` (B,N,2) int32 -> (B,N,8)."""
        B, N = xy.shape[:2]
        if xy.dtype != torch.int32 or not xy.is_contiguous():
            raise EnerfError("xy must be a contiguous int32 tensor (B,N,2)")
        rays = torch.empty((B, N, 8), dtype=torch.float32, device=tar_ext.device)
        self._check(self.dll.enerf_gen_rays_at(_ptr(tar_ext), _ptr(tar_ixt), xy.data_ptr(), B, N, float(scale), _ptr(rays),
                                               self.stream_of(rays)), "gen_rays_at")
        return rays

    def rays_bbox_mask(self, rays, bounds):
        """lib/utils/net_utils.py:13-28 gen_rays_bbox: rays (N,8), bounds (2,3) -> int32 mask (N)."""
        n = rays.shape[0]
        mask = torch.empty((n,), dtype=torch.int32, device=rays.device)
        self._check(self.dll.enerf_rays_bbox_mask(_ptr(rays), _ptr(bounds), n, mask.data_ptr(), self.stream_of(rays)),
                    "rays_bbox_mask")
        return mask

    def select_views(self, cam_points, c2w, k):
        """zjumocap/enerf_interactive.py:207-210: indices (k) int32 of the cameras nearest to the target camera."""
        idx = torch.empty((k,), dtype=torch.int32, device=cam_points.device)
        self._check(self.dll.enerf_select_views(_ptr(cam_points), cam_points.shape[0], _ptr(c2w), k, idx.data_ptr(),
                                                self.stream_of(cam_points)), "select_views")
        return idx

    def gather_views(self, inps, exts, ixts, idx):
        """:214-217: inps (V,H,W,3), exts (V,4,4), ixts (V,3,3) + idx (k) -> src_inps (k,3,H,W), src_exts, src_ixts."""
        V, H, W, _ = inps.shape
        k, dev = idx.numel(), inps.device
        si = torch.empty((k, 3, H, W), dtype=torch.float32, device=dev)
        se = torch.empty((k, 4, 4), dtype=torch.float32, device=dev)
        sk = torch.empty((k, 3, 3), dtype=torch.float32, device=dev)
        self._check(self.dll.enerf_gather_views(_ptr(inps), _ptr(exts), _ptr(ixts), idx.data_ptr(), k, H, W, _ptr(si),
                                                _ptr(se), _ptr(sk), self.stream_of(inps)), "gather_views")
        return si, se, sk

    def eval_stats(self, pred_rgb, gt_rgb, mask=None, pred_depth=None, gt_depth=None, image_hw=None, crop=(0, 0),
                   sync=True):
        """evaluators/enerf.py:45-71,88-103 on device.  Returns dict(psnr[, abs, acc_2, acc_10]) after ONE 48-byte D2H
        copy — or, with ``sync=False``, the 6-double device accumulator (see include/enerf_hip.h)."""
        acc = torch.empty((6,), dtype=torch.float64, device=pred_rgb.device)
        n_rgb = pred_rgb.numel() // 3
        mb = 0
        if mask is not None:
            if mask.dtype not in (torch.int32, torch.uint8, torch.bool) or not mask.is_contiguous():
                raise EnerfError("mask must be a contiguous int32 / uint8 / bool tensor")
            mb = mask.element_size()
        n_d = 0 if pred_depth is None else pred_depth.numel()
        h, w = image_hw if image_hw is not None else (0, 0)
        self._check(self.dll.enerf_eval_stats(_ptr(pred_rgb), _ptr(gt_rgb), None if mask is None else mask.data_ptr(), mb,
                                              n_rgb, int(w), int(h), int(crop[0]), int(crop[1]), _ptr(pred_depth),
                                              _ptr(gt_depth), n_d, acc.data_ptr(), self.stream_of(pred_rgb)), "eval_stats")
        if not sync:
            return acc
        return stats_from_acc(acc.cpu().tolist())


    def depth_stats(self, pred_depth, gt_depth, sync=True):
        """The depth statistics alone (evaluators/enerf.py:96-103: abs / acc_2 / acc_10 over gt != 0): enerf_eval_stats with
        no rgb.  Returns dict(abs, acc_2, acc_10) — empty when no pixel has ground truth."""
        acc = torch.empty((6,), dtype=torch.float64, device=pred_depth.device)
        self._check(self.dll.enerf_eval_stats(None, None, None, 0, 0, 0, 0, 0, 0, _ptr(pred_depth), _ptr(gt_depth),
                                              pred_depth.numel(), acc.data_ptr(), self.stream_of(pred_depth)), "eval_stats")
        if not sync:
            return acc
        a = acc.cpu().tolist()
        return dict(abs=a[2] / a[3], acc_2=a[4] / a[3], acc_10=a[5] / a[3]) if a[3] > 0 else {}


def stats_from_acc(a) -> dict:
    import math
    out = {"psnr": 10.0 * math.log10(a[1] / a[0]) if a[0] > 0 else float("inf")}
    if a[3] > 0:
        out.update(abs=a[2] / a[3], acc_2=a[4] / a[3], acc_10=a[5] / a[3])
    return out


_LIB: Optional[EnerfLib] = None


def get_lib() -> EnerfLib:
    """The product library (HIP, gfx950).  Raises if it has not been built — never falls back."""
    global _LIB
    if _LIB is None:
        _LIB = EnerfLib(LIB_PATH)
    return _LIB
