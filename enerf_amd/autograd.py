"""torch.autograd.Functions over the HIP forward + backward kernels of the training path.

Each Function's forward is the SAME kernel the inference path uses (or its training-mode sibling) and its backward is a
hand-written HIP kernel (enerf_amd/csrc/backward.hip) — no torch ops in between, raw pointers through the C ABI.
``enerf_amd/train_path.py`` routes a stage through here when the library is available for the tensors' device (the GPU
build, or the CPU lane-emulator build in tests); otherwise the stage runs as the equivalent torch ops.  Built so far:
the cost-volume warp + variance, depth regression, alpha compositing, and the WEIGHT gradient of every convolution
(the library GEMM MIOpen picks for it took 89 % of a training step).  Not yet: render MLP, conv dgrad / BN-train.
"""
from __future__ import annotations

import torch

from .lib import EnerfLib


def _c(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous()


class FeatureVolumeFn(torch.autograd.Function):
    """homo_warp x S + variance (utils.py:57-95, 322-349).  feats (B,S,C,Hs,Ws) NCHW, proj (B,S,3,4), dv (B,D,h,w)
    -> cost volume (B,C,D,h,w).  Gradients: feats (scatter-add of the bilinear taps) and dv (through the warp grid)."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, feats, proj, dv):
        B, S, C, Hs, Ws = feats.shape
        feat_cl = _c(feats.permute(0, 1, 3, 4, 2))                      # channels-last, what the kernels read
        proj, dv = _c(proj.detach()), _c(dv)
        vol = lib.build_feature_volume(feat_cl, proj, dv, C)              # (B,D,h,w,C)
        ctx.lib = lib
        ctx.save_for_backward(feat_cl, proj, dv)
        return vol.permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, g_vol):
        feat_cl, proj, dv = ctx.saved_tensors
        g_cl = _c(g_vol.permute(0, 2, 3, 4, 1))
        g_feat, g_dv = ctx.lib.build_feature_volume_bwd(feat_cl, proj, dv, g_cl)
        return None, g_feat.permute(0, 1, 4, 2, 3), None, g_dv


class DepthRegressionFn(torch.autograd.Function):
    """depth_regression (utils.py:658-667): prob, dv (B,D,h,w) -> depth, std (B,h,w)."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, prob, dv, depth_inv: bool):
        prob, dv = _c(prob), _c(dv)
        depth, std = lib.depth_regression(prob, dv, depth_inv)
        ctx.lib, ctx.depth_inv = lib, depth_inv
        ctx.save_for_backward(prob, dv)
        return depth, std

    @staticmethod
    def backward(ctx, g_depth, g_std):
        prob, dv = ctx.saved_tensors
        g_prob, g_dv = ctx.lib.depth_regression_bwd(prob, dv, _c(g_depth), _c(g_std), ctx.depth_inv)
        return None, g_prob, g_dv, None


class CompositeFn(torch.autograd.Function):
    """raw2outputs (utils.py:571-603): raw (B,N,Ns,4), z (B,N,Ns) -> rgb (B,N,3), depth (B,N), weights (B,N,Ns)."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, raw, z, white_bkgd: bool):
        B, N, Ns = z.shape
        raw2, z2 = _c(raw).reshape(B * N, Ns, 4), _c(z).reshape(B * N, Ns)
        rgb, depth, weights = lib.composite(raw2, z2, white_bkgd)
        ctx.lib, ctx.shape = lib, (B, N, Ns)
        ctx.save_for_backward(raw2, z2)
        return rgb.view(B, N, 3), depth.view(B, N), weights.view(B, N, Ns)

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_weights):
        raw2, z2 = ctx.saved_tensors
        B, N, Ns = ctx.shape
        g_raw, g_z = ctx.lib.composite_bwd(raw2, z2, _c(g_rgb).reshape(B * N, 3), _c(g_depth).reshape(B * N),
                                           _c(g_weights).reshape(B * N, Ns))
        return None, g_raw.view(B, N, Ns, 4), g_z.view(B, N, Ns), None


class ConvFn(torch.autograd.Function):
    """A bias-free Conv2d / Conv3d / ConvTranspose3d(k3,s2,p1,op1) whose WEIGHT gradient runs on the matrix cores
    (enerf_conv_wgrad).  Forward and the input gradient stay on the library convolution (MIOpen) for now."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, x, w, stride: int, padding: int, transposed: bool):
        nd = x.dim() - 2
        if transposed:
            y = torch.nn.functional.conv_transpose3d(x, w, None, stride=stride, padding=padding, output_padding=stride - 1)
        elif nd == 3:
            y = torch.nn.functional.conv3d(x, w, None, stride, padding)
        else:
            y = torch.nn.functional.conv2d(x, w, None, stride, padding)
        ctx.lib, ctx.cfg = lib, (stride, padding, transposed, nd)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, transposed, nd = ctx.cfg
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[1]:
            if transposed:                              # dgrad of a transposed conv = the plain strided conv
                gx = torch.nn.functional.conv3d(gy, w, None, stride, padding)
            elif nd == 3:
                gx = torch.nn.grad.conv3d_input(x.shape, w, gy, stride, padding)
            else:
                gx = torch.nn.grad.conv2d_input(x.shape, w, gy, stride, padding)
        if ctx.needs_input_grad[2]:
            k, pad = tuple(w.shape[2:]), (padding,) * nd
            gw = ctx.lib.conv_wgrad(x, gy, k, stride, pad) if transposed else ctx.lib.conv_wgrad(gy, x, k, stride, pad)
        return None, gx, gw, None, None, None


def conv_module(lib, m, x):
    """Apply an nn.Conv2d / nn.Conv3d / nn.ConvTranspose3d of the network through ConvFn (bias added outside)."""
    transposed = isinstance(m, torch.nn.ConvTranspose3d)
    y = ConvFn.apply(lib, x, m.weight, int(m.stride[0]), int(m.padding[0]), transposed)
    if m.bias is not None:
        y = y + m.bias.view(1, -1, *([1] * (x.dim() - 2)))
    return y
