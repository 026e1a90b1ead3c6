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

from . import pack_plan
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
        ctx.set_materialize_grads(False)                   # outputs the loss ignores arrive as None, not as zero tensors
        return rgb.view(B, N, 3), depth.view(B, N), weights.view(B, N, Ns)

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_weights):
        raw2, z2 = ctx.saved_tensors
        B, N, Ns = ctx.shape
        opt = lambda g, *sh: None if g is None else _c(g).reshape(*sh)
        g_raw, g_z = ctx.lib.composite_bwd(raw2, z2, opt(g_rgb, B * N, 3), opt(g_depth, B * N), opt(g_weights, B * N, Ns))
        # utils.py:595: depth_map = sum(weights * z_vals.detach()) — the sample depths never receive a gradient from the
        # composited depth (k_composite_bwd's g_z is the un-detached derivative: dropped here)
        return None, g_raw.view(B, N, Ns, 4), None, None


def gather_cameras(batch, render_scale: float, lib):
    """The per-view constants of enerf_gather_*: cam (B,S,16) = K'E33 | K't | source centre | 0 and tcen (B,4), with
    K' = K scaled to the level (utils.py:697-704); products in fp64, stored fp32 (as the inference kernel's table).
    ONE launch on the device (enerf_camera_tables; no host synchronisation, capturable).  The torch-op twin the kernel is
    tested against is tests/torch_twins.py::gather_cameras."""
    if lib is None:
        raise RuntimeError("gather_cameras: the HIP library is required")
    return lib.camera_tables(_c(batch["src_ixts"]), _c(batch["src_exts"]), _c(batch["tar_ext"]), render_scale)


class GatherFn(torch.autograd.Function):
    """get_img_feat + get_vox_feat (utils.py:689-722, 456-458) on the HIP kernels: xyz (B,P,3), dn (B,P), uv (B,P,2),
    tex_cl (B,S,Hr,Wr,F) and vol_cl (B,D,h,w,8) CHANNELS-LAST -> x (B,P,S,F+4), vox (B,P,8).  Differentiable in xyz, dn,
    tex_cl, vol_cl (gradients in the same layouts)."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, xyz, dn, uv, tex_cl, vol_cl, cam, tcen, n_samples=0, ray_w=0):
        xyz, dn, uv, tex_cl, vol_cl = _c(xyz), _c(dn), _c(uv), _c(tex_cl), _c(vol_cl)
        x, vox = lib.gather_fwd(xyz, dn, uv, tex_cl, vol_cl, cam, tcen)
        ctx.lib, ctx.hints = lib, (int(n_samples), int(ray_w))
        ctx.save_for_backward(xyz, dn, uv, tex_cl, vol_cl, cam, tcen)
        return x, vox

    @staticmethod
    def backward(ctx, g_x, g_vox):
        xyz, dn, uv, tex_cl, vol_cl, cam, tcen = ctx.saved_tensors
        g_tex, g_vol, g_xyz, g_dn = ctx.lib.gather_bwd(xyz, dn, uv, tex_cl, vol_cl, cam, tcen, _c(g_x), _c(g_vox), *ctx.hints)
        return None, g_xyz, g_dn, None, g_tex, g_vol, None, None, None, None


class DepthValuesFn(torch.autograd.Function):
    """get_depth_values of a cascade level > 0 (utils.py:112-151): the previous level's depth, std (B,hp,wp; disparity space)
    and near_far (B,2,hp,wp; detached) -> depth_values (B,D,h,w), near_far (B,2,h,w; detached, utils.py:148).  Forward = the
    inference kernel; backward through the clamps, the reciprocals and the align-corners upsampling (enerf_get_depth_values_bwd)."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, depth, std, near_far, batch_near_far, D: int, h: int, w: int, depth_inv: bool):
        depth, std, near_far = _c(depth), _c(std), _c(near_far)
        dv, nf = lib.get_depth_values(batch_near_far, (depth, std, near_far), depth.shape[0], D, h, w, depth_inv)
        ctx.lib, ctx.depth_inv = lib, depth_inv
        ctx.save_for_backward(depth, std, near_far)
        ctx.mark_non_differentiable(nf)
        ctx.set_materialize_grads(False)
        return dv, nf

    @staticmethod
    def backward(ctx, g_dv, _g_nf):
        depth, std, near_far = ctx.saved_tensors
        if g_dv is None:
            return (None,) * 9
        g_d, g_s = ctx.lib.get_depth_values_bwd(depth, std, near_far, _c(g_dv), ctx.depth_inv)
        return None, g_d, g_s, None, None, None, None, None, None


class ReciprocalFn(torch.autograd.Function):
    """1 / x (depth_mvs of a disparity-space level, network.py:105-108) as one library launch.  The trainer's loss never reads
    depth_mvs; if something does, the backward is the two-op torch expression."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, x):
        y = lib.reciprocal(_c(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return None, -g * y * y


class RaySamplesFn(torch.autograd.Function):
    """build_rays + sample_along_depth (utils.py:390-441): depth, std (B,h,w), near_far (B,2,h,w; detached), rays (B,N,8)
    -> z (B,N,Ns), xyz (B,N,Ns,3), dn (B,N,Ns), uv (B,N,Ns,2).  Differentiable in depth and std (through xyz and dn; z only
    feeds the composited depth, which detaches it: utils.py:595)."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, depth, std, near_far, rays8, Ns: int, Hr: int, Wr: int, depth_inv: bool):
        depth, std, near_far, rays8 = _c(depth), _c(std), _c(near_far), _c(rays8[..., :8])
        z, xyz, dn, uv = lib.ray_samples_fwd(rays8, depth, std, near_far, Ns, Hr, Wr, depth_inv)
        ctx.lib, ctx.cfg = lib, (Ns, Hr, Wr, depth_inv)
        ctx.save_for_backward(depth, std, near_far, rays8)
        ctx.mark_non_differentiable(z, uv)
        ctx.set_materialize_grads(False)
        return z, xyz, dn, uv

    @staticmethod
    def backward(ctx, _g_z, g_xyz, g_dn, _g_uv):
        depth, std, near_far, rays8 = ctx.saved_tensors
        Ns, Hr, Wr, depth_inv = ctx.cfg
        if g_xyz is None and g_dn is None:
            return (None,) * 9
        B, N = rays8.shape[:2]
        g_xyz = g_xyz if g_xyz is not None else rays8.new_zeros(B, N, Ns, 3)
        g_dn = g_dn if g_dn is not None else rays8.new_zeros(B, N, Ns)
        g_d, g_s = ctx.lib.ray_samples_bwd(rays8, depth, std, near_far, _c(g_xyz), _c(g_dn), Ns, Hr, Wr, depth_inv)
        return None, g_d, g_s, None, None, None, None, None, None


class TexelsFn(torch.autograd.Function):
    """The render-side gather source (network.py:28-33): im_feat (B,S,C,Hr,Wr) — an NCHW view of the FeatureNet's channels-last
    map, already at the render resolution — and src_inps (B,S,3,H,W) -> tex_cl (B,S,Hr,Wr,C+3) = [features | unpreprocessed,
    resized colours] channels-last.  Gradient: the feature channels of d tex (the images are inputs)."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, im_feat, src_inps, Hr: int, Wr: int):
        B, S, Cc = im_feat.shape[:3]
        feat_cl = _c(im_feat.permute(0, 1, 3, 4, 2)).view(B * S, Hr, Wr, Cc)
        H, W = src_inps.shape[-2:]
        tex = lib.pack_texels_train(feat_cl, _c(src_inps).view(B * S, 3, H, W), Hr, Wr)
        ctx.lib, ctx.C = lib, Cc
        return tex.view(B, S, Hr, Wr, Cc + 3)

    @staticmethod
    def backward(ctx, g_tex):
        g = ctx.lib.slice_channels(_c(g_tex), 0, ctx.C)                    # (B,S,Hr,Wr,C)
        return None, g.permute(0, 1, 4, 2, 3), None, None, None


class ConvFn(torch.autograd.Function):
    """A Conv2d / Conv3d / ConvTranspose3d(k3,s2,p1,op1) whose WEIGHT and BIAS gradients come from the HIP kernels
    (enerf_conv_wgrad on the matrix cores; enerf_channel_sums).  Forward and the input gradient stay on the library
    convolution (MIOpen) for now."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, x, w, bias, stride: int, padding: int, transposed: bool):
        nd = x.dim() - 2
        if transposed:
            y = torch.nn.functional.conv_transpose3d(x, w, bias, stride=stride, padding=padding, output_padding=stride - 1)
        elif nd == 3:
            y = torch.nn.functional.conv3d(x, w, bias, stride, padding)
        else:
            y = torch.nn.functional.conv2d(x, w, bias, stride, padding)
        ctx.lib, ctx.cfg = lib, (stride, padding, transposed, nd)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, transposed, nd = ctx.cfg
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[1]:
            if transposed:                              # dgrad of a transposed conv = the plain strided conv
                gx = torch.nn.functional.conv3d(gy, w, None, stride, padding)
            elif nd == 3:
                gx = torch.nn.grad.conv3d_input(x.shape, w, gy, stride, padding)
            else:
                gx = torch.nn.grad.conv2d_input(x.shape, w, gy, stride, padding)
        want_b = ctx.needs_input_grad[3]
        if ctx.needs_input_grad[2] or want_b:
            k, pad = tuple(w.shape[2:]), (padding,) * nd
            if transposed:
                gw = ctx.lib.conv_wgrad(x, gy, k, stride, pad)
                if want_b:
                    gb = gy.sum(dim=[0] + list(range(2, gy.dim())))
            else:
                gw, gb = ctx.lib.conv_wgrad(gy, x, k, stride, pad, bias=True) if want_b else (ctx.lib.conv_wgrad(gy, x, k, stride, pad), None)
        return None, gx, gw, gb, None, None, None


def conv_module(lib, m, x):
    """Apply an nn.Conv2d / nn.Conv3d / nn.ConvTranspose3d of the network through ConvFn."""
    transposed = isinstance(m, torch.nn.ConvTranspose3d)
    return ConvFn.apply(lib, x, m.weight, m.bias, int(m.stride[0]), int(m.padding[0]), transposed)


# ---------------------------------------------------------------------------------------------------------------------
# MinCostRegNet / CostRegNet (cost_reg_net.py:4-86) in TRAINING mode, forward and backward on the HIP kernels:
# convolutions and their input gradients on the inference path's MFMA kernels (enerf_conv3d_layer), weight gradients on
# the matrix cores (enerf_conv_wgrad), BatchNorm batch statistics / normalise+ReLU+skip / backward on the two channel
# kernels of train.hip.  Everything between the cost volume and (feat, prob) stays channels-last on the device; only
# C-sized vectors (means, scales, d gamma, d beta) come from one coefficient kernel per BatchNorm and direction.
# ---------------------------------------------------------------------------------------------------------------------
_S1, _S2, _T2 = 0, 1, 2


SYNC_SINGLE_RANK = False     # tests: run the statistics exchange on a 1-rank group too (captures RCCL nodes on a 1-GPU box)


class _BatchNormTrain:
    """BatchNorm (+ ReLU) (+ skip add) in training mode on a channels-last tensor z (..., C): batch statistics, running
    statistics update, normalise — and the backward — on the channel kernels of train.hip: per direction one statistics
    launch, one C-sized coefficient launch (enerf_bn_train[_bwd]_coeffs, fp64) and one affine launch; under SyncBatchNorm
    (trainer.py:16) one small all-reduce of the statistics (+ the position count, which stays on the device) in between.
    Shared by the 3-D blocks of the cost-volume networks and the 2-D blocks of the FeatureNet."""

    def __init__(self, lib, bn, relu):
        self.lib, self.bn, self.relu = lib, bn, relu

    @staticmethod
    def _group(bn):
        """The module's own process group (``SyncBatchNorm(process_group=...)`` / ``convert_sync_batchnorm(m, group)``);
        None = the default group — what torch's SyncBatchNorm uses as well."""
        return getattr(bn, "process_group", None)

    @classmethod
    def _synced(cls, bn):
        import torch.distributed as dist
        return isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized() and \
            (dist.get_world_size(cls._group(bn)) > 1 or SYNC_SINGLE_RANK)

    def forward(self, z, residual=None):
        lib, bn = self.lib, self.bn
        C_ = z.shape[-1]
        self.fused = not self._synced(bn) and lib.bn_stats_fit(z)
        if self.fused:       # no statistics exchange: partial sums, then the affine pass with their reduction + the coefficients in its
            y, self.mean_invstd, self.ss, n = lib.bn_train_apply(z, bn, residual, self.relu)      # prologue (ABI v11; large layers: three launches)
            self.z, self.n, self.scale, self.shift = z, n, self.ss[0], self.ss[1]
            return y
        sums, n = lib.channel_sums_raw(z, z), z.numel() // C_
        if self._synced(bn):
            import torch.distributed as dist
            buf = torch.cat([sums.view(-1), sums.new_full((1,), float(n))])          # fill kernel, not a host copy: capture-safe
            dist.all_reduce(buf, group=self._group(bn))
            sums, n = buf[:2 * C_].view(2, C_), buf[2 * C_:]                         # the global count stays on the device
        self.mean_invstd, ss = lib.bn_train_coeffs(sums, n, bn)
        self.z, self.n, self.scale, self.shift = z, n, ss[0], ss[1]
        return lib.channel_affine(z, self.scale, self.shift, residual=residual, relu=self.relu)

    def backward(self, g):
        """g = gradient w.r.t. the normalised (+ ReLU) output -> (d z, d gamma, d beta)."""
        lib, bn, z = self.lib, self.bn, self.z
        mask = dict(z_mask=z, mask_scale=self.scale, mask_shift=self.shift) if self.relu else {}
        if self.fused:
            dz, dgb = lib.bn_train_bwd_apply(g, z, self.mean_invstd, self.ss, self.relu)
            return dz, dgb[0], dgb[1]
        local = lib.channel_sums_raw(g, z, **mask)                                   # sum gm, sum gm*z over this rank
        glob = local
        if self._synced(bn):                                                         # the input gradient needs the global sums;
            import torch.distributed as dist                                         # d gamma / d beta stay LOCAL (DDP averages
            glob = local.clone()                                                     # them), like torch's SyncBatchNorm
            dist.all_reduce(glob, group=self._group(bn))
        dgb, k23 = lib.bn_train_bwd_coeffs(local, glob, self.n, self.mean_invstd, self.scale)
        dz = lib.channel_affine(g, self.scale, k23[1], b=z, q=k23[0], **mask)
        return dz, dgb[0], dgb[1]


class _Block:
    """One Conv3d/ConvTranspose3d + BatchNorm3d (+ ReLU) (+ skip add) in training mode.  ``images``: the packed weight images
    of the forward layer and of its input-gradient layer (pack_plan.cost_reg_plan: one gather launch per network and step)."""

    def __init__(self, lib, w, bn, kind, relu, images):
        self.lib, self.w, self.kind = lib, w, kind
        self.norm = _BatchNormTrain(lib, bn, relu)
        self.pk_fwd, self.pk_bwd = images
        if kind == _T2:
            self.cin, self.cout = w.shape[0], w.shape[1]
        else:
            self.cout, self.cin = w.shape[0], w.shape[1]

    def forward(self, x, residual=None):
        lib = self.lib
        self.x = x
        z = lib.conv3d_layer(self.pk_fwd, self.cin, self.cout, self.kind, x)
        return self.norm.forward(z, residual)

    def backward(self, g, add=None):
        """g = gradient w.r.t. the block's output (the skip branch's share is the same tensor); ``add``: a tensor of the input's
        shape added to the input gradient in the convolution's epilogue (the skip branch's gradient of the block's input).
        Returns (grad_input (+ add), grad_weight, grad_bn_weight, grad_bn_bias)."""
        lib = self.lib
        dz, dgamma, dbeta = self.norm.backward(g)
        if self.kind == _S1:                                                         # dgrad: flipped, channel-transposed weights
            gx = lib.conv3d_layer(self.pk_bwd, self.cout, self.cin, _S1, dz, residual=add)
            gw = lib.conv_wgrad_cl(dz, self.x, 1)
        elif self.kind == _S2:                                                       # dgrad of a stride-2 conv = transposed conv on w
            gx = lib.conv3d_layer(self.pk_bwd, self.cout, self.cin, _T2, dz, residual=add)
            gw = lib.conv_wgrad_cl(dz, self.x, 2)
        else:                                                                        # dgrad of a transposed conv = stride-2 conv on w
            gx = lib.conv3d_layer(self.pk_bwd, self.cout, self.cin, _S2, dz, residual=add)
            gw = lib.conv_wgrad_cl(self.x, dz, 2)
        return gx, gw, dgamma, dbeta


class CostRegTrainFn(torch.autograd.Function):
    """vol (B,C,D,h,w) -> feat (B,8,D,h,w), prob (B,D,h,w) through ``CostRegParams`` ``m`` in training mode."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, m, vol, *params):
        x = vol.permute(0, 2, 3, 4, 1).contiguous()                                   # channels-last (a view of FeatureVolumeFn's output)
        blk = {}
        img = pack_plan.plan_of(lib, m, pack_plan.cost_reg_plan, x.device).run()     # every weight image of the step: one launch

        def cbr(i, kind):
            mod = getattr(m, f"conv{i}")
            blk[i] = _Block(lib, mod.conv.weight, mod.bn, kind, True, (img[i, "fwd"], img[i, "bwd"]))
            return blk[i]

        def up(i):
            mod = getattr(m, f"conv{i}")
            blk[i] = _Block(lib, mod[0].weight, mod[1], _T2, False, (img[i, "fwd"], img[i, "bwd"]))
            return blk[i]
        c0 = cbr(0, _S1).forward(x)
        c2 = cbr(2, _S1).forward(cbr(1, _S2).forward(c0))
        c4 = cbr(4, _S1).forward(cbr(3, _S2).forward(c2))
        y = c4
        if m.full:
            c6 = cbr(6, _S1).forward(cbr(5, _S2).forward(c4))
            y = up(7).forward(c6, residual=c4)
        y = up(9).forward(y, residual=c2)
        y = up(11).forward(y, residual=c0)
        # heads: feat_conv (8 -> 8) ++ depth_conv (8 -> 1) as one 8 -> 16 layer (rows 9..15 zero)
        heads = lib.conv3d_layer(img["heads", "fwd"], 8, 16, _S1, y)
        ctx.lib, ctx.m, ctx.blk, ctx.y, ctx.heads_bwd = lib, m, blk, y, img["heads", "bwd"]
        feat = lib.slice_channels(heads, 0, 8).permute(0, 4, 1, 2, 3)      # contiguous channels-last (what the render-side fetch reads)
        prob = lib.slice_channels(heads, 8, 1)[..., 0]                     # contiguous (B,D,h,w)
        return feat, prob

    @staticmethod
    def backward(ctx, g_feat, g_prob):
        lib, m, blk, y = ctx.lib, ctx.m, ctx.blk, ctx.y
        B, D, h, w, _ = y.shape
        g16 = lib.concat_channels(_c(g_feat.permute(0, 2, 3, 4, 1)), _c(g_prob).unsqueeze(-1), 16)
        with lib.wgrad_reduce_batch(g16):            # the eleven weight gradients' second stages: one launch at the end of the pass
            return CostRegTrainFn._backward(ctx, g16)

    @staticmethod
    def _backward(ctx, g16):
        lib, m, blk, y = ctx.lib, ctx.m, ctx.blk, ctx.y
        g = lib.conv3d_layer(ctx.heads_bwd, 16, 8, _S1, g16)                         # d y11
        gw16 = lib.conv_wgrad_cl(g16, y, 1)
        grads = {"feat": gw16[:8], "depth": gw16[8:9]}

        def back(i, gi, add=None):
            gx, gw, dg, db = blk[i].backward(gi, add)
            grads[i] = (gw, dg, db)
            return gx
        g_c0 = g
        g = back(11, g)                                           # -> d y9 ; skip share of y11 goes to c0
        g_c2 = g
        g = back(9, g)                                            # -> d y7
        # a skip tensor's gradient = the skip share + what comes back through the stride-2 block below it: the sum rides in the
        # epilogue of that block's input-gradient convolution (residual operand), not in a launch of its own
        g_c4 = back(5, back(6, back(7, g)), add=g) if m.full else g
        g_c2 = back(3, back(4, g_c4), add=g_c2)
        g_c0 = back(1, back(2, g_c2), add=g_c0)
        g_x = back(0, g_c0)
        out = []
        for i in ctx.order:
            if i == "feat":
                out.append(grads["feat"])
            elif i == "depth":
                out.append(grads["depth"])
            else:
                out.extend(grads[i])
        return (None, None, g_x.permute(0, 4, 1, 2, 3)) + tuple(out)


def cost_reg_train(lib, m, vol):
    """Apply CostRegTrainFn with the module's parameters as differentiable inputs (fixed order)."""
    order = [0, 1, 2, 3, 4] + ([5, 6, 7] if m.full else []) + [9, 11, "feat", "depth"]
    params = []
    for i in order:
        if i == "feat":
            params.append(m.feat_conv[0].weight)
        elif i == "depth":
            params.append(m.depth_conv[0].weight)
        else:
            mod = getattr(m, f"conv{i}")
            conv, bn = (mod[0], mod[1]) if i in (7, 9, 11) else (mod.conv, mod.bn)
            params += [conv.weight, bn.weight, bn.bias]

    class _Fn(CostRegTrainFn):
        @staticmethod
        def forward(ctx, *a):
            ctx.order = order
            return CostRegTrainFn.forward(ctx, *a)
    return _Fn.apply(lib, m, vol, *params)


# ---------------------------------------------------------------------------------------------------------------------
# FeatureNet (feature_net.py:4-36) in TRAINING mode, forward and backward on the HIP kernels: every convolution on the
# inference path's MFMA kernel with an identity epilogue (enerf_conv2d_layer; the top-down `up2 + lateral` adds ride in the
# lateral convolution's epilogue exactly as in inference), the input gradients of the stride-1 layers on the same kernel
# (flipped, channel-transposed weights), BatchNorm2d on the channel kernels shared with the cost-volume networks, weight
# gradients on the matrix cores, the adjoint of the 2x upsampling as a gather kernel.  Everything stays channels-last
# between the image and the three output maps.  Still torch ops: the input gradients of the two stride-2 5x5 layers (a
# transposed 5x5 convolution: no kernel of ours).
# ---------------------------------------------------------------------------------------------------------------------
class _Conv2d:
    """One convolution of the FeatureNet on channels-last tensors (cin = 3: the NCHW image batch).  ``img``: the step's packed
    weight images (pack_plan.feature_net_plan), ``name`` this layer's key in it."""

    def __init__(self, lib, conv, img, name):
        self.lib, self.conv, self.img, self.name = lib, conv, img, name
        self.cout, self.cin, self.k, _ = conv.weight.shape
        self.stride = int(conv.stride[0])

    def forward(self, x, up=None):
        self.x = x
        return self.lib.conv2d_layer(self.img[self.name, "fwd"], self.cin, self.cout, self.k, self.stride, x, up)

    def backward(self, dz, need_input=True, add=None):
        """dz = gradient w.r.t. the convolution's output (channels-last) -> (grad_input (+ add) or None, grad_weight, grad_bias or None)."""
        lib = self.lib
        if self.cin == 3:                                       # the NCHW image batch -> channels-last (the library's adapter kernel)
            N, _, H, W = self.x.shape
            x_cl = lib.channels_last(self.x.reshape(N, 3, H * W), N, 3, H * W).view(N, H, W, 3)
        else:
            x_cl = self.x
        if self.k == 1 and self.stride == 1 and self.conv.bias is not None:
            # toplayer / lat1 / lat0: a 1x1 layer is a GEMM over the positions, and its bias gradient is one more column of the same
            # pass (enerf_gemm_wgrad's virtual all-ones column) instead of three launches (channel sums, their finish, the fp32 cast)
            gw, gb = lib.gemm_wgrad(dz.reshape(-1, self.cout), x_cl.reshape(-1, self.cin), bias=True)
            gw = gw.view(self.cout, self.cin, 1, 1)
        else:
            gw = lib.conv_wgrad_cl2d(dz, x_cl, self.k, self.stride)
            gb = None if self.conv.bias is None else lib.cast_f32(lib.channel_sums(dz, dz)[0])
        gx = None
        if need_input and self.stride == 1:                    # the stride-1 kernel on the flipped, channel-transposed weights
            gx = lib.conv2d_layer(self.img[self.name, "bwd"], self.cout, self.cin, self.k, 1, dz)
            if add is not None:
                gx = lib.add(gx, add)
        elif need_input:                                        # transposed 5x5 stride-2 convolution (feature_net.py:11,14): four
            gx = lib.conv2d_s2k5_dgrad_packed(self.img[self.name, "s2k5"], self.cin, self.cout, dz, add=add)   # parity classes on the
        return gx, gw, gb                                       # stride-1 MFMA kernel + depth-to-space


_FEAT_ORDER = ("conv0.0", "conv0.1", "conv1.0", "conv1.1", "conv2.0", "conv2.1", "toplayer", "lat1", "lat0", "smooth1", "smooth0")


class FeatureNetTrainFn(torch.autograd.Function):
    """x (N,3,H,W) -> (f2 (N,32,H/4,W/4), smooth1(f1) (N,16,H/2,W/2), smooth0(f0) (N,8,H,W)) through the ``FeatureNet``
    parameter module ``m`` in training mode.  The outputs are NCHW VIEWS of channels-last tensors (what the warp and the
    render-side fetch read)."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, m, x, *params):
        conv, norm = {}, {}
        img = pack_plan.plan_of(lib, m, pack_plan.feature_net_plan, x.device).run()  # every weight image of the step: one launch

        def cbr(name, t):
            blk = getattr(m, name[:5])[int(name[6])]
            conv[name] = _Conv2d(lib, blk.conv, img, name)
            norm[name] = _BatchNormTrain(lib, blk.bn, True)
            return norm[name].forward(conv[name].forward(t))

        def plain(name, t, up=None):
            conv[name] = _Conv2d(lib, getattr(m, name), img, name)
            return conv[name].forward(t, up)
        c0 = cbr("conv0.1", cbr("conv0.0", x.contiguous()))
        c1 = cbr("conv1.1", cbr("conv1.0", c0))
        c2 = cbr("conv2.1", cbr("conv2.0", c1))
        f2 = plain("toplayer", c2)
        f1 = plain("lat1", c1, up=f2)                           # up2(f2) + lat1(c1)
        f0 = plain("lat0", c0, up=f1)
        s1 = plain("smooth1", f1)
        s0 = plain("smooth0", f0)
        ctx.conv, ctx.norm, ctx.lib = conv, norm, lib
        return f2.permute(0, 3, 1, 2), s1.permute(0, 3, 1, 2), s0.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g_f2, g_s1, g_s0):
        with ctx.lib.wgrad_reduce_batch(g_s0):       # the eleven weight gradients' second stages: one launch at the end of the pass
            return FeatureNetTrainFn._backward(ctx, g_f2, g_s1, g_s0)

    @staticmethod
    def _backward(ctx, g_f2, g_s1, g_s0):
        conv, norm, lib = ctx.conv, ctx.norm, ctx.lib
        grads = {}
        cl = lambda g: g.permute(0, 2, 3, 1).contiguous()       # (a no-op when the gradient arrives as a channels-last view)

        def plain_back(name, g):
            gx, gw, gb = conv[name].backward(g)
            grads[name] = (gw, gb)
            return gx

        def cbr_back(name, g, need_input=True, add=None):
            dz, dgamma, dbeta = norm[name].backward(g)
            gx, gw, _ = conv[name].backward(dz, need_input, add)
            grads[name] = (gw, dgamma, dbeta)
            return gx
        g_f0 = plain_back("smooth0", cl(g_s0))
        g_c0 = plain_back("lat0", g_f0)
        g_f1 = lib.up2_adjoint(g_f0, add=plain_back("smooth1", cl(g_s1)))       # up2^T(d f0) + d f1 through smooth1
        g_c1 = plain_back("lat1", g_f1)
        g_top = lib.up2_adjoint(g_f1, add=cl(g_f2))
        g_c2 = plain_back("toplayer", g_top)
        g_c1 = cbr_back("conv2.0", cbr_back("conv2.1", g_c2), add=g_c1)         # the skip branch's gradient rides in the
        g_c0 = cbr_back("conv1.0", cbr_back("conv1.1", g_c1), add=g_c0)         # depth-to-space pass of the stride-2 dgrad
        cbr_back("conv0.0", cbr_back("conv0.1", g_c0), need_input=False)
        out = []
        for name in _FEAT_ORDER:
            out.extend(grads[name])
        return (None, None, None) + tuple(out)


def feature_net_train(lib, m, x):
    """Apply FeatureNetTrainFn with the module's parameters as differentiable inputs (fixed order)."""
    params = []
    for name in _FEAT_ORDER:
        if name.startswith("conv"):
            blk = getattr(m, name[:5])[int(name[6])]
            params += [blk.conv.weight, blk.bn.weight, blk.bn.bias]
        else:
            c = getattr(m, name)
            params += [c.weight, c.bias]
    return FeatureNetTrainFn.apply(lib, m, x, *params)


# ---------------------------------------------------------------------------------------------------------------------
# Agg + NeRF MLP (nerf.py:29-89): fused HIP backward (mlp_train.hip) + weight gradients as position-reductions on the
# matrix cores (enerf_gemm_wgrad).
# ---------------------------------------------------------------------------------------------------------------------
_IMAGE_INDEX_CACHE = {}


def mlp_backward_images(m, S, lib=None):
    """Transposed-weight MFMA images of one ``NerfParams`` for enerf_nerf_mlp_bwd, in the kernel's unit / slot layouts
    (mlp_train.hip header).  Returns (flat image tensor, the 8 offsets b1, b2, b3, b4, b5, b6v, b6m, b7).  The index maps
    only depend on (F, device) and are cached; per step this is ONE index-gather launch over the five weight tensors
    (enerf_gather_images) — or, without the library, eight torch gathers and one concatenation."""
    key = (m.feat_ch, str(m.lr0[0].weight.device), hasattr(m.agg, "view_fc"))
    if key not in _IMAGE_INDEX_CACHE:
        _IMAGE_INDEX_CACHE[key] = _mlp_image_indices(m)
    specs = _IMAGE_INDEX_CACHE[key]
    F = m.feat_ch
    col0, glob = m.color[0].weight.detach(), m.agg.global_fc[0].weight.detach()
    has_view = hasattr(m.agg, "view_fc")
    if lib is not None:
        fkey = key + ("flat",)
        if fkey not in _IMAGE_INDEX_CACHE:
            # (source tensor, first column, row length) of the eight matrices: element [k, r] of matrix e is
            # srcs[src][k * ld + col0 + r]
            ld_c, ld_g = col0.shape[1], glob.shape[1]
            where = [(0, 88, ld_c), (0, 0, ld_c), (1, 0, 24), (2, 0, 32), (3, 0, ld_g), (3, F, ld_g), (3, 2 * F, ld_g), (4, 0, 4)]
            which, idx, offs, o = [], [], [], 0
            for e, ((rows, ks, valid), (src, c0, ld)) in enumerate(zip(specs, where)):
                ok = valid if (e < 7 or has_view) else torch.zeros_like(valid)
                flat = torch.where(ok, ks * ld + c0 + rows, -torch.ones_like(rows)).reshape(-1)
                which.append(torch.full_like(flat, src))
                idx.append(flat)
                offs.append(o)
                o += flat.numel()
            _IMAGE_INDEX_CACHE[fkey] = (torch.cat(which).int().contiguous(), torch.cat(idx).int().contiguous(), offs)
        which, idx, offs = _IMAGE_INDEX_CACHE[fkey]
        srcs = [_c(col0), _c(m.lr0[0].weight.detach()), _c(m.agg.fc[0].weight.detach()), _c(glob)]
        srcs.append(_c(m.agg.view_fc[0].weight.detach()) if has_view else srcs[0])
        return lib.gather_images(srcs, which, idx), offs
    mats = [col0[:, 88:], col0[:, :88], m.lr0[0].weight.detach(), m.agg.fc[0].weight.detach(), glob[:, :F], glob[:, F:2 * F],
            glob[:, 2 * F:], m.agg.view_fc[0].weight.detach() if has_view else None]
    imgs, offs, o = [], [], 0
    for W, (rows, ks, valid) in zip(mats, specs):
        if W is None:
            im = torch.zeros(valid.numel(), device=valid.device)
        else:
            im = torch.where(valid, W[ks, rows], torch.zeros((), device=W.device)).reshape(-1)
        offs.append(o)
        o += im.numel()
        imgs.append(im)
    return torch.cat(imgs).contiguous(), offs


def _mlp_image_indices(m):
    """(row index, k index, validity) tensors of the eight backward tile sets (see mlp_backward_images)."""
    F = m.feat_ch
    R = (F + 3) // 4
    TR, TX = (R + 3) // 4, (R + 1 + 3) // 4
    dev = m.lr0[0].weight.device
    j = torch.arange(16, device=dev)
    g4 = torch.arange(4, device=dev)

    def unit_rows(t, base=0, n=None):                              # row j -> input index base + 16 t + j
        idx = base + 16 * t + j
        return idx if n is None else torch.where(16 * t + j < n, idx, -torch.ones_like(idx))

    def slot_rows(t, with_dir, base=0):                            # row j -> slot (g' = j>>2, r' = 4t + (j&3))
        gp, rp = j >> 2, 4 * t + (j & 3)
        ch = gp * R + rp
        idx = torch.where((rp < R) & (ch < F), base + ch, -torch.ones_like(ch))
        if with_dir:
            idx = torch.where(rp == R, base + F + gp, idx)
        return idx

    vox_rows = lambda base: torch.where((j & 3) < 2, base + 2 * (j >> 2) + (j & 3), -torch.ones_like(j))
    unit_k = lambda kk: 16 * (kk >> 2) + 4 * g4 + (kk & 3)         # k-step kk = (tile, r): lane group g -> unit
    def slot_k(r):                                                 # k-step r over slot-layout outputs: group g -> channel gR + r
        ch = g4 * R + r
        return torch.where(ch < F, ch, -torch.ones_like(ch))

    def build(row_sets, k_sets):
        """image[e, lane = 16 g + j] = W[k_idx[e, g], row_idx[e, j]]: index tensors (E,4,16) + validity."""
        rows = torch.stack([rs for rs in row_sets for _ in k_sets])
        ks = torch.stack([k for _ in row_sets for k in k_sets])
        E = rows.shape[0]
        r3 = rows[:, None, :].expand(E, 4, 16)
        k3 = ks[:, :, None].expand(E, 4, 16)
        return r3.clamp_min(0).contiguous(), k3.clamp_min(0).contiguous(), ((r3 >= 0) & (k3 >= 0)).contiguous()
    k16 = [unit_k(kk) for kk in range(16)]
    k8, k4 = [unit_k(kk) for kk in range(8)], [unit_k(kk) for kk in range(4)]
    dir_rows = lambda t: torch.where(4 * t + (j & 3) == R, j >> 2, -torch.ones_like(j))
    return [
        build([slot_rows(t, True) for t in range(TX)], k16),                                        # b1: color.0, per-view columns
        build([unit_rows(t) for t in range(4)] + [vox_rows(64), unit_rows(0, 72)], k16),            # b2: color.0, shared columns
        build([vox_rows(0), unit_rows(0, 8)], k16),                                                 # b3: lr0
        build([unit_rows(0), unit_rows(1)], k4),                                                    # b4: agg.fc
        build([slot_rows(t, False) for t in range(TR)], k8),                                        # b5: global_fc, a columns
        build([slot_rows(t, False) for t in range(TR)], k8),                                        # b6: var columns
        build([slot_rows(t, False) for t in range(TR)], k8),                                        # b6: mean columns
        build([dir_rows(t) for t in range(TX)], [slot_k(r) for r in range(R)]),                     # b7: view_fc
    ]


class NerfMlpFn(torch.autograd.Function):
    """raw (P,4) = NeRF(vox (P,8), x (P,S,F+4)), forward and backward fused on the HIP kernels (enerf_nerf_mlp_fwd /
    enerf_nerf_mlp_bwd): nothing but the inputs is kept for the backward, which recomputes the forward in registers."""

    @staticmethod
    def forward(ctx, lib: EnerfLib, m, forward_fn, vox, x, *params):
        vox, x = vox.contiguous(), x.contiguous()
        P, S, XW = x.shape
        packed = lib.nerf_pack(m.raw(), XW - 4, m.viewdir_agg, vox.device)
        raw = lib.nerf_mlp_fwd(vox, x, packed, S, XW - 4)
        ctx.lib, ctx.m, ctx.packed = lib, m, packed
        ctx.save_for_backward(vox, x)
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        lib, m = ctx.lib, ctx.m
        vox, x = ctx.saved_tensors
        P, S, XW = x.shape
        F = XW - 4
        packed = ctx.packed
        bimg, offs = mlp_backward_images(m, S, lib)
        # F = 11: the weight gradients of the per-(point, view) layers accumulated inside the kernel (mlp_train.hip WG; the aggregation branch
        # too when the kernel has the registers: S <= 3), the per-point layers through the saved rows
        level = (2 if S <= 3 else 1) if (F == 11 and P > 0) else 0
        res = lib.nerf_mlp_bwd(vox.contiguous(), x.contiguous(), g_raw.contiguous(), packed, bimg, offs, S, F, level=level)
        g_vox, g_x, sv = res[:3]
        part = res[3] if level else None
        hv, G, q, gs, a_, vm, d_c, d_q, d_p2, d_s, d_h, d_agg, d_u, d_g, d_gsum, d_v = sv
        PS = P * S
        x2 = x.reshape(PS, XW)
        # color.2 (1,64) / color.0 (64, 88+F+4) / sigma (1,64) / lr0 (64,24) / fc (16,32) / agg_w (1,32) / global_fc (32,3F) / view_fc (F,4):
        # position reductions over the saved rows (or over the kernel's per-wave partial sums) as ONE grouped call (enerf_gemm_wgrad_group);
        # color.0 and global_fc are assembled in place from their two column blocks
        w_c0 = torch.empty((64, 88 + XW), dtype=torch.float32, device=x.device)
        w_gl = torch.empty((32, 3 * F), dtype=torch.float32, device=x.device)
        gw, members, names = {}, [], []

        def member(name, **kw):
            members.append(kw)
            names.append(name)
        if level == 0:
            member("color.2", a=d_c.reshape(PS, 1), b=q.reshape(PS, 64), bias=True)
        member("color.0", a=d_p2, b=hv, bias=True, into=(w_c0, 0))                               # shared columns [h | vox | agg]
        if level == 0:
            member(None, a=d_q.reshape(PS, 64), b=x2, into=(w_c0, 88))                           # per-view columns [x_s | dir_s]
        else:
            member(None, partials=part["q"], Ca=64, Cb=XW, into=(w_c0, 88))
        member("sigma.0", a=d_s.reshape(P, 1), b=hv, Cb=64, bias=True)
        member("lr0.0", a=d_h, b=hv[:, 64:], bias=True)
        member("agg.fc.0", a=d_agg, b=G, bias=True)
        if level < 2:
            member("agg.agg_w_fc.0", a=d_u.reshape(PS, 1), b=gs.reshape(PS, 32), bias=True)
            member(None, a=d_g.reshape(PS, 32), b=a_.reshape(PS, F), into=(w_gl, 0))             # global_fc, a columns
        else:
            member(None, partials=part["g"], Ca=32, Cb=F, into=(w_gl, 0))
        member("agg.global_fc.0", a=d_gsum, b=vm, bias=True, into=(w_gl, F))                     # global_fc, [var | mean] columns
        if m.viewdir_agg:
            if level < 2:
                member("agg.view_fc.0", a=d_v.reshape(PS, F), b=x2[:, F:], bias=True)
            else:
                w_vf = torch.empty((F, 4), dtype=torch.float32, device=x.device)
                member(None, partials=part["v"], Ca=F, Cb=4, into=(w_vf, 0))
                gw["agg.view_fc.0.weight"] = w_vf
        if level:
            rows = lib.colsum(part["rows"])          # [color.2 w 64 | b | 0 x 15 | agg_w w 32 | b | view_fc b F | 0 ...]
            gw["color.2.weight"], gw["color.2.bias"] = rows[:64].view(1, 64), rows[64:65]
            if level == 2:
                gw["agg.agg_w_fc.0.weight"], gw["agg.agg_w_fc.0.bias"] = rows[80:112].view(1, 32), rows[112:113]
                if m.viewdir_agg:
                    gw["agg.view_fc.0.bias"] = rows[113:113 + F]
        for name, (w_, b_) in zip(names, lib.gemm_wgrad_group(members)):
            if name is not None:
                gw[name + ".weight"] = w_
                if b_ is not None:
                    gw[name + ".bias"] = b_
        grads = [gw.get(n) for n, _ in m.named_parameters()]
        return (None, None, None, g_vox, g_x) + tuple(grads)


def nerf_mlp(lib, m, forward_fn, vox, x):
    """vox (B,P,8), x (B,P,S,F+4) -> raw (B,P,4) through NerfMlpFn (all points of the batch as one row list)."""
    B, P = vox.shape[:2]
    params = [p for _, p in m.named_parameters()]
    raw = NerfMlpFn.apply(lib, m, forward_fn, vox.reshape(B * P, 8), x.reshape(B * P, x.shape[2], x.shape[3]), *params)
    return raw.reshape(B, P, 4)
