"""Evaluator-shaped consumer of the rendered frame that keeps the images on the device.

Mirrors the part of ``lib/evaluators/enerf.py`` the hot path can serve: masked / centre-cropped PSNR per rendered level
(:45-71) and the NeRF / MVS depth statistics (:88-103), same ``evaluate(output, batch)`` / ``summarize()`` surface, so
``run.py:69-70`` can call it in place of the skimage/numpy evaluator.  The reductions run in ``enerf_eval_stats`` (io.hip):
one 48-byte D2H copy per (frame, level) instead of the fp32 images.  SSIM and LPIPS stay where they are (skimage / lpips).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .config import EnerfConfig
from .lib import EnerfLib, get_lib


def nearest_resize_index(src: int, dst: int, device) -> torch.Tensor:
    """Source index of every destination pixel along one axis under cv2.resize(..., interpolation=cv2.INTER_NEAREST):
    ``min(floor(d * (1 / (dst / src))), src - 1)`` in double precision (OpenCV resizeNN)."""
    inv = 1.0 / (float(dst) / float(src))
    idx = torch.floor(torch.arange(dst, dtype=torch.float64) * inv).to(torch.int64).clamp_(max=src - 1)
    return idx.to(device)


class DeviceEvaluator:
    def __init__(self, cfg: EnerfConfig, eval_center: bool = False, eval_depth: bool = False,
                 lib: Optional[EnerfLib] = None):
        self.cfg, self.eval_center, self.eval_depth = cfg, eval_center, eval_depth
        self._lib = lib
        self.reset()

    def reset(self):
        self.psnrs: List[float] = []
        self.level_psnrs: Dict[int, List[float]] = {}
        self.abs, self.acc_2, self.acc_10 = [], [], []
        self.mvs_abs, self.mvs_acc_2, self.mvs_acc_10 = [], [], []

    @property
    def lib(self) -> EnerfLib:
        if self._lib is None:
            self._lib = get_lib()
        return self._lib

    def evaluate(self, output: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor]):
        """evaluators/enerf.py:38-103 (PSNR and depth parts)."""
        cas = self.cfg.cas
        B, S, _, H, W = batch["src_inps"].shape
        for i in range(cas.num):
            if not cas.render_if[i]:
                continue
            h, w = int(H * cas.render_scale[i]), int(W * cas.render_scale[i])
            crop = (int(h * 0.1), int(w * 0.1)) if self.eval_center else (0, 0)
            for b in range(B):
                pred = output[f"rgb_level{i}"][b].contiguous()
                gt = batch[f"rgb_{i}"][b].reshape(h * w, 3).contiguous()
                mask = batch[f"msk_{i}"][b].reshape(h * w).contiguous()
                depth_args = {}
                last = i == cas.num - 1
                if self.eval_depth and last and "tar_dpt" in batch:
                    depth_args = dict(pred_depth=output[f"depth_level{i}"][b].contiguous(),
                                      gt_depth=batch["tar_dpt"][b].reshape(-1).contiguous())
                st = self.lib.eval_stats(pred, gt, mask, image_hw=(h, w), crop=crop, **depth_args)
                self.level_psnrs.setdefault(i, []).append(st["psnr"])
                if last:
                    self.psnrs.append(st["psnr"])
                    if "abs" in st:
                        self.abs.append(st["abs"]); self.acc_2.append(st["acc_2"]); self.acc_10.append(st["acc_10"])
                    mvs_key = f"depth_mvs_level{i}"
                    if depth_args and mvs_key in output:
                        # :91-103 — the cost-volume depth map against the ground truth resized to ITS resolution with
                        # cv2.INTER_NEAREST (src index = min(floor(dst * src/dst), src-1)), gathered on the device
                        mvs = output[mvs_key][b]
                        gt_map = batch["tar_dpt"][b].reshape(h, w)
                        ys = nearest_resize_index(h, mvs.shape[0], mvs.device)
                        xs = nearest_resize_index(w, mvs.shape[1], mvs.device)
                        mvs_gt = gt_map.index_select(0, ys).index_select(1, xs).contiguous()
                        ms = self.lib.depth_stats(mvs.contiguous(), mvs_gt)
                        if ms:
                            self.mvs_abs.append(ms["abs"]); self.mvs_acc_2.append(ms["acc_2"]); self.mvs_acc_10.append(ms["acc_10"])

    def summarize(self) -> dict:
        mean = lambda v: sum(v) / len(v) if v else float("nan")
        ret = {"psnr": mean(self.psnrs)}
        ret.update({f"psnr_level{i}": mean(v) for i, v in self.level_psnrs.items()})
        if self.abs:
            ret.update(abs=mean(self.abs), acc_2=mean(self.acc_2), acc_10=mean(self.acc_10))
        if self.mvs_abs:                 # the reference accumulates these (:101-103) but never prints them; reported here
            ret.update(mvs_abs=mean(self.mvs_abs), mvs_acc_2=mean(self.mvs_acc_2), mvs_acc_10=mean(self.mvs_acc_10))
        self.reset()
        return ret
