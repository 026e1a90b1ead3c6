"""Golden vectors for the steps either side of the path from the UNMODIFIED reference helpers.

TEST INFRASTRUCTURE (build container only).  Imports ``lib/datasets/enerf_utils.py`` (with the cv2 stand-in of
oracle/kornia_shim) and ``lib/utils/net_utils.py`` (termcolor stand-in) and records

  * ``build_rays`` (enerf_utils.py:25-71) in the test split (full image) at both cascade levels,
  * ``build_rays`` in the train split under ``configs/enerf/zjumocap_eval.yaml`` (mask sampling 75 % + uniform + 4 patches
    of 64 px: :33-56) with ``np.random.seed(0)``,
  * ``gen_rays_bbox`` (net_utils.py:13-28) on the level-1 rays,

into ``tests/golden/io_rays.npz`` together with their (synthetic, seeded) inputs.

    python oracle/make_golden_io.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle.ref_loader import load_reference
    cfg, _ = load_reference("configs/enerf/zjumocap_eval.yaml", [])
    from lib.datasets import enerf_utils as ref_rays
    from lib.utils import net_utils as ref_net
    from enerf_amd.synth import make_zju_batch, ZJU_BBOX

    H = W = 256
    b = make_zju_batch(H, W, 2, seed=11)
    tar_ext, tar_ixt = b["tar_ext"][0], b["tar_ixt"][0]
    tar_img = np.random.default_rng(0).uniform(0, 1, (H, W, 3)).astype(np.float32)
    tar_msk = b["mask_at_box"][0].astype(np.uint8)
    save = {"in/tar_ext": tar_ext, "in/tar_ixt": tar_ixt, "in/tar_msk": tar_msk, "in/H": np.array(H), "in/W": np.array(W)}
    for level in range(2):
        rays, rgb, msk = ref_rays.build_rays(tar_img, tar_ext, tar_ixt, tar_msk, level, "test")
        save[f"test/rays_{level}"] = rays
    np.random.seed(0)
    for level in range(2):
        rays, rgb, msk = ref_rays.build_rays(tar_img, tar_ext, tar_ixt, tar_msk, level, "train")
        save[f"train/rays_{level}"] = rays
        save[f"train/msk_{level}"] = msk
    bounds = np.stack(ZJU_BBOX).astype(np.float32)
    rays1 = torch.from_numpy(save["test/rays_1"])
    save["bbox/bounds"] = bounds
    save["bbox/mask"] = ref_net.gen_rays_bbox(rays1, torch.from_numpy(bounds)).numpy().astype(np.int32)
    save["meta/scales"] = np.array([float(s) for s in cfg.enerf.cas_config.render_scale])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "io_rays.npz"), **save)
    print({k: v.shape for k, v in save.items()}, "bbox hits", save["bbox/mask"].mean())


if __name__ == "__main__":
    main()
