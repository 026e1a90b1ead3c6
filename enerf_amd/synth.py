"""Seeded synthetic inputs for the rendering path.

No dataset or checkpoint is reachable offline, so parity and benchmarks use a fixed camera rig
(target at the origin looking at (0,0,650); sources on a small ring looking at the same point) and
either uniform-random or procedurally textured source images.  The batch dict has exactly the
schema ``lib/datasets/dtu/enerf.py:100-119`` produces after ``default_collate``; ``rays_{i}``
follow ``lib/datasets/enerf_utils.py:61-71`` (full-image branch).  numpy only.
"""
from __future__ import annotations

import numpy as np

from .config import EnerfConfig

DTU_K = np.array([[1446.2, 0.0, 331.6], [0.0, 1446.2, 265.6], [0.0, 0.0, 1.0]], dtype=np.float64)
DTU_NEAR_FAR = (425.0, 905.0)
SRC_CENTERS = [(120.0, 0.0, 0.0), (-80.0, 90.0, 0.0), (-60.0, -110.0, 0.0), (95.0, 105.0, 0.0)]


def look_at_w2c(center, target=(0.0, 0.0, 650.0)) -> np.ndarray:
    """World->camera 4x4 (OpenCV convention: +z forward, +y down)."""
    c = np.asarray(center, np.float64)
    z = np.asarray(target, np.float64) - c
    z /= np.linalg.norm(z)
    x = np.cross(np.array([0.0, 1.0, 0.0]), z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], 0)
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = -R @ c
    return E


def full_image_rays(ext: np.ndarray, ixt: np.ndarray, H: int, W: int, scale: float) -> np.ndarray:
    """(H*W, 8) = [o(3), d(3), u, v]; d is the un-normalised K^-1 back-projection in world axes."""
    K = ixt.copy()
    K[:2] *= scale
    h, w = int(H * scale), int(W * scale)
    c2w = np.linalg.inv(ext)
    X, Y = np.meshgrid(np.arange(w), np.arange(h))
    pix = np.stack([X, Y, np.ones_like(X)], -1).astype(np.float64)
    d = pix @ (np.linalg.inv(K).T @ c2w[:3, :3].T)
    o = np.broadcast_to(c2w[:3, 3], d.shape)
    rays = np.concatenate([o, d, X[..., None], Y[..., None]], -1)
    return rays.astype(np.float32).reshape(-1, 8)


def _textured_views(rng, S, H, W, exts, K, plane_z=650.0):
    """Render a textured fronto-parallel plane (z = plane_z) into every source view, so the cost
    volume sees photo-consistent content and the depth distribution is peaked."""
    imgs = np.empty((S, 3, H, W), np.float32)
    X, Y = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    pix = np.stack([X, Y, np.ones_like(X)], -1)
    fr = rng.uniform(0.01, 0.05, size=(3, 4))
    ph = rng.uniform(0, 2 * np.pi, size=(3, 4))
    for s in range(S):
        c2w = np.linalg.inv(exts[s])
        d = pix @ (np.linalg.inv(K).T @ c2w[:3, :3].T)
        o = c2w[:3, 3]
        t = (plane_z - o[2]) / d[..., 2]
        P = o + d * t[..., None]
        for c in range(3):
            v = (np.sin(fr[c, 0] * P[..., 0] + ph[c, 0]) * np.cos(fr[c, 1] * P[..., 1] + ph[c, 1])
                 + 0.5 * np.sin(fr[c, 2] * (P[..., 0] + P[..., 1]) + ph[c, 2])
                 + 0.25 * np.cos(fr[c, 3] * (P[..., 0] - 2 * P[..., 1]) + ph[c, 3]))
            imgs[s, c] = np.clip(v / 1.75, -1, 1)
    imgs += rng.normal(0, 0.02, imgs.shape).astype(np.float32)
    return np.clip(imgs, -1, 1)


def make_batch(H: int = 512, W: int = 640, S: int = 3, cfg: EnerfConfig | None = None, seed: int = 0,
               B: int = 1, textured: bool = False, near_far=DTU_NEAR_FAR, focal_scale: float | None = None,
               mask_box: bool = False) -> dict:
    """Return a dict of float32 numpy arrays with the reference batch schema.

    ``focal_scale`` rescales the DTU intrinsics for reduced-resolution test cases so the field of
    view (and hence the overlap between views) stays that of the 512x640 rig.
    """
    cfg = cfg or EnerfConfig()
    rng = np.random.default_rng(seed)
    if focal_scale is None:
        focal_scale = W / 640.0
    K = DTU_K.copy()
    K[:2] *= focal_scale
    out = {k: [] for k in ("src_inps", "src_exts", "src_ixts", "tar_ext", "tar_ixt", "near_far")}
    rays = {i: [] for i in range(cfg.cas.num)}
    for b in range(B):
        jitter = rng.normal(0, 4.0, size=(S + 1, 3)) if b > 0 else np.zeros((S + 1, 3))
        tar_ext = look_at_w2c(np.array([0.0, 0.0, 0.0]) + jitter[0])
        exts = np.stack([look_at_w2c(np.array(SRC_CENTERS[s % 4]) * (1 + 0.2 * (s // 4)) + jitter[s + 1])
                         for s in range(S)])
        if textured:
            imgs = _textured_views(rng, S, H, W, exts, K)
        else:
            imgs = rng.uniform(-1, 1, size=(S, 3, H, W)).astype(np.float32)
        out["src_inps"].append(imgs)
        out["src_exts"].append(exts)
        out["src_ixts"].append(np.stack([K] * S))
        out["tar_ext"].append(tar_ext)
        out["tar_ixt"].append(K)
        out["near_far"].append(np.array(near_far))
        for i in range(cfg.cas.num):
            rays[i].append(full_image_rays(tar_ext, K, H, W, cfg.cas.render_scale[i]))
    batch = {k: np.stack(v).astype(np.float32) for k, v in out.items()}
    for i in range(cfg.cas.num):
        batch[f"rays_{i}"] = np.stack(rays[i]).astype(np.float32)
    if mask_box:
        m = np.zeros((B, H, W), np.int32)
        m[:, H // 6: H - H // 5, W // 5: W - W // 7] = 1
        batch["mask_at_box"] = m
    return batch
