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


# --------------------------------------------------------------------------------------------------
# BASELINE config 3: NeRF-Synthetic "lego" rig (lib/datasets/nerf/enerf.py:39-51,92; configs/enerf/nerf/lego.yaml)
# --------------------------------------------------------------------------------------------------
LEGO_CAMERA_ANGLE_X = 0.6911112070083618        # transforms_train.json of the Blender scenes (all eight share it)
LEGO_NEAR_FAR = (2.5, 5.5)                      # nerf/enerf.py:92
LEGO_RADIUS = 4.0311289                         # Blender cameras sit on a sphere of this radius around the origin


def lego_intrinsics(H: int, W: int) -> np.ndarray:
    """nerf/enerf.py:46-49 at 800x800: f = 400/tan(angle/2), principal point (400,400); scaled for smaller test images."""
    f = 0.5 * W / np.tan(0.5 * LEGO_CAMERA_ANGLE_X)
    return np.array([[f, 0.0, 0.5 * W], [0.0, f, 0.5 * H], [0.0, 0.0, 1.0]], np.float64)


def orbit_w2c(azim_deg: float, elev_deg: float, radius: float, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)) -> np.ndarray:
    """World->camera (OpenCV axes: +z forward, +y down) of a camera on a sphere around ``target`` — what
    ``inv(transform_matrix @ diag(1,-1,-1,1))`` gives for the Blender scenes (nerf/enerf.py:40-44)."""
    a, e = np.deg2rad(azim_deg), np.deg2rad(elev_deg)
    c = np.asarray(target, np.float64) + radius * np.array([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)])
    z = np.asarray(target, np.float64) - c
    z /= np.linalg.norm(z)
    x = np.cross(z, np.asarray(up, np.float64))       # image right
    x /= np.linalg.norm(x)
    y = np.cross(z, x)                                # image down
    R = np.stack([x, y, z], 0)
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = -R @ c
    return E


def _render_ellipsoid(rng_tex, ext, K, H, W, axes, centre=(0.0, 0.0, 0.0), background=1.0):
    """A procedurally textured ellipsoid seen from camera (ext, K): (3,H,W) in [0,1] and the hit mask.  The texture is
    a function of the surface point, so every view sees photo-consistent content."""
    fr, ph = rng_tex
    c2w = np.linalg.inv(ext)
    X, Y = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    d = np.stack([X, Y, np.ones_like(X)], -1) @ (np.linalg.inv(K).T @ c2w[:3, :3].T)
    o = c2w[:3, 3] - np.asarray(centre, np.float64)
    s = 1.0 / np.asarray(axes, np.float64)
    os_, ds_ = o * s, d * s
    A = (ds_ * ds_).sum(-1)
    Bq = 2.0 * (ds_ * os_).sum(-1)
    Cq = (os_ * os_).sum() - 1.0
    disc = Bq * Bq - 4 * A * Cq
    hit = disc > 0
    t = np.where(hit, (-Bq - np.sqrt(np.maximum(disc, 0))) / (2 * A), 0.0)
    P = o + d * t[..., None]
    img = np.empty((3, H, W), np.float32)
    for c in range(3):
        v = (np.sin(fr[c, 0] * P[..., 0] + ph[c, 0]) * np.cos(fr[c, 1] * P[..., 1] + ph[c, 1])
             + 0.5 * np.sin(fr[c, 2] * (P[..., 0] + P[..., 2]) + ph[c, 2])
             + 0.25 * np.cos(fr[c, 3] * (P[..., 1] - 2 * P[..., 2]) + ph[c, 3]))
        img[c] = np.where(hit, 0.5 + 0.5 * np.clip(v / 1.75, -1, 1), background)
    return img, hit


def make_lego_batch(H: int = 800, W: int = 800, S: int = 4, cfg: EnerfConfig | None = None, seed: int = 0) -> dict:
    """BASELINE config 3: lego pinhole intrinsics, near_far [2.5, 5.5], S nearest training
    cameras of an orbit, white background (read_image: rgb*a + (1-a), nerf/enerf.py:125-128)."""
    cfg = cfg or EnerfConfig()
    rng = np.random.default_rng(seed)
    K = lego_intrinsics(H, W)
    tex = (rng.uniform(2.0, 7.0, size=(3, 4)), rng.uniform(0, 2 * np.pi, size=(3, 4)))
    tar_ext = orbit_w2c(30.0, 32.0, LEGO_RADIUS)
    offs = [(-14.0, 3.0), (12.0, -4.0), (5.0, 11.0), (-7.0, -12.0), (18.0, 7.0)]
    exts = np.stack([orbit_w2c(30.0 + a, 32.0 + e, LEGO_RADIUS) for a, e in offs[:S]])
    imgs = np.stack([_render_ellipsoid(tex, exts[s], K, H, W, (1.1, 0.8, 0.7))[0] for s in range(S)])
    imgs = np.clip(imgs + rng.normal(0, 0.01, imgs.shape), 0, 1).astype(np.float32)
    batch = {"src_inps": (imgs * 2 - 1)[None], "src_exts": exts[None], "src_ixts": np.stack([K] * S)[None],
             "tar_ext": tar_ext[None], "tar_ixt": K[None], "near_far": np.array([LEGO_NEAR_FAR])}
    batch = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in batch.items()}
    for i in range(cfg.cas.num):
        batch[f"rays_{i}"] = full_image_rays(tar_ext, K, H, W, cfg.cas.render_scale[i])[None]
    return batch


# --------------------------------------------------------------------------------------------------
# BASELINE config 4: ZJU-MoCap rig (lib/datasets/zjumocap/enerf.py:35-70,157-166; configs/enerf/zjumocap_eval.yaml)
# --------------------------------------------------------------------------------------------------
ZJU_BBOX = (np.array([-0.45, -0.35, -0.95]), np.array([0.45, 0.35, 0.85]))      # SMPL vertices min/max -+ 0.1 m
ZJU_FOCAL_1024 = 1075.0                                                          # CoreView_313 cameras, 1024x1024


def _hull_mask(pts2d: np.ndarray, H: int, W: int) -> np.ndarray:
    """Filled convex hull of the projected box corners (data_utils.get_bound_2d_mask fills the six faces with
    cv2.fillPoly; their union is this hull)."""
    p = np.unique(np.round(pts2d).astype(np.int64), axis=0)
    p = p[np.lexsort((p[:, 1], p[:, 0]))]

    def half(points):
        h = []
        for q in points:
            while len(h) >= 2 and ((h[-1][0] - h[-2][0]) * (q[1] - h[-2][1]) - (h[-1][1] - h[-2][1]) * (q[0] - h[-2][0])) <= 0:
                h.pop()
            h.append(q)
        return h
    hull = half(list(p))[:-1] + half(list(p[::-1]))[:-1]
    X, Y = np.meshgrid(np.arange(W), np.arange(H))
    inside = np.ones((H, W), bool)
    for a, b in zip(hull, hull[1:] + hull[:1]):
        inside &= ((b[0] - a[0]) * (Y - a[1]) - (b[1] - a[1]) * (X - a[0])) >= 0
    return inside.astype(np.int32)


def make_zju_batch(H: int = 1024, W: int = 1024, S: int = 4, cfg: EnerfConfig | None = None, seed: int = 0) -> dict:
    """BASELINE config 4: a ring of inward-looking cameras 3 m from a person-sized box,
    near_far = depth range of the box corners in the target camera (zjumocap/enerf.py:162-164), ``mask_at_box`` =
    projected box, source images zero outside the foreground mask (zjumocap/enerf.py:153: ``img[mask == 0] = 0``)."""
    cfg = cfg or EnerfConfig()
    rng = np.random.default_rng(seed)
    f = ZJU_FOCAL_1024 * W / 1024.0
    K = np.array([[f, 0.0, 0.5 * W], [0.0, f, 0.5 * H], [0.0, 0.0, 1.0]], np.float64)
    tex = (rng.uniform(4.0, 12.0, size=(3, 4)), rng.uniform(0, 2 * np.pi, size=(3, 4)))
    ring = lambda az, el=4.0: orbit_w2c(az, el, 3.0, target=(0.0, 0.0, -0.05))
    tar_ext = ring(20.0)
    exts = np.stack([ring(20.0 + a, 4.0 + e) for a, e in [(-17.0, 1.0), (17.0, -1.0), (-34.0, 2.0), (34.0, 0.5), (51.0, 0.0)][:S]])
    axes = (0.30, 0.22, 0.80)
    imgs = np.stack([_render_ellipsoid(tex, exts[s], K, H, W, axes, centre=(0.0, 0.0, -0.05), background=0.0)[0]
                     for s in range(S)]).astype(np.float32)
    lo, hi = ZJU_BBOX
    corners = np.array([[x, y, z, 1.0] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    cam = corners @ tar_ext.T
    near_far = np.array([max(cam[:, 2].min(), 0.1), cam[:, 2].max()])
    pts = cam[:, :3] @ K.T
    mask = _hull_mask(pts[:, :2] / pts[:, 2:], H, W)
    batch = {"src_inps": (imgs * 2 - 1)[None], "src_exts": exts[None], "src_ixts": np.stack([K] * S)[None],
             "tar_ext": tar_ext[None], "tar_ixt": K[None], "near_far": near_far[None]}
    batch = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in batch.items()}
    for i in range(cfg.cas.num):
        batch[f"rays_{i}"] = full_image_rays(tar_ext, K, H, W, cfg.cas.render_scale[i])[None]
    batch["mask_at_box"] = mask[None]
    return batch
