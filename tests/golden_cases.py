"""Shared helpers: golden-case registry, batch/config/weights loading (tests only)."""
from __future__ import annotations

import os

import numpy as np
import torch

from enerf_amd.config import EnerfConfig
from enerf_amd.synth import make_batch, make_lego_batch, make_zju_batch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# must mirror oracle/make_golden.py::CASES (the generator is the source of truth)
CASES = {
    "tiny_s3": dict(H=32, W=64, S=3, planes=(8, 8), render_if=(True, True), seed=1, textured=False, human=False),
    "tiny_s2": dict(H=32, W=64, S=2, planes=(8, 8), render_if=(True, True), seed=2, textured=True, human=False),
    "tiny_s4_mask": dict(H=32, W=64, S=4, planes=(8, 8), render_if=(False, True), seed=3, textured=True, human=True),
    "small_s3_eval": dict(H=64, W=96, S=3, planes=(16, 8), render_if=(False, True), seed=4, textured=True, human=False),
    "lego_small": dict(H=64, W=64, S=4, planes=(64, 8), render_if=(True, True), seed=5, human=False, rig="lego"),
    "zju_small": dict(H=64, W=64, S=4, planes=(32, 8), render_if=(False, True), seed=6, human=True, rig="zju"),
    # configs/enerf/llff_eval.yaml (planes 32,8; 640x960) at 1/5 size (2:3 aspect, level-0 volume 32 x 16 x 24)
    "llff_small": dict(H=128, W=192, S=3, planes=(32, 8), render_if=(False, True), seed=7, textured=True, human=False),
}


def case_config(name: str) -> EnerfConfig:
    c = CASES[name]
    return EnerfConfig().with_cas(volume_planes=c["planes"], render_if=c["render_if"])


def case_batch(name: str, as_torch: bool = True) -> dict:
    c = CASES[name]
    if c.get("rig") == "lego":
        b = make_lego_batch(c["H"], c["W"], c["S"], case_config(name), seed=c["seed"])
    elif c.get("rig") == "zju":
        b = make_zju_batch(c["H"], c["W"], c["S"], case_config(name), seed=c["seed"])
    else:
        b = make_batch(c["H"], c["W"], c["S"], case_config(name), seed=c["seed"], textured=c["textured"],
                       mask_box=c["human"])
    return {k: torch.from_numpy(v) for k, v in b.items()} if as_torch else b


def load_weights() -> dict:
    z = np.load(os.path.join(GOLDEN, "weights_seed0.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def load_golden(name: str) -> dict:
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    return {k: z[k] for k in z.files}


# ---- full-size frames pinned to the reference itself (oracle/make_golden.py::FULL_CASES; sparse digests, < 1 MB each) ----
FULL_CASES = {
    "dtu_full": dict(H=512, W=640, S=3, planes=(48, 8), render_if=(False, True), seed=0, textured=True, human=False),
    # BASELINE configs[0] (configs/enerf/dtu/scan114.yaml shape): 512x640, 3 views, planes 48,8, BOTH levels rendered
    "dtu_full_tt": dict(H=512, W=640, S=3, planes=(48, 8), render_if=(True, True), seed=0, textured=True, human=False),
    "lego_full": dict(H=800, W=800, S=4, planes=(64, 8), render_if=(True, True), seed=5, human=False, rig="lego"),
    "zju_full": dict(H=1024, W=1024, S=4, planes=(32, 8), render_if=(False, True), seed=6, human=True, rig="zju"),
    # configs/enerf/llff_eval.yaml at its own size (the reference's fourth eval config; not a BASELINE config)
    "llff_full": dict(H=640, W=960, S=3, planes=(32, 8), render_if=(False, True), seed=7, textured=True, human=False),
}


def full_case_config(name: str) -> EnerfConfig:
    c = FULL_CASES[name]
    return EnerfConfig().with_cas(volume_planes=c["planes"], render_if=c["render_if"])


def full_case_batch(name: str) -> dict:
    c = FULL_CASES[name]
    cfg = full_case_config(name)
    if c.get("rig") == "lego":
        return make_lego_batch(c["H"], c["W"], c["S"], cfg, seed=c["seed"])
    if c.get("rig") == "zju":
        return make_zju_batch(c["H"], c["W"], c["S"], cfg, seed=c["seed"])
    return make_batch(c["H"], c["W"], c["S"], cfg, seed=c["seed"], textured=c["textured"])


def check_sparse_golden(name: str, out: dict, rel_tol: float, mids: dict | None = None) -> dict:
    """Compare full-size outputs (torch tensors or arrays) with the reference's sparse digest: the stored rows
    (every ``meta/stride``-th ray / pixel) within rel_tol of max|ref|, and the whole-tensor L2 norm and sum.
    Returns {key: worst relative error} for reporting."""
    g = load_golden(name)
    stride = int(g["meta/stride"])
    keys = sorted({k.split("/")[1] for k in g if k.startswith("out/")})
    assert sorted(out) == keys, (sorted(out), keys)
    worst = {}
    groups = [("out", out)] + ([("mid", mids)] if mids else [])
    for grp, tensors in groups:
        for k in sorted({k.split("/")[1] for k in g if k.startswith(grp + "/")}):
            if k not in tensors:
                continue
            a = tensors[k]
            a = np.asarray(a.detach().cpu().numpy() if hasattr(a, "detach") else a)
            pre = f"{grp}/{k}"
            if grp == "out":
                assert tuple(a.shape) == tuple(g[pre + "/shape"]), (k, a.shape, g[pre + "/shape"])
            ref_rows = g[pre + "/rows"]
            rows = a.reshape(-1, ref_rows.shape[1])[::stride]
            scale = max(float(g[pre + "/absmax"]), 1e-12)
            err = float(np.abs(rows.astype(np.float64) - ref_rows).max()) / scale
            f = a.astype(np.float64).reshape(-1)
            nerr = abs(float(np.sqrt((f * f).sum())) - float(g[pre + "/norm"])) / max(float(g[pre + "/norm"]), 1e-12)
            assert err < rel_tol, (name, k, "rows", err)
            assert nerr < rel_tol, (name, k, "norm", nerr)
            worst[k] = max(err, nerr)
    return worst
