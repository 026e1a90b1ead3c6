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
