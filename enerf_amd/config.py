"""Configuration for the ENeRF hot path.

The reference reads a process-global yacs ``cfg`` (``lib/config/config.py:191-201``) inside its
hot path (``lib/networks/enerf/network.py:8``, ``utils.py:6``, ``nerf.py:4``).  The hot path only
ever *reads* ``cfg.enerf.*``; this module restates those fields as plain dataclasses so the HIP
path has no import-time side effects, and :meth:`EnerfConfig.from_yacs` adapts the reference's
global when the package is used as a drop-in inside the reference tree.

Defaults are ``configs/enerf/dtu_pretrain.yaml:17-43``.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Tuple


def _t(x) -> tuple:
    return tuple(x)


@dataclass(frozen=True)
class CascadeConfig:
    """``cfg.enerf.cas_config`` (dtu_pretrain.yaml:27-43). One entry per cascade level."""
    num: int = 2
    depth_inv: Tuple[bool, ...] = (True, False)
    volume_scale: Tuple[float, ...] = (0.125, 0.5)
    volume_planes: Tuple[int, ...] = (64, 8)
    im_feat_scale: Tuple[float, ...] = (0.25, 0.5)
    im_ibr_scale: Tuple[float, ...] = (0.25, 1.0)
    render_scale: Tuple[float, ...] = (0.25, 1.0)
    render_im_feat_level: Tuple[int, ...] = (0, 2)
    nerf_model_feat_ch: Tuple[int, ...] = (32, 8)
    render_if: Tuple[bool, ...] = (True, True)
    num_samples: Tuple[int, ...] = (8, 2)

    def validate(self) -> None:
        for name in ("depth_inv", "volume_scale", "volume_planes", "im_feat_scale", "im_ibr_scale",
                     "render_scale", "render_im_feat_level", "nerf_model_feat_ch", "render_if",
                     "num_samples"):
            if len(getattr(self, name)) != self.num:
                raise ValueError(f"cas_config.{name} must have {self.num} entries")


@dataclass(frozen=True)
class EnerfConfig:
    """``cfg.enerf`` fields the hot path reads."""
    cas: CascadeConfig = field(default_factory=CascadeConfig)
    viewdir_agg: bool = True          # nerf.py:51
    white_bkgd: bool = False          # network.py:42
    chunk_size: int = 1000000         # network.py:47

    def with_cas(self, **kw) -> "EnerfConfig":
        kw = {k: (_t(v) if isinstance(v, (list, tuple)) else v) for k, v in kw.items()}
        return replace(self, cas=replace(self.cas, **kw))

    @classmethod
    def dtu_eval(cls) -> "EnerfConfig":
        """README.md:113 eval command: render_if False,True / volume_planes 48,8."""
        return cls().with_cas(render_if=(False, True), volume_planes=(48, 8))

    @classmethod
    def from_yacs(cls, cfg) -> "EnerfConfig":
        """Adapt the reference's global ``cfg`` (a yacs CfgNode)."""
        e = cfg.enerf
        c = e.cas_config
        cas = CascadeConfig(
            num=int(c.num), depth_inv=_t(bool(v) for v in c.depth_inv),
            volume_scale=_t(float(v) for v in c.volume_scale),
            volume_planes=_t(int(v) for v in c.volume_planes),
            im_feat_scale=_t(float(v) for v in c.im_feat_scale),
            im_ibr_scale=_t(float(v) for v in c.im_ibr_scale),
            render_scale=_t(float(v) for v in c.render_scale),
            render_im_feat_level=_t(int(v) for v in c.render_im_feat_level),
            nerf_model_feat_ch=_t(int(v) for v in c.nerf_model_feat_ch),
            render_if=_t(bool(v) for v in c.render_if),
            num_samples=_t(int(v) for v in c.num_samples))
        cas.validate()
        return cls(cas=cas, viewdir_agg=bool(e.viewdir_agg), white_bkgd=bool(e.white_bkgd),
                   chunk_size=int(e.chunk_size))
