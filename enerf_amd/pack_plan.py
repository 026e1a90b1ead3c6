"""Every packed weight image of a network's training step in ONE launch.

A training step re-packs each convolution's weights into the MFMA operand order of its kernel — once for the forward layer
and once for the input-gradient layer (flipped / channel-transposed weights, the transposed twin of a strided layer, the
parity sub-kernels of the 5x5 stride-2 layers).  Through round 5's first half that was one tiny launch per image: 38
k_conv3d_pack + 22 k_conv2d_pack + 17 k_weights_flip_transpose + the t5 / t2-pair / concat helpers = ~85 launches of ~4.7 us
each inside the captured step, i.e. 0.4 ms of an 11.3 ms step spent on launch floors.

Packing is a pure permutation with zero padding, so it is also an index gather.  ``PackPlan`` derives the gather map of a
network by running the library's own pack entries ONCE on tensors that hold element indices instead of weights (exact in
fp32: every tensor here has < 2^24 elements), then serves a step with one ``enerf_gather_images`` launch over the
parameters.  The images are bit-identical to the per-layer pack kernels' by construction (tests/test_training.py compares
them).  Reference for what is being packed: cost_reg_net.py:7-86 (layers), feature_net.py:4-36.
"""
from __future__ import annotations

import weakref

import torch

_S1, _S2, _T2 = 0, 1, 2
_ALIGN = 64                                  # floats: images start on 256-byte boundaries (LDS-DMA / dwordx4 operand loads)


class PackPlan:
    """``sources``: ordered {name: tensor getter}; images are added by tracing (``trace``) and served by ``run``."""

    def __init__(self, lib, device):
        self.lib, self.device = lib, device
        self._getters, self._names, self._base, self._numel = [], [], [], []
        self._total = 0
        self._parts, self._which, self._idx = [], None, None
        self.slices = {}
        self._n = 0
        self._one = torch.ones(1, dtype=torch.float32, device=device)
        self._const = self.source("__one__", lambda: self._one)

    # ---- sources ----
    def source(self, name, getter):
        t = getter()
        self._getters.append(getter)
        self._names.append(name)
        self._base.append(self._total)
        self._numel.append(t.numel())
        self._total += t.numel()
        if self._total >= 1 << 24:
            raise ValueError("PackPlan: index tracing is exact below 2^24 source elements")
        return len(self._getters) - 1

    def index_tensor(self, src):
        """A tensor shaped like source ``src`` whose elements are their own 1-based global indices (0 = padding)."""
        t = self._getters[src]()
        return (torch.arange(t.numel(), dtype=torch.float32, device=self.device) + float(self._base[src] + 1)).view(t.shape)

    # ---- images ----
    def add(self, key, traced, *regions):
        """``traced``: a packed image produced from index tensors (0 = padding).  ``regions``: (offset, cp, bias source or
        None, bias length) of each [scale (cp) | shift (cp)] epilogue block inside it: scale is the constant 1, shift the bias
        (or 0) — the pack kernels write VALUES there, not indices."""
        g = traced.round().to(torch.int64) - 1                                     # -1 = padding
        bases = torch.tensor(self._base, dtype=torch.int64, device=self.device)
        which = torch.searchsorted(bases, g.clamp(min=0), right=True) - 1
        idx = torch.where(g >= 0, g - bases[which], g)
        which = torch.where(g >= 0, which, torch.zeros_like(which))
        for o, cp, bias_src, bias_n in regions:
            which[o:o + cp] = self._const
            idx[o:o + cp] = 0
            which[o + cp:o + 2 * cp] = 0
            idx[o + cp:o + 2 * cp] = -1
            if bias_src is not None:
                which[o + cp:o + cp + bias_n] = bias_src
                idx[o + cp:o + cp + bias_n] = torch.arange(bias_n, dtype=torch.int64, device=self.device)
        n = traced.numel()
        pad = (-n) % _ALIGN
        if pad:
            which = torch.cat([which, torch.zeros(pad, dtype=torch.int64, device=self.device)])
            idx = torch.cat([idx, torch.full((pad,), -1, dtype=torch.int64, device=self.device)])
        self.slices[key] = (self._n, n)
        self._n += n + pad
        self._parts.append((which, idx))

    def finish(self):
        self._which = torch.cat([w for w, _ in self._parts]).to(torch.int32).contiguous()
        self._idx = torch.cat([i for _, i in self._parts]).to(torch.int32).contiguous()
        self._parts = None
        return self

    def run(self):
        """-> {key: packed image} (views of one buffer), one launch."""
        srcs = [g().detach() for g in self._getters]
        srcs = [s if s.is_contiguous() else s.contiguous() for s in srcs]
        buf = self.lib.gather_images(srcs, self._which, self._idx)
        return {k: buf[o:o + n] for k, (o, n) in self.slices.items()}


def _conv3d_regions(lib, cin, cout, kind):
    rt = (cout + 15) // 16
    wf, cp = 27 * (cin // 4) * rt * 64, rt * 16
    total = lib.dll.enerf_conv3d_layer_packed_floats(cin, cout, kind)
    assert total - wf - 2 * cp in (0, 18 * 4 * 64), (cin, cout, kind, total)
    return (wf, cp, None, 0)


def cost_reg_plan(lib, m, device):
    """Images of ``CostRegTrainFn``: (i, "fwd") / (i, "bwd") per block, ("heads", "fwd" / "bwd")."""
    plan = PackPlan(lib, device)
    blocks = [0, 1, 2, 3, 4] + ([5, 6, 7] if m.full else []) + [9, 11]
    kinds = {0: _S1, 1: _S2, 2: _S1, 3: _S2, 4: _S1, 5: _S2, 6: _S1, 7: _T2, 9: _T2, 11: _T2}
    for i in blocks:
        mod = getattr(m, f"conv{i}")
        conv = mod[0] if i in (7, 9, 11) else mod.conv
        src = plan.source(f"conv{i}", lambda c=conv: c.weight)
        wi, kind = plan.index_tensor(src), kinds[i]
        if kind == _T2:
            cin, cout = wi.shape[0], wi.shape[1]
        else:
            cout, cin = wi.shape[0], wi.shape[1]
        plan.add((i, "fwd"), lib.conv3d_layer_pack(wi, cin, cout, kind), _conv3d_regions(lib, cin, cout, kind))
        if kind == _S1:
            bwd, bk = lib.conv3d_layer_pack(lib.weights_flip_transpose(wi), cout, cin, _S1), _S1
        elif kind == _S2:
            bwd, bk = lib.conv3d_layer_pack(wi, cout, cin, _T2), _T2
        else:
            bwd, bk = lib.conv3d_layer_pack(wi, cout, cin, _S2), _S2
        plan.add((i, "bwd"), bwd, _conv3d_regions(lib, cout, cin, bk))
    # heads: feat_conv (8 -> 8) ++ depth_conv (8 -> 1) as one 8 -> 16 layer (rows 9..15 zero)
    # (the getters capture the SUBMODULES, never `m` itself: `m` is the weak key of _PLANS, and a value that holds its key
    # strongly would keep module, index tensors and library alive for ever)
    sf = plan.source("feat_conv", lambda c=m.feat_conv[0]: c.weight)
    sd = plan.source("depth_conv", lambda c=m.depth_conv[0]: c.weight)
    w16 = lib.concat2_pad(plan.index_tensor(sf).reshape(-1), plan.index_tensor(sd).reshape(-1), 16 * 8 * 27).view(16, 8, 3, 3, 3)
    plan.add(("heads", "fwd"), lib.conv3d_layer_pack(w16, 8, 16, _S1), _conv3d_regions(lib, 8, 16, _S1))
    plan.add(("heads", "bwd"), lib.conv3d_layer_pack(lib.weights_flip_transpose(w16), 16, 8, _S1), _conv3d_regions(lib, 16, 8, _S1))
    return plan.finish()


FEAT_LAYERS = ("conv0.0", "conv0.1", "conv1.0", "conv1.1", "conv2.0", "conv2.1", "toplayer", "lat1", "lat0", "smooth1", "smooth0")


def feature_net_plan(lib, m, device):
    """Images of ``FeatureNetTrainFn``: (name, "fwd"), (name, "bwd") for the stride-1 layers that need an input gradient,
    (name, "s2k5") for the two 5x5 stride-2 layers."""
    plan = PackPlan(lib, device)
    for name in FEAT_LAYERS:
        conv = getattr(m, name[:5])[int(name[6])].conv if name.startswith("conv") else getattr(m, name)
        src = plan.source(name, lambda c=conv: c.weight)
        bsrc = None if conv.bias is None else plan.source(name + ".bias", lambda c=conv: c.bias)
        wi = plan.index_tensor(src)
        cout, cin, k, _ = wi.shape
        cp = (cout + 15) // 16 * 16
        fwd = lib.conv2d_layer_pack(wi, None, cin, cout, k)
        plan.add((name, "fwd"), fwd, (fwd.numel() - 2 * cp, cp, bsrc, cout))
        if name == "conv0.0":
            continue                                                             # the image needs no gradient
        if int(conv.stride[0]) == 1:
            cpb = (cin + 15) // 16 * 16
            bwd = lib.conv2d_layer_pack(lib.weights_flip_transpose(wi), None, cout, cin, k)
            plan.add((name, "bwd"), bwd, (bwd.numel() - 2 * cpb, cpb, None, 0))
        else:
            # parts x [weights | scale | shift | slack to a 64-float boundary] (train_glue.hip: enerf_conv2d_s2k5_dgrad_pack)
            traced, parts, pk, pf, cout3 = lib.conv2d_s2k5_dgrad_pack(wi)
            cp3 = (cout3 + 15) // 16 * 16
            plan.add((name, "s2k5"), traced, *[(q * pk + pf - 2 * cp3, cp3, None, 0) for q in range(parts)])
    return plan.finish()


_PLANS = weakref.WeakKeyDictionary()          # module -> {(builder, device, library): plan}; NOT an attribute of the module: a plan
                                              # holds the ctypes library, and modules must stay deep-copyable / picklable


def _lib_key(lib):
    """What identifies a loaded library: its path (two EnerfLib objects of one .so serve the same plan; `id()` of a freed object
    can be handed to the next one)."""
    return getattr(lib, "path", None) or getattr(getattr(lib, "dll", None), "_name", None) or repr(lib)


def plan_of(lib, m, build, device):
    """The module's cached plan (built on first use — the eager warm-up steps of GraphedTrainStep — per device).  Building
    launches pack kernels, host-to-device copies and index arithmetic, none of which may be captured: a plan that is missing
    while the stream is capturing is an error, not a silent build."""
    cache = _PLANS.setdefault(m, {})
    key = (build.__name__, str(device), _lib_key(lib))
    if key not in cache:
        if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"pack_plan.{build.__name__}: no plan for this module / device / library yet and the stream is capturing "
                               "— run one eager training step first (GraphedTrainStep's warm-up does)")
        cache[key] = build(lib, m, device)
    return cache[key]
