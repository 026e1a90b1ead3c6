"""Adversarial-regime parity: near-one-hot depth softmax / clamped std, [d +- std] on the near_far
clamps, a source camera whose pz changes sign inside the volume, exp(-sigma) underflow, white_bkgd.  The fixtures are outputs
of the UNMODIFIED reference on these inputs (``oracle/make_golden.py --case adv_*``, case table ``tests/adversarial.py``).

CPU: the oracle and the lane-emulated kernel sources against the reference; GPU (-m gpu): the product library, 1e-4.

Measured on the lane emulator: worst deviation 3.7e-5 of max|ref| (adv_clamp's std_level1); the oracle at 8 threads moves by
up to 2e-5 against the reference's 1-thread run (adv_sigma's rgb_level1).
"""
import json

import numpy as np
import pytest
import torch

from adversarial import ADV_CASES, adv_batch, adv_config, regime_stats, tweak_weights
from golden_cases import load_golden, load_weights
from oracle import enerf_oracle as O

REL_TOL = 1e-4


def _rel(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-12))


def _tbatch(case, dev=None):
    b = {k: torch.from_numpy(v) for k, v in adv_batch(case).items()}
    return b if dev is None else {k: v.to(dev) for k, v in b.items()}


def _compare(case, out, gold, tol):
    worst = {}
    for k in sorted(k[4:] for k in gold if k.startswith("out/")):
        a = out[k].detach().cpu().numpy() if hasattr(out[k], "detach") else np.asarray(out[k])
        assert a.shape == gold["out/" + k].shape, (case, k)
        assert np.isfinite(a).all(), (case, k)
        worst[k] = _rel(a, gold["out/" + k])
    bad = {k: v for k, v in worst.items() if v >= tol}
    assert not bad, (case, bad)
    return worst


def test_cases_are_in_their_regimes():
    """The fixture's recorded regime fractions (measured on the reference's own intermediates when it was generated)."""
    st = {c: json.loads(str(load_golden(c)["meta/regime"])) for c in ADV_CASES}
    assert st["adv_onehot"]["onehot_frac_0"] > 0.5 and st["adv_onehot"]["onehot_frac_1"] > 0.9
    assert st["adv_onehot"]["std_clamped_frac_0"] > 0.3 and st["adv_onehot"]["std_clamped_frac_1"] > 0.5
    assert min(st["adv_clamp"][k] for k in ("ray_clamp_frac_0", "ray_clamp_frac_1", "dv1_clamp_frac")) > 0.3
    assert st["adv_behind"]["warp_pz_le_eps_frac_per_view"][2] > 0.3
    assert min(st["adv_behind"]["render_pz_le_eps_frac_0"], st["adv_behind"]["render_pz_le_eps_frac_1"]) > 0.1
    assert st["adv_sigma"]["alpha_one_frac_0"] > 0.01 and st["adv_sigma"]["sigma_max_1"] > 104.0


@pytest.mark.parametrize("case", list(ADV_CASES))
def test_oracle_matches_reference_on_adversarial_inputs(case):
    cfg, gold = adv_config(case), load_golden(case)
    sd = tweak_weights(load_weights(), case)
    with torch.no_grad():
        out = O.forward(cfg, sd, _tbatch(case))
    _compare(case, out, gold, 5e-5)


def _net(cfg, case, lib=None, dev=None):
    from enerf_amd.network import Network
    net = Network(cfg, lib=lib) if lib is not None else Network(cfg)
    net.load_state_dict(tweak_weights(load_weights(), case), strict=False)
    return (net.to(dev) if dev is not None else net).eval()


def _check(case, out, gold):
    return _compare(case, out, gold, REL_TOL)


@pytest.mark.parametrize("case", list(ADV_CASES))
def test_emulated_kernels_on_adversarial_inputs(case):
    from emu_lib import emu_lib
    cfg, gold = adv_config(case), load_golden(case)
    out = _net(cfg, case, lib=emu_lib())(_tbatch(case))
    _check(case, out, gold)


def test_bf16x3_fast_mode_is_why_it_is_opt_in():
    """enerf_options_t.render_precision = 2 (the render MLP's dense layers as bf16x3 on the bf16 matrix cores: kernel 192 -> 128 us)
    carries a ~1e-5 operand error.  On ordinary inputs that is far inside the parity bar; where a head has a large gain (adv_sigma:
    the density head scaled x150 / x400) the error is amplified past it — which is why the exact fp32 kernel stays the default."""
    from emu_lib import emu_lib
    from enerf_amd.lib import Options
    worst = {}
    for case in ("adv_white", "adv_sigma"):
        cfg, gold = adv_config(case), load_golden(case)
        net = _net(cfg, case, lib=emu_lib())
        net.options = Options(render_precision=2)
        out = net(_tbatch(case))
        worst[case] = max(_rel(out[k].numpy(), gold["out/" + k]) for k in ("rgb_level1", "weights_level1", "depth_level1"))
    assert worst["adv_white"] < 2e-5, worst                       # ordinary regime: ~7e-6
    assert worst["adv_sigma"] > REL_TOL, worst                    # amplified: measured 1.3e-3 (the exact default: 3.6e-5, above)


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("backend", ["hip", "torch"])
@pytest.mark.parametrize("case", list(ADV_CASES))
def test_gpu_on_adversarial_inputs(case, backend):
    from enerf_amd.network import Network
    dev = torch.device("cuda:0")
    cfg, gold = adv_config(case), load_golden(case)
    net = Network(cfg, feature_backend=backend)
    net.load_state_dict(tweak_weights(load_weights(), case), strict=False)
    out = net.to(dev).eval()(_tbatch(case, dev))
    torch.cuda.synchronize()
    w = _check(case, out, gold)
    print(case, backend, {k: f"{v:.1e}" for k, v in w.items()})
