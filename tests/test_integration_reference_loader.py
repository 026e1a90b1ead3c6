"""The shipped binding modules (integration/lib/networks/enerf/network[_human]_amd.py) loaded through the reference's OWN
loader: ``lib.networks.make_network.make_network(cfg)`` (make_network.py:5-9) with the reference's yacs ``cfg``
(``EnerfConfig.from_yacs``), a reference-saved ``state_dict`` loaded with ``strict=True`` (net_utils.py:443), one forward
on the CPU lane emulator compared with the reference network's own forward on the same batch.

Build-container test: needs /root/reference (skipped elsewhere, e.g. on the GPU box).  One subprocess per case because the
reference's cfg is an import-time global."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REF = os.path.isdir("/root/reference/lib/networks/enerf")

_SCRIPT = r'''
import os, sys
ROOT, case = sys.argv[1], sys.argv[2]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle.ref_loader import load_reference
human = case == "human"
module = "lib.networks.enerf.network_human_amd" if human else "lib.networks.enerf.network_amd"
cfg_file = "configs/enerf/zjumocap_eval.yaml" if human else "configs/enerf/dtu_pretrain.yaml"
opts = ["network_module", module] + ([] if human else ["enerf.cas_config.volume_planes", "8,8"])
cfg, ref_network = load_reference(cfg_file, opts)
assert cfg.network_path == module.replace(".", "/") + ".py"          # config.py:166-168 derived it
cfg.network_path = os.path.join(ROOT, "integration", cfg.network_path)  # the file a maintainer copies into lib/networks/enerf/
os.chdir("/root/reference")
from lib.networks.make_network import make_network
net = make_network(cfg)                                                # imp.load_source(...).Network()
from enerf_amd.network import Network as Amd, NetworkHuman as AmdH
assert isinstance(net, AmdH if human else Amd) and net.human == human
assert tuple(net.cfg.cas.volume_planes) == tuple(cfg.enerf.cas_config.volume_planes)
if human:
    from lib.networks.enerf import network_human as ref_network
torch.manual_seed(0)
ref = ref_network.Network().eval()
sd = ref.state_dict()                                                   # what net_utils.save_model stores under 'net'
g = torch.Generator().manual_seed(3)
for k, v in sd.items():
    if k.endswith("running_mean"): v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    elif k.endswith("running_var"): v.copy_(torch.rand(v.shape, generator=g) + 0.5)
missing = net.load_state_dict(sd, strict=True)                         # net_utils.py:443
net.eval()
from emu_lib import emu_lib
net._lib = emu_lib()
from enerf_amd.synth import make_batch, make_zju_batch
b = make_zju_batch(32, 32, 2, net.cfg, seed=1) if human else make_batch(32, 64, 3, net.cfg, seed=1, textured=True)
batch = {k: torch.from_numpy(v) for k, v in b.items()}
torch.set_num_threads(1)
with torch.no_grad():
    want = ref(batch)
got = net(batch)
assert sorted(got) == sorted(want), (sorted(got), sorted(want))
worst = 0.0
for k in want:
    assert got[k].shape == want[k].shape, k
    err = float((got[k] - want[k]).abs().max() / (want[k].abs().max() + 1e-12))
    worst = max(worst, err)
    assert err < 1e-4, (k, err)
print("INTEGRATION_OK", case, len(sd), worst)
'''


@pytest.mark.skipif(not HAVE_REF, reason="reference tree only exists in the build container")
@pytest.mark.parametrize("case", ["dtu", "human"])
def test_binding_module_through_reference_make_network(case, tmp_path):
    script = tmp_path / "run_integration.py"
    script.write_text(_SCRIPT)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, str(script), ROOT, case], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "INTEGRATION_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
