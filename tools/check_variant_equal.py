"""GPU dev check: do two library variants (tools/build_variant.py -> enerf_amd/_ab/lib_<name>.so) produce BIT-IDENTICAL frames?
    python tools/check_variant_equal.py base other [workload ...]      (through gpurun)
Used for the round-5 claims "k_smooth0_cb / k_conv0_fused_cb / k_conv3d_s1_b4c / the render tile deal change no output bit"."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _seeded_network  # noqa: E402
from enerf_amd.lib import EnerfLib  # noqa: E402

sys.argv[0] = "bench"
import bench  # noqa: E402

a, b = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
for wl in (sys.argv[3:] or ["dtu", "zju", "lego"]):
    cfg, batch_np, human, _ = bench.make_workload(wl, 0)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in batch_np.items()}
    outs = []
    for name in (a, b):
        lib = EnerfLib(os.path.join(ROOT, "enerf_amd", "_ab", f"lib_{name}.so"))
        net = _seeded_network(cfg, dev, human=human, lib=lib)
        with torch.no_grad():
            o = net(batch)
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in o.items()})
    worst = {k: float((outs[0][k] - outs[1][k]).abs().max()) for k in outs[0]}
    print(f"{wl}: {a} vs {b}: " + ("BIT-IDENTICAL" if all(torch.equal(outs[0][k], outs[1][k]) for k in outs[0]) else f"DIFFERENT {worst}"), flush=True)
