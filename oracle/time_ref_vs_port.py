"""Time the UNMODIFIED reference modules and the oracle port on the same frame, same thread count (build container only).

TEST INFRASTRUCTURE.  bench.py's ``cpu_baseline`` on the GPU box is the oracle port (``kind: "port"``: /root/reference does
not travel); this script records, once per round, that the port's CPU time is the reference's CPU time, so the port is a
fair stand-in.  Usage (repo root):  python oracle/time_ref_vs_port.py [--threads N] > profiles/rNN_cpu_ref_vs_port.txt
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--render-if", default="False,True")
    a = ap.parse_args()
    from oracle.ref_loader import load_reference
    from oracle.make_golden import seeded_state_dict
    from oracle import enerf_oracle as O
    from enerf_amd.config import EnerfConfig
    from enerf_amd.synth import make_batch
    cfg, ref_network = load_reference("configs/enerf/dtu_pretrain.yaml",
                                      ["enerf.cas_config.volume_planes", "48,8", "enerf.cas_config.render_if", a.render_if])
    torch.manual_seed(0)
    torch.set_num_threads(a.threads)
    net = ref_network.Network().eval()
    sd = seeded_state_dict(net)
    net.load_state_dict(sd)
    ecfg = EnerfConfig.from_yacs(cfg)
    batch = {k: torch.from_numpy(v) for k, v in make_batch(512, 640, 3, ecfg, seed=0, textured=True).items()}
    sd_cpu = {k: v.detach() for k, v in sd.items()}

    def timed(fn):
        with torch.no_grad():
            fn()                                   # warm-up
            t0 = time.perf_counter()
            for _ in range(a.frames):
                out = fn()
            return (time.perf_counter() - t0) / a.frames, out
    t_ref, o_ref = timed(lambda: net(batch))
    t_port, o_port = timed(lambda: O.forward(ecfg, sd_cpu, batch))
    err = max(float((o_ref[k] - o_port[k]).abs().max() / (o_ref[k].abs().max() + 1e-12)) for k in o_ref)
    print(f"frame: DTU 512x640, 3 source views, volume_planes 48,8, render_if {a.render_if}; torch {torch.__version__}; "
          f"{a.threads} threads of {os.cpu_count()}; 1 warm-up + {a.frames} timed frames")
    print(f"reference modules (lib/networks/enerf/network.py, unmodified): {t_ref:.3f} s/frame = {1 / t_ref:.4f} frames/s")
    print(f"oracle port       (oracle/enerf_oracle.py)                   : {t_port:.3f} s/frame = {1 / t_port:.4f} frames/s")
    print(f"port / reference time: {t_port / t_ref:.3f}; max relative output difference {err:.2e}")


if __name__ == "__main__":
    main()
