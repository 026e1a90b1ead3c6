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
    ap.add_argument("--train", action="store_true",
                    help="time ONE training step (forward in .train(), MSE loss, backward) of BASELINE config 5 instead of eval frames: "
                         "the reference's modules vs oracle.train_step (what bench.py --train's cpu_baseline times on the GPU box)")
    a = ap.parse_args()
    if a.train:
        return train_main(a)
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


def train_main(a) -> None:
    import numpy as np
    from oracle.ref_loader import load_reference
    from oracle.make_golden import seeded_state_dict
    from oracle import enerf_oracle as O
    from enerf_amd.config import EnerfConfig
    from enerf_amd.synth import make_batch
    cfg, ref_network = load_reference("configs/enerf/dtu_pretrain.yaml", [])      # planes 64,8, render_if True,True (bench.py --train)
    torch.manual_seed(0)
    torch.set_num_threads(a.threads)
    net = ref_network.Network().train()
    sd = seeded_state_dict(net)
    net.load_state_dict(sd)
    ecfg = EnerfConfig.from_yacs(cfg)
    b = make_batch(512, 640, 3, ecfg, seed=0, textured=True)
    rng = np.random.default_rng(0)
    for i in range(2):
        b[f"rgb_{i}"] = rng.uniform(0, 1, size=(1, b[f"rays_{i}"].shape[1], 3)).astype(np.float32)
    batch = {k: torch.from_numpy(v) for k, v in b.items()}
    sd_cpu = {k: v.detach().clone() for k, v in sd.items()}
    t0 = time.perf_counter()
    out = net(batch)
    loss = sum(w * torch.nn.functional.mse_loss(batch[f"rgb_{i}"], out[f"rgb_level{i}"]) for i, w in enumerate((0.1, 1.0)))
    loss.backward()
    t_ref = time.perf_counter() - t0
    g_ref = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    t0 = time.perf_counter()
    loss_p, g_port = O.train_step(ecfg, sd_cpu, batch)
    t_port = time.perf_counter() - t0
    worst = max(float((g_port[k] - g_ref[k]).abs().max() / (g_ref[k].abs().max() + 1e-30)) for k in g_ref if k in g_port)
    print(f"training step: dtu_pretrain 512x640, 3 source views, volume_planes 64,8, full-image rays at both levels, MSE loss; "
          f"torch {torch.__version__}; {a.threads} threads of {os.cpu_count()}; one step each, no warm-up")
    print(f"reference modules (.train(), forward + loss + backward): {t_ref:.2f} s/step = {1 / t_ref:.4f} samples/s; loss {float(loss):.6f}")
    print(f"oracle port (oracle/enerf_oracle.py::train_step)       : {t_port:.2f} s/step = {1 / t_port:.4f} samples/s; loss {float(loss_p):.6f}")
    print(f"port / reference time: {t_port / t_ref:.3f}; {len(g_port)} parameter gradients, worst max|port - ref| / max|ref| = {worst:.2e}")


if __name__ == "__main__":
    main()
