"""Build the CPU-emulated twin of libenerf_hip.so (tests only; see tests/emu/hip_emu.h)."""
from __future__ import annotations

import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "enerf_amd", "csrc")
OUT = os.path.join(HERE, "_build")
SOURCES = ["geometry.hip", "volume.hip", "conv3d.hip", "conv3d_pk8.hip", "conv3d_b4.hip", "conv3d_s2.hip", "conv3d_t2.hip", "conv3d_wl.hip", "conv2d.hip", "render.hip", "io.hip", "frame.hip", "backward.hip", "wgrad.hip", "train.hip", "train_glue.hip", "mlp_train.hip", "gather.hip", "selftest.hip", "capi.hip"]


def _digest() -> str:
    h = hashlib.sha1()
    srcs = [f for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    for f in srcs + ["../../tests/emu/hip_emu.h", "../../include/enerf_hip.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(SOURCES).encode())
    return h.hexdigest()[:16]


def build(verbose: bool = False) -> str:
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, f"libenerf_emu_{_digest()}.so")
    if os.path.exists(so):
        return so
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(OUT, s.replace(".hip", ".emu.o"))
        cmd = ["g++", "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-DENERF_EMU", "-I", HERE, "-I", CSRC,
               "-include", os.path.join(HERE, "hip_emu.h"), "-Wno-unused-variable",
               "-ffp-contract=off", "-c", os.path.join(CSRC, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"emu build failed for {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    for f in os.listdir(OUT):
        if f.startswith("libenerf_emu_") and f.endswith(".so"):
            os.remove(os.path.join(OUT, f))
    subprocess.run(["g++", "-shared", "-o", so] + objs + ["-lpthread"], check=True)
    return so


if __name__ == "__main__":
    print(build(verbose=True))
