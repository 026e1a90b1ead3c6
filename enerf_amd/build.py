"""Compile the HIP sources in enerf_amd/csrc into enerf_amd/libenerf_hip.so for gfx950 (in-tree)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")
SOURCES = ["geometry.hip", "volume.hip", "conv3d.hip", "conv3d_pk8.hip", "conv3d_b4.hip", "conv3d_s2.hip", "conv3d_t2.hip", "conv3d_wl.hip", "conv2d.hip", "render.hip", "io.hip", "frame.hip", "backward.hip", "wgrad.hip", "train.hip", "train_glue.hip", "mlp_train.hip", "gather.hip", "selftest.hip", "capi.hip"]
LIB = os.path.join(PKG, "libenerf_hip.so")
STAMP = os.path.join(PKG, "csrc", ".build_stamp")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wno-unused-variable",
         "-ffp-contract=fast",
         # packed-f32 VALU (v_pk_mul/add/fma_f32) next to MFMAs is slower than the scalar forms on CDNA4;
         # measured on MI355X: render kernel 227 -> 220 us, whole frame +2 % with SLP vectorisation off
         "-fno-slp-vectorize"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC)")


def source_digest() -> str:
    h = hashlib.sha1()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(INCLUDE, "enerf_hip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False, extra_flags: list[str] | None = None) -> str:
    digest = source_digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == digest:
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(PKG, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for s in SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *(extra_flags or []), "-I", CSRC, "-c", os.path.join(CSRC, s), "-o", o]
        procs.append((s, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    for s, cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{' '.join(cmd)}\n{out}")
        if verbose and out.strip():
            print(out)
    subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs], check=True)
    with open(STAMP, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
