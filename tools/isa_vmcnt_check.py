"""Build-time check of hand-counted `s_waitcnt vmcnt(N)` waits (conv2d.hip k_smooth1_fused).

`vmem_wait_pending<N>()` behind an LDS-DMA copy means "the copy has landed" only if EXACTLY N vector-memory loads were issued
after the copy and before the wait — hipcc may merge, hoist or scalarise loads on a toolchain bump, and the CPU lane emulator
compiles both sides to no-ops.  This tool compiles a source to gfx950 assembly and, for every `s_waitcnt vmcnt(N)` with N >= 8
inside the named kernel, counts the vector-memory loads between the last `global_load_lds_*` in front of it and the wait.

    python tools/isa_vmcnt_check.py conv2d.hip k_smooth1_fused        (exit code 1 on a mismatch)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernel_asm(source: str, kernel_substr: str) -> list:
    from enerf_amd import build as B
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([B._hipcc(), *B.FLAGS, "-I", B.CSRC, "-S", "--cuda-device-only", "-o", out, os.path.join(B.CSRC, source)],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    body, inside = [], False
    for l in lines:
        if re.match(r"^_Z\w*" + re.escape(kernel_substr) + r"\w*:", l):
            inside = True
            continue
        if inside:
            if "s_endpgm" in l:
                break
            body.append(l.strip())
    if not body:
        raise SystemExit(f"kernel *{kernel_substr}* not found in {source}")
    return body


def check(source: str, kernel_substr: str, min_n: int = 8) -> list:
    """[(N of the wait, vector-memory loads counted behind the last LDS-DMA copy)] for every counted wait of the kernel."""
    body = kernel_asm(source, kernel_substr)
    res = []
    for i, l in enumerate(body):
        m = re.match(r"s_waitcnt\s+vmcnt\((\d+)\)", l)
        if not m or int(m.group(1)) < min_n:
            continue
        n, loads, found = int(m.group(1)), 0, False
        for k in range(i - 1, -1, -1):
            op = body[k].split()[0] if body[k] else ""
            if op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in body[k]):
                found = True
                break
            if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                loads += 1
        res.append((n, loads if found else None))
    return res


if __name__ == "__main__":
    r = check(sys.argv[1], sys.argv[2])
    print(r)
    sys.exit(0 if r and all(n == c for n, c in r) else 1)
