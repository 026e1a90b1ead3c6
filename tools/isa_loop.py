"""Dev tool: instruction-class histogram of the innermost loop(s) of one kernel's gfx950 ISA.
    python tools/isa_loop.py render.hip k_render_raysILi3ELi3ELi4ELi2 [extra hipcc flags...]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
out = "/tmp/_isa_loop.s"
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize", *extra, "-S",
                    "--cuda-device-only", "-I", os.path.join(ROOT, "enerf_amd/csrc"), os.path.join(ROOT, "enerf_amd/csrc", src), "-o", out],
                   capture_output=True, text=True)
if r.returncode: sys.exit(r.stderr)
L = open(out).read().splitlines()
s = [i for i, l in enumerate(L) if l.startswith("_Z") and pat in l and l.split(";")[0].strip().endswith(":")][0]
e = [i for i in range(s, len(L)) if "s_endpgm" in L[i]][0]
body = L[s:e]
def cls(op):
    return ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
            "vmem" if op.startswith(("global_", "buffer_", "scratch_")) else "nop" if op == "s_nop" else "wait" if op == "s_waitcnt" else "salu")
# blocks that belong to a Depth=2 loop (the sample loop of the render kernel)
depth2 = [i for i, l in enumerate(body) if "Depth=2" in l]
if not depth2: sys.exit("no depth-2 loop")
lo, hi = depth2[0], depth2[-1]
# extend to the end of the last depth-2 block
j = hi + 1
while j < len(body) and not (body[j].startswith(".LBB") or body[j].lstrip().startswith("; %bb.")): j += 1
c = collections.Counter(); ops = collections.Counter()
for l in body[lo:j]:
    l = l.strip()
    if not l or l[0] in ";." or l.split(";")[0].strip().endswith(":"): continue
    op = l.split()[0]; ops[op] += 1; c[cls(op)] += 1
    if op.startswith("scratch_"): c["scratch"] += 1
print("depth-2 loop body:", dict(c))
print(ops.most_common(int(os.environ.get("TOP", "30"))))
open("/tmp/_isa_loop_body.s", "w").write("\n".join(body[lo:j]))
