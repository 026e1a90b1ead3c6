"""Dev tool: instruction-class histogram of one kernel's gfx950 ISA.
    python tools/isa_count.py render.hip k_render_raysILi3ELi3ELi3 [extra hipcc flags...]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
out = "/tmp/_isa_count.s"
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize", *extra, "-S",
                    "--cuda-device-only", "-I", os.path.join(ROOT, "enerf_amd/csrc"), os.path.join(ROOT, "enerf_amd/csrc", src), "-o", out],
                   capture_output=True, text=True)
if r.returncode: sys.exit(r.stderr)
L = open(out).read().splitlines()
s = [i for i, l in enumerate(L) if l.startswith("_Z") and pat in l and l.rstrip().split(";")[0].strip().endswith(":")][0]
e = [i for i in range(s, len(L)) if "s_endpgm" in L[i]][0]
c = collections.Counter(); ops = collections.Counter()
for l in L[s:e]:
    l = l.strip()
    if not l or l[0] in ";." or l.split(";")[0].strip().endswith(":"): continue
    op = l.split()[0]; ops[op] += 1
    k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
         "vmem" if op.startswith(("global_", "buffer_", "scratch_")) else "nop" if op == "s_nop" else "wait" if op == "s_waitcnt" else "salu")
    c[k] += 1
    if op.startswith("scratch_"): c["scratch"] += 1
print(dict(c))
meta = "\n".join(L)
m = re.search(r"\.name:\s+" + re.escape(L[s].split(":")[0]) + r"\b.*?\.vgpr_count:\s+(\d+).*?", meta, re.S)
print(ops.most_common(int(os.environ.get("TOP", "25"))))
