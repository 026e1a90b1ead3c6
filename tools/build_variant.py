"""Dev tool: build a named variant of libenerf_hip.so into enerf_amd/_ab/ with extra hipcc flags
(for same-box A/B runs: tools/ab_variants.sh copies each variant over enerf_amd/libenerf_hip.so in turn).

    python tools/build_variant.py NAME [--all FLAG ...] [--file render.hip FLAG ...]
"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enerf_amd import build as B

def main():
    name = sys.argv[1]
    all_flags, per_file, cur = [], {}, None
    args = sys.argv[2:]
    i = 0
    while i < len(args):
        if args[i] == "--all": cur = all_flags
        elif args[i] == "--file": i += 1; cur = per_file.setdefault(args[i], [])
        else: cur.append(args[i])
        i += 1
    outdir = os.path.join(B.PKG, "_ab"); objdir = os.path.join(outdir, "obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    hipcc = B._hipcc(); procs = []; objs = []
    for s in B.SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        cmd = [hipcc, *B.FLAGS, *all_flags, *per_file.get(s, []), "-I", B.CSRC, "-c", os.path.join(B.CSRC, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))); objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode: raise SystemExit(f"{s}: {out}")
    lib = os.path.join(outdir, f"lib_{name}.so")
    subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", lib, *objs], check=True)
    subprocess.run(["rm", "-rf", objdir]); print(lib)

if __name__ == "__main__":
    main()
