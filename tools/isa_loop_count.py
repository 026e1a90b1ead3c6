"""Dev tool: instruction-class histogram of the LOOPS of one kernel (spans closed by a backward branch), from /tmp/_isa_count.s
(run tools/isa_count.py on the source first).   python tools/isa_loop_count.py <mangled-name-substring>"""
import collections, re, sys
L = open("/tmp/_isa_count.s").read().splitlines()
pat = sys.argv[1]
s = [i for i, l in enumerate(L) if l.startswith("_Z") and pat in l and l.rstrip().split(";")[0].strip().endswith(":")][0]
e = [i for i in range(s, len(L)) if "s_endpgm" in L[i]][0]
labels = {}
for i in range(s, e):
    m = re.match(r"^(\.LBB\d+_\d+):", L[i])
    if m: labels[m.group(1)] = i
def cls(op):
    return ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
            "vmem" if op.startswith(("global_", "buffer_", "scratch_")) else "wait" if op == "s_waitcnt" else "nop" if op == "s_nop" else "salu")
for i in range(s, e):
    t = L[i].strip().split(";")[0].split()
    if len(t) >= 2 and t[0].startswith(("s_cbranch", "s_branch")) and t[1] in labels and labels[t[1]] < i:
        c = collections.Counter(); ops = collections.Counter()
        for l in L[labels[t[1]]:i + 1]:
            l = l.strip()
            if not l or l[0] in ";." or l.split(";")[0].strip().endswith(":"): continue
            op = l.split()[0]; c[cls(op)] += 1
            if cls(op) == "valu": ops[op] += 1
        if c["mfma"] == 0: continue                       # (only the loops that feed the matrix pipe)
        print(f"loop {t[1]} .. line {i - s}: {dict(c)}")
        print("   ", ops.most_common(8))
