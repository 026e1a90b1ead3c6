"""Summarise hipcc -Rpass-analysis=kernel-resource-usage for the kernels in enerf_amd/csrc (dev tool)."""
import re, subprocess, sys, os
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "enerf_amd", "csrc")
files = sys.argv[1:] or ["geometry", "volume", "conv3d", "render"]
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize",
                          "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, f + ".hip"), "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m: continue
        t = m.group(1).strip()
        if t.startswith("Function Name:") or t.startswith("Name:"):
            if cur: print(cur)
            name = subprocess.run(["c++filt", t.split(":",1)[1].strip()], capture_output=True, text=True).stdout.strip()
            cur = {"kernel": re.sub(r"\(.*", "", name)}
        else:
            k, _, v = t.rpartition(":")
            k = k.strip().split(" [")[0]
            if k in ("VGPRs", "AGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize", "Occupancy", "LDS Size", "TotalSGPRs"):
                cur[k] = v.strip()
    if cur: print(cur)
