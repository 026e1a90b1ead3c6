#!/bin/bash
# kernel timeline of the last frame (start/end relative to frame start, queue id) — to see stream overlap
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tr -o p -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-stages --in-flight 1 "$@" > /dev/null 2>&1
python - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/p_tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "enerf" in r["Kernel_Name"]]
# last frame = from the last k_conv2d<4 (conv0.0) on
idx = max(i for i, r in enumerate(rows) if "k_conv0_fused" in r["Kernel_Name"] or "k_conv2d<4" in r["Kernel_Name"])
fr = rows[idx:]
t0 = int(fr[0]["Start_Timestamp"])
for r in fr:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void enerf::", "").replace("enerf::", "")
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:8.1f} {(int(r["End_Timestamp"])-t0)/1e3:8.1f}  q{r["Queue_Id"]}  {n[:44]}')
PY
