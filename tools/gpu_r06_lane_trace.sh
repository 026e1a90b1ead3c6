#!/bin/bash
# kernel trace of the graph-replayed training step with the weight-gradient lane: do the lane's kernels overlap the chain's?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_lane_trace; mkdir -p $O; cd /tmp
ENERF_WGRAD_LANE=${1:-1} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python $R/bench.py --train --no-perceptual --steps 8 --warmup 2 --no-stages --no-cpu-baseline --no-live-pmc > $O/bench.json 2> $O/bench.err
F=$(find /tmp/lt -name '*kernel_trace.csv' | head -1)
python - "$F" <<'P' | tee $O/overlap_lane${1:-1}.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last replay: the last 430-ish kernels; find the last k_mlp_bwd<3, 3> and take a window of one step before the end
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_mlp_bwd<3, 3" in n]
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
queues = sorted({(r["Queue_Id"], r.get("Stream_Id", "")) for r in step})
print("kernels in one step:", len(step), " span us:", (int(step[-1]["End_Timestamp"]) - t0) / 1e3, " queues/streams:", queues)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3
print("sum of kernel durations us:", busy)
ov = 0
last_end = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < last_end: ov += 1
    last_end = max(last_end, e)
print("kernels that start before an earlier one ended:", ov)
for r in step[:60]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} q{r["Queue_Id"]} {r["Kernel_Name"][:70]}')
P
