#!/bin/bash
# per-kernel time of the config-5 training step (graph replays).  usage: gpu_train_stats.sh TAG [extra bench flags]
export TMPDIR=/tmp
TAG=${1:-r03_train_ks}; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o p -- python $R/bench.py --train --steps 10 --warmup 2 "$@" > $O/bench.log 2>&1
cp $(find /tmp/pt -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
tail -1 $O/bench.log | cut -c1-300
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/train_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:45]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}")
PY
