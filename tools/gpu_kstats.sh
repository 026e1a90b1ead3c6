#!/bin/bash
# per-kernel durations (kernels alone: --single-stream) of the dtu frame under option variants.  usage: gpu_kstats.sh TAG "opts1" "opts2" ...
export TMPDIR=/tmp
TAG=${1:-r03_ks}; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp
i=0
for OPTS in "$@"; do
  i=$((i+1)); X=""; [ "$OPTS" != "none" ] && X="--options $OPTS"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_$i -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stages --no-sync-per-frame --single-stream $X > $O/stats_$i.log 2>&1
  cp $(find /tmp/pk_$i -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$i.csv
  echo "== variant $i: $OPTS"
  python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_$i.csv")))
tot=0
for r in rows:
    if "enerf" in r["Name"]:
        n=r["Name"].replace("void enerf::","").replace("enerf::","").split("(")[0]
        if "conv3d" in n or "volume" in n or "depth" in n:
            print(f"{n:46s} n/frame {int(r['Calls'])/25:4.1f} avg {float(r['AverageNs'])/1e3:7.1f} min {float(r['MinNs'])/1e3:7.1f}")
PY
done
