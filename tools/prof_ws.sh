#!/bin/bash
# per-kernel conv3d durations for a list of "name:ENV=.. ENV=.." configurations (one box, one call)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ab_ws; mkdir -p $O
cd /tmp
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}; [ "$name" = "$cfg" ] && envs=""
  env $envs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stages > $O/prof_$name.log 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/stats_$name.csv
  echo "== $name"; python - "$O/stats_$name.csv" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Name"]
    if "s1_" in k:
        short = re.sub(r"\(.*", "", k).replace("void enerf::", "")
        print(f"  {short:28s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:7.1f} min={float(r['MinNs'])/1e3:7.1f} max={float(r['MaxNs'])/1e3:7.1f}")
PY
done
