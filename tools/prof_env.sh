#!/bin/bash
# per-kernel stats for a list of "name:ENV=.. ENV=.." configurations of the CURRENT library (same box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_env; mkdir -p $O; cd /tmp
PAT="$1"; shift
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}; [ "$name" = "$cfg" ] && envs=""
  env $envs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$name -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stages --in-flight 1 > $O/$name.log 2>&1
  f=$(find /tmp/pe_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/stats_$name.csv
  echo "== $name"; python - "$O/stats_$name.csv" "$PAT" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Name"]
    if re.search(sys.argv[2], k):
        short = re.sub(r"\(.*", "", k).replace("void enerf::", "")
        print(f"  {short:34s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:7.1f} min={float(r['MinNs'])/1e3:7.1f} max={float(r['MaxNs'])/1e3:7.1f}")
PY
done
