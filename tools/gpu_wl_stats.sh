#!/bin/bash
# per-kernel durations (single stream) + default-mode frame rate for several workloads.  usage: gpu_wl_stats.sh TAG "grep-pattern" wl1 wl2 ...
export TMPDIR=/tmp
TAG=$1; PAT=$2; shift 2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
for wl in "$@"; do
  cd $R; python bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline --no-stages > $O/bench_$wl.json 2>$O/bench_$wl.err
  python -c "import json; d=json.loads(open('$O/bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', round(d['value'],1), 'frames/s')"
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw_$wl -o p -- python $R/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-stages --no-sync-per-frame --single-stream > $O/prof_$wl.log 2>&1
  f=$(find /tmp/pw_$wl -name "*kernel_stats.csv" | head -1); cp "$f" $O/stats_$wl.csv
  grep -E "$PAT" $O/stats_$wl.csv | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   ', r[0][:70], 'avg us', round(float(r[3])/1e3,1))"
done
