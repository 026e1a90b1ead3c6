#!/bin/bash
# per-frame-sync latency of the default frame vs whole-frame hipGraph replays (bench.py --graph).  usage (gpurun): bash tools/ab_graph.sh
cd $GRAFT_REPO_ROOT
for wl in dtu lego zju; do
for cfg in "" "--graph" "--single-stream" "--graph --single-stream"; do
    python bench.py --workload $wl --no-cpu-baseline --no-stages --steps 400 --warmup 50 $cfg 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', '$cfg', round(d['value'],1), round(d['ms_per_step'],4))"
done
done
