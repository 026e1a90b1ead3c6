#!/bin/bash
# A/B of runtime environment knobs (signal wait mode, kernarg placement) on the per-frame-sync latency.  usage (gpurun): bash tools/ab_env.sh
cd $GRAFT_REPO_ROOT
for cfg in "" "HSA_ENABLE_INTERRUPT=0" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0 HIP_FORCE_DEV_KERNARG=1"; do
  for rep in 1 2; do
    env $cfg python bench.py --no-cpu-baseline --no-stages --steps 400 --warmup 50 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'rep$rep', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
