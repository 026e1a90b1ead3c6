#!/bin/bash
# round 6: runtime environment knobs that could shorten the host side of the per-frame-sync latency (the library itself reads no environment)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "" "ROC_ACTIVE_WAIT_TIMEOUT=5000" "HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=5000" "GPU_MAX_HW_QUEUES=2" "AMD_DIRECT_DISPATCH=0" "HIP_FORCE_DEV_KERNARG=1 ROC_ACTIVE_WAIT_TIMEOUT=5000"; do
    env $cfg python bench.py --no-cpu-baseline --no-stages --no-secondary --steps 300 --warmup 50 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg]', 'rep$rep', round(d['value'],1), round(d['ms_per_step'],4))"
done
done
