#!/bin/bash
# the training step with the weight gradients' second stages deferred to one launch per network backward pass (ENERF_WGRAD_DEFER=1, the
# default) against immediate second stages (=0): ms per step (graph replays, MSE-only), interleaved repeats.   usage: gpu_r06_defer_ab.sh TAG REPS
export TMPDIR=/tmp
TAG=$1; REPS=$2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for rep in $(seq 1 $REPS); do
  for d in 0 1; do
    ENERF_WGRAD_DEFER=$d timeout 300 python bench.py --train --no-perceptual --steps 40 --warmup 5 --no-stages --no-cpu-baseline --no-live-pmc > $O/t_${d}_$rep.json 2>$O/t_${d}_$rep.err
    echo "train defer=$d #$rep: $(python -c "import json; d=json.loads(open('$O/t_${d}_$rep.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step', d['config']['step_launch'][:60], 'loss', d['final_loss'])" 2>&1)"
  done
done
