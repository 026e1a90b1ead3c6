#!/bin/bash
# the training step with the weight-gradient kernels on a second stream (autograd.WgradLane) against the single-stream step:
# ms per step (graph replays), interleaved repeats, MSE-only and with the perceptual term; then the training GPU tests.
#   usage: gpu_r06_lane_ab.sh TAG REPS [notest] [what ...]      what: comma lists of conv (cost-reg nets), feat (FeatureNet), mlp
export TMPDIR=/tmp
TAG=$1; REPS=$2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
TEST=$3; shift 3
for rep in $(seq 1 $REPS); do
  for what in "$@"; do
    ENERF_WGRAD_LANE=1 ENERF_WGRAD_LANE_WHAT=$what timeout 300 python bench.py --train --no-perceptual --steps 40 --warmup 5 --no-stages --no-cpu-baseline --no-live-pmc > $O/t_${what}_$rep.json 2>$O/t_${what}_$rep.err
    echo "train lane what=$what #$rep: $(python -c "import json; d=json.loads(open('$O/t_${what}_$rep.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step', d['config']['step_launch'][:60], 'loss', d['final_loss'])" 2>&1)"
  done
  for lane in 0; do
    ENERF_WGRAD_LANE=$lane timeout 300 python bench.py --train --no-perceptual --steps 40 --warmup 5 --no-stages --no-cpu-baseline --no-live-pmc > $O/t_lane${lane}_$rep.json 2>$O/t_lane${lane}_$rep.err
    echo "train lane=$lane #$rep: $(python -c "import json; d=json.loads(open('$O/t_lane${lane}_$rep.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step', d['config']['step_launch'][:60], 'loss', d['final_loss'])" 2>&1)"
  done
done
for lane in; do
  ENERF_WGRAD_LANE=$lane timeout 300 python bench.py --train --steps 20 --warmup 5 --no-stages --no-cpu-baseline --no-live-pmc > $O/p_lane${lane}.json 2>$O/p_lane${lane}.err
  echo "train+perceptual lane=$lane: $(python -c "import json; d=json.loads(open('$O/p_lane${lane}.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step', d['config']['step_launch'][:60], 'loss', d['final_loss'])" 2>&1)"
done
[ "$TEST" = notest ] || timeout 1500 python -m pytest tests/test_training.py -m gpu -x -q 2>&1 | tail -5
