#!/bin/bash
# the in-tree training step: ms per step (graph replays) MSE-only x REPS and once with the perceptual term, then the training GPU tests
#   usage: gpu_r06_train_now.sh TAG REPS [notest]
export TMPDIR=/tmp
TAG=$1; REPS=$2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for rep in $(seq 1 $REPS); do
  timeout 300 python bench.py --train --no-perceptual --steps 40 --warmup 5 --no-stages --no-cpu-baseline --no-live-pmc > $O/t_$rep.json 2>$O/t_$rep.err
  echo "train #$rep: $(python -c "import json; d=json.loads(open('$O/t_$rep.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step', d['config']['step_launch'][:60], 'loss', d['final_loss'])" 2>&1)"
done
timeout 300 python bench.py --train --steps 20 --warmup 5 --no-stages --no-cpu-baseline --no-live-pmc > $O/p.json 2>$O/p.err
echo "train+perceptual: $(python -c "import json; d=json.loads(open('$O/p.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step', d['config']['step_launch'][:60], 'loss', d['final_loss'])" 2>&1)"
[ "$3" = notest ] || timeout 1500 python -m pytest tests/test_training.py tests/test_capi_symbols.py tests/test_capi_errors.py -m gpu -x -q 2>&1 | tail -5
