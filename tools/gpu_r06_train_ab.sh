#!/bin/bash
# training step under library variants: ms per step (graph replays, MSE-only), interleaved repeats; then the training GPU tests on the in-tree library.
#   usage: gpu_r06_train_ab.sh TAG REPS v1 v2 ...   ("base" = in-tree)
export TMPDIR=/tmp
TAG=$1; REPS=$2; shift 2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; cp enerf_amd/libenerf_hip.so /tmp/lib_keep.so
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    [ "$v" != base ] && cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
    timeout 300 python bench.py --train --no-perceptual --steps 40 --warmup 5 --no-stages --no-cpu-baseline --no-live-pmc > $O/t_${v}_$rep.json 2>/dev/null
    echo "train $v #$rep: $(python -c "import json; d=json.loads(open('$O/t_${v}_$rep.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step', d['config']['step_launch'][:40], 'loss', d['final_loss'])" 2>&1)"
    cp /tmp/lib_keep.so $R/enerf_amd/libenerf_hip.so
  done
done
[ -n "$NOTEST" ] || timeout 1500 python -m pytest tests/test_training.py -m gpu -x -q 2>&1 | tail -3
