export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; R=${1:-r05_run8}; O=gpurun_out/$R; mkdir -p $O
timeout 900 python -m pytest tests/test_training.py tests/test_train_glue.py -m gpu -x -q 2>&1 | tail -3
python bench.py --train --no-perceptual --steps 20 --warmup 3 --no-live-pmc > $O/train_noperc.json 2> $O/train_noperc.err; python -c "
import json; t=json.loads(open('$O/train_noperc.json').read().strip().splitlines()[-1]); print('train noperc', round(t['ms_per_step'],3), 'ms/step')"
bash tools/gpu_train_stats.sh $R/train --no-perceptual > $O/train_stats.log 2>&1; grep -E "launches per step|library kernels" $O/train_stats.log
