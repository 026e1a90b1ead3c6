#!/bin/bash
# Round-6 evidence in ONE gpurun call: the driver's bench command (lego / zju / train / train_perceptual inside), pytest -m gpu, smoke,
# bench + rocprofv3 kernel stats + PMC per kernel for dtu / lego / zju, the per-step training kernel table.   usage: bash tools/gpu_round6_full.sh TAG
export TMPDIR=/tmp
TAG=${1:-r06_run2}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/$TAG
cd $R
T0=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/$TAG/bench_default.json 2> $O/$TAG/bench_default.err; echo "bench rc=$?"
echo "bench wall $(python -c "import time; print(round(time.time() - $T0, 1))") s"
timeout 1800 python -m pytest tests -m gpu -q > $O/$TAG/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/$TAG/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/$TAG/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/$TAG/smoke.log
bash tools/collect_profiles.sh ${TAG}_dtu dtu 1
bash tools/collect_profiles.sh ${TAG}_lego lego 1
bash tools/collect_profiles.sh ${TAG}_zju zju 1
cd $R
bash tools/gpu_train_stats.sh $TAG/train --no-perceptual > $O/$TAG/train_stats.log 2>&1; head -4 $O/$TAG/train_stats.log
bash tools/gpu_frame_gap.sh $TAG/gap > $O/$TAG/frame_gap.txt 2>&1; head -3 $O/$TAG/frame_gap.txt
