#!/bin/bash
# Round-5 evidence in ONE gpurun call: the driver's bench command (with the lego / zju / train workloads inside), pytest -m gpu, smoke,
# bench + rocprofv3 kernel stats + PMC per kernel for dtu / lego / zju, both training lines, the per-step training kernel table.
# usage: bash tools/gpu_round5_full.sh TAG
export TMPDIR=/tmp
TAG=${1:-r05_run2}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/$TAG
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
timeout 600 python bench.py --train --steps 20 --warmup 3 > $O/$TAG/train_bench.json 2> $O/$TAG/train_bench.err; tail -c 300 $O/$TAG/train_bench.json; echo
timeout 600 python bench.py --train --no-perceptual --steps 20 --warmup 3 > $O/$TAG/train_bench_noperc.json 2> $O/$TAG/train_bench_noperc.err; tail -c 300 $O/$TAG/train_bench_noperc.json; echo
bash tools/gpu_train_stats.sh $TAG/train --no-perceptual > $O/$TAG/train_stats.log 2>&1; head -4 $O/$TAG/train_stats.log
