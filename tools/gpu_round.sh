#!/bin/bash
# One gpurun call: GPU parity tests, smoke, the three bench workloads, profiles.  usage: bash tools/gpu_round.sh TAG
export TMPDIR=/tmp
TAG=${1:-r03_run1}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu_$TAG.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke_$TAG.log
bash tools/collect_profiles.sh ${TAG}_dtu dtu 1
bash tools/collect_profiles.sh ${TAG}_lego lego 1
bash tools/collect_profiles.sh ${TAG}_zju zju 1
