#!/bin/bash
# One gpurun call for kernel A/B work: selected GPU tests, the per-layer conv3d microbench, frame-level option A/B.
#   usage: bash tools/gpu_ab.sh TAG "pytest -k expr" "name=field:val,... (ab_options variants)"
export TMPDIR=/tmp
TAG=${1:-r03_ab}; KEXPR=${2:-full_size_dtu}; VARS=${3:-}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 600 python tools/bench_conv3d_layers.py 50 > $O/layers.txt 2>&1; cat $O/layers.txt
if [ -n "$VARS" ]; then timeout 900 python tools/ab_options.py dtu $VARS > $O/ab_options.txt 2>&1; cat $O/ab_options.txt; fi
