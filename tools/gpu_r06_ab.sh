#!/bin/bash
# round-6 kernel A/B in one gpurun call: selected GPU tests on the in-tree library, then per library variant (enerf_amd/_ab/lib_*.so,
# tools/build_variant.py; "base" = the in-tree library) the per-layer conv3d microbench, and the frame-level option A/B on dtu.
#   usage: bash tools/gpu_r06_ab.sh TAG "pytest -k expr" "ab_options variants" v1 v2 ...
export TMPDIR=/tmp
TAG=$1; KEXPR=$2; VARS=$3; shift 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
cp enerf_amd/libenerf_hip.so /tmp/lib_keep.so
for v in base "$@"; do
  [ "$v" != base ] && cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  echo "== layers, library $v"; timeout 600 python tools/bench_conv3d_layers.py 50 2>&1 | grep -v amdgpu.ids | tee $O/layers_$v.txt
  cp /tmp/lib_keep.so $R/enerf_amd/libenerf_hip.so
done
if [ -n "$VARS" ]; then timeout 900 python tools/ab_options.py dtu $VARS 2>&1 | grep -v amdgpu.ids | tee $O/ab_options.txt; fi
