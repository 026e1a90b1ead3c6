#!/bin/bash
# parity tests of one library variant, then tools/ab_variants.sh over all.  usage: gpu_variants.sh TAG test_variant base v1 v2 ...
export TMPDIR=/tmp
TAG=$1; TV=$2; shift 2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
cp enerf_amd/libenerf_hip.so /tmp/lib_keep.so
cp enerf_amd/_ab/lib_$TV.so enerf_amd/libenerf_hip.so
timeout 900 python -m pytest tests -m gpu -x -q -k "full_size_dtu or goldens or against_oracle_medium" > $O/pytest_$TV.log 2>&1; echo "pytest($TV) rc=$?"; tail -3 $O/pytest_$TV.log
cp /tmp/lib_keep.so enerf_amd/libenerf_hip.so
bash tools/ab_variants.sh "$@" 2>&1 | tee $O/ab_variants.txt
