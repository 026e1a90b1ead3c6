#!/bin/bash
# run one pytest selection against several library variants (same box)
R=$GRAFT_REPO_ROOT; SEL="$1"; shift
cp $R/enerf_amd/libenerf_hip.so /tmp/lib_orig.so
for v in "$@"; do
  cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  echo "== $v: $(cd $R && python -m pytest tests/test_gpu_parity.py -m gpu -q -k "$SEL" 2>&1 | grep -E "passed|failed" | tail -1)"
done
cp /tmp/lib_orig.so $R/enerf_amd/libenerf_hip.so
