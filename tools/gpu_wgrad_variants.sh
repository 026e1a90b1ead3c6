#!/bin/bash
# tools/bench_wgrad_layers.py under library variants in enerf_amd/_ab/.  usage: gpu_wgrad_variants.sh v1 v2 ...
R=$GRAFT_REPO_ROOT; cp $R/enerf_amd/libenerf_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  echo "== $v"; python $R/tools/bench_wgrad_layers.py 2>/dev/null | grep -E "L1|total" | cut -c1-62
done
cp /tmp/lib_keep.so $R/enerf_amd/libenerf_hip.so
