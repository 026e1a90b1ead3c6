#!/bin/bash
# A/B of enerf_channel_sums' block cap (library variants built with -DENERF_CS_CAP=...): tools/bench_channel_sums.py per variant
R=$GRAFT_REPO_ROOT; cp $R/enerf_amd/libenerf_hip.so /tmp/lib_keep_cs.so
for v in "$@"; do cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so; echo "== $v"; timeout 200 python $R/tools/bench_channel_sums.py 2>&1 | tail -6; done
cp /tmp/lib_keep_cs.so $R/enerf_amd/libenerf_hip.so
