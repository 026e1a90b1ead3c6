#!/bin/bash
# per-kernel durations (single stream: kernels alone) for library variants in enerf_amd/_ab/.  usage: gpu_kvariants.sh TAG "grep-pattern" v1 v2 ...
export TMPDIR=/tmp
TAG=$1; PAT=$2; shift 2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cp $R/enerf_amd/libenerf_hip.so /tmp/lib_keep.so
cd /tmp
for v in "$@"; do
  cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pkv_$v -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stages --no-sync-per-frame --single-stream > $O/prof_$v.log 2>&1
  f=$(find /tmp/pkv_$v -name "*kernel_stats.csv" | head -1); cp "$f" $O/stats_$v.csv
  echo "== $v:"; grep -E "$PAT" $O/stats_$v.csv | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   ', r[0][:70], 'avg us', round(float(r[3])/1e3,1), 'min', round(float(r[5])/1e3,1))"
done
cp /tmp/lib_keep.so $R/enerf_amd/libenerf_hip.so
