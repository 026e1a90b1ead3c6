#!/bin/bash
# round-5 A/B of FeatureNet kernel variants (library variants built by tools/build_variant.py into enerf_amd/_ab/):
#   micro (A-operand broadcast semantics + rate), parity of the default library on the FeatureNet tests, then per variant and
#   workload: frames/s under the reference's per-frame-sync protocol + per-kernel durations with every kernel alone (--single-stream).
# usage: ab_r05_featnet.sh TAG "kernel-grep-pattern" "wl1 wl2" v1 v2 ...
export TMPDIR=/tmp
TAG=$1; PAT=$2; WLS=$3; shift 3; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
[ -x tools/micro/mfma_cbsz.bin ] && timeout 120 tools/micro/mfma_cbsz.bin 2>&1 | tee $O/mfma_cbsz.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "feature_net or goldens or full_size_zju or full_size_dtu_eval" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cp enerf_amd/libenerf_hip.so /tmp/lib_keep.so
for wl in $WLS; do
  for v in "$@"; do
    cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
    for rep in 1 2; do
      (cd $R && timeout 300 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-stages --no-live-pmc > $O/bench_${wl}_${v}_$rep.json 2> $O/bench_${wl}_${v}_$rep.err)
      echo "$wl $v #$rep: $(python -c "import json; d=json.loads(open('$O/bench_${wl}_${v}_$rep.json').read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],4), 'ms')" 2>&1)"
    done
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_${wl}_$v -o p -- python $R/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-stages --no-live-pmc --no-sync-per-frame --single-stream > $O/prof_${wl}_$v.log 2>&1)
    f=$(find /tmp/pk_${wl}_$v -name "*kernel_stats.csv" | head -1); cp "$f" $O/stats_${wl}_$v.csv
    echo "== $wl $v (kernels alone):"; grep -E "$PAT" $O/stats_${wl}_$v.csv | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   ', r[0][:72], 'avg us', round(float(r[3])/1e3,1), 'min', round(float(r[5])/1e3,1))"
  done
done
cp /tmp/lib_keep.so $R/enerf_amd/libenerf_hip.so
