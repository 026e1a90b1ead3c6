#!/bin/bash
# Same-box A/B of library variants built by tools/build_variant.py:  bash tools/ab_variants.sh [-w WORKLOAD] base v1 v2 ...
# Each variant is benched (sequential frames, no per-frame sync) and its per-kernel stats collected once.
export TMPDIR=/tmp
WL=dtu; if [ "$1" = "-w" ]; then WL=$2; shift 2; fi
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_var; mkdir -p $O
cp $R/enerf_amd/libenerf_hip.so /tmp/lib_orig.so
FLAGS="--workload $WL --no-cpu-baseline --no-stages --no-sync-per-frame"
for rep in 1 2; do
  for v in "$@"; do
    cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
    (cd $R && timeout 300 python bench.py --steps 100 --warmup 10 $FLAGS > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err)
    echo "$v #$rep: $(python -c "import json; d=json.loads(open('$O/bench_${v}_$rep.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4))" 2>&1)"
  done
done
cd /tmp
for v in "$@"; do
  cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $R/bench.py --steps 20 --warmup 5 $FLAGS > $O/prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1); cp "$f" $O/stats_$v.csv
  echo "== $v:"; grep -E "render_rays|smooth0|feature_volume" $O/stats_$v.csv | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   ', r[0][:60], 'avg us', round(float(r[3])/1e3,1))"
done
cp /tmp/lib_orig.so $R/enerf_amd/libenerf_hip.so
