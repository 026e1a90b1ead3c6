#!/bin/bash
# Same-box A/B of library variants built by tools/build_variant.py:  bash tools/ab_variants.sh base v1 v2 ...
# Each variant is benched twice (interleaved) and its per-kernel stats collected once.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_var; mkdir -p $O
cp $R/enerf_amd/libenerf_hip.so /tmp/lib_orig.so
for rep in 1 2; do
  for v in "$@"; do
    cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
    (cd $R && timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stages --in-flight 1 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err)
    echo "$v #$rep: $(python -c "import json; d=json.loads(open('$O/bench_${v}_$rep.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4))" 2>&1)"
  done
done
cd /tmp
for v in "$@"; do
  cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stages --in-flight 1 > $O/prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1); cp "$f" $O/stats_$v.csv
  echo "== $v: $(grep render_rays $O/stats_$v.csv | python -c "import sys,csv; r=next(csv.reader(sys.stdin)); print('render avg us', float(r[3])/1e3)")"
done
cp /tmp/lib_orig.so $R/enerf_amd/libenerf_hip.so
