#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace, every kernel alone: --single-stream) of one workload under enerf_options_t variants
# and library variants.   usage: gpu_r06_kstats.sh TAG "grep-pattern" WORKLOAD "opt1 opt2 ..." [lib variants ...]
#   an option spec is bench.py's --options string ("-" = defaults), e.g. "- conv3d_small_variant:1"
export TMPDIR=/tmp
TAG=$1; PAT=$2; WL=$3; OPTS=$4; shift 4; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cp $R/enerf_amd/libenerf_hip.so /tmp/lib_keep.so
cd /tmp
for v in base "$@"; do
  [ "$v" != base ] && cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  for o in $OPTS; do
    oo=""; [ "$o" != "-" ] && oo="--options $o"
    n=$(echo "${v}_$o" | tr ':,' '__')
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pkv_$n -o p -- python $R/bench.py --workload $WL --steps 40 --warmup 5 --no-cpu-baseline --no-stages --no-live-pmc --no-secondary --no-sync-per-frame --single-stream $oo > $O/prof_$n.log 2>&1
    f=$(find /tmp/pkv_$n -name "*kernel_stats.csv" | head -1); cp "$f" $O/stats_$n.csv
    echo "== lib $v options $o:"; grep -E "$PAT" $O/stats_$n.csv | python -c "
import sys,csv
tot=0
for r in csv.reader(sys.stdin):
    print('   ', r[0][:78], 'n', r[1], 'avg us', round(float(r[3])/1e3,2), 'min', round(float(r[5])/1e3,2)); tot+=float(r[2])/45/1e3
print('    sum per frame us', round(tot,1))"
  done
  cp /tmp/lib_keep.so $R/enerf_amd/libenerf_hip.so
done
