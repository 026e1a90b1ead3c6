export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_abl; mkdir -p $O; cd $R
cp enerf_amd/libenerf_hip.so /tmp/lib_keep.so
for wl in ${WLS:-dtu zju}; do for v in ${VARIANTS:-r5def r5abl1 r5abl2 r5abl3}; do
  cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_${wl}_$v -o p -- python $R/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-stages --no-live-pmc --no-sync-per-frame --single-stream > $O/prof_${wl}_$v.log 2>&1)
  f=$(find /tmp/pk_${wl}_$v -name "*kernel_stats.csv" | head -1); cp "$f" $O/stats_${wl}_$v.csv 2>/dev/null
  echo "== $wl $v:"; grep -E "${PAT:-k_conv2d<}" $O/stats_${wl}_$v.csv | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   ', r[0][:60], 'avg us', round(float(r[3])/1e3,1))"
done; done
cp /tmp/lib_keep.so $R/enerf_amd/libenerf_hip.so
