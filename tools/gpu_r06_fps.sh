#!/bin/bash
# frames/s of library variants under the reference's per-frame-sync protocol, interleaved repeats.  usage: gpu_r06_fps.sh TAG "wl1 wl2" REPS v1 v2 ... ("base" = in-tree)
export TMPDIR=/tmp
TAG=$1; WLS=$2; REPS=$3; shift 3; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R; cp enerf_amd/libenerf_hip.so /tmp/lib_keep.so
for rep in $(seq 1 $REPS); do
  for wl in $WLS; do
    for v in "$@"; do
      [ "$v" != base ] && cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
      timeout 300 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-stages --no-live-pmc --no-secondary > $O/b_${wl}_${v}_$rep.json 2>/dev/null
      echo "$wl $v #$rep: $(python -c "import json; d=json.loads(open('$O/b_${wl}_${v}_$rep.json').read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', round(d['ms_per_step'],4), 'ms')" 2>&1)"
      cp /tmp/lib_keep.so $R/enerf_amd/libenerf_hip.so
    done
  done
done
