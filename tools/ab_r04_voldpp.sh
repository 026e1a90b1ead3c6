#!/bin/bash
# round-4 A/B: the warp kernel's group broadcasts as DPP register permutes (default) vs ds_bpermute (rounds 1-3): per-kernel time
# (kernels alone), the per-frame-sync frame rate of all three workloads, and the parity tests that read the cost volume.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_r04_voldpp; mkdir -p $O
cd $R; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "goldens or full_size_dtu or stage" 2>&1 | tail -2
bash $R/tools/gpu_kvariants.sh ab_r04_voldpp "feature_volume" volshfl cur
cp $R/enerf_amd/libenerf_hip.so /tmp/lib_keep4.so
for rep in 1 2; do for v in volshfl cur; do
  cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  for wl in dtu zju; do
  python bench.py --workload $wl --no-cpu-baseline --no-stages --no-live-pmc --steps 400 --warmup 50 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep$rep $wl', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done; done
cp /tmp/lib_keep4.so $R/enerf_amd/libenerf_hip.so
