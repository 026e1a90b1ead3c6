#!/bin/bash
# round-4 A/B: k_mlp_bwd with the backward weight images in LDS (default) vs global memory (round 3), one block per CU, and the
# timing ablation without the per-layer saves.  Per variant: ms per training step and the stand-alone launch time of
# k_mlp_bwd<3,3> (bench.py's roofline object) + the per-kernel table of the step.
# variants are built beforehand: python tools/build_variant.py NAME --file mlp_train.hip -D...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_r04_mlpb; mkdir -p $O
cp $R/enerf_amd/libenerf_hip.so /tmp/lib_keep3.so
cd $R
for v in "$@"; do
  cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  timeout 600 python bench.py --train --no-perceptual --steps 10 --warmup 2 --no-cpu-baseline > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$v.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print("$v ms_per_step", round(d["ms_per_step"],3), "k_mlp_bwd<3,3> ms", r.get("avg_launch_ms"), "graph", d.get("config",{}).get("graph"), "loss", d.get("final_loss"))
except Exception as e:
    print("$v failed", e); print(open("$O/$v.err").read()[-600:])
PY
done
cp /tmp/lib_keep3.so $R/enerf_amd/libenerf_hip.so
cd /tmp
for v in "$@"; do
  cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  rm -rf /tmp/pt_$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$v -o p -- python $R/bench.py --train --no-perceptual --steps 8 --warmup 2 --no-stages --no-cpu-baseline > $O/prof_$v.log 2>&1
  python - <<PY
import csv, glob
f=glob.glob("/tmp/pt_$v/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(k in r["Name"] for k in ("mlp_bwd", "gather", "feature_volume_bwd", "channel_sums", "channel_affine", "k_zero", "conv_wgrad", "wgrad_reduce")):
        print("$v", r["Name"][:60], "calls", r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1), "min", round(float(r["MinNs"])/1e3,1), "max", round(float(r["MaxNs"])/1e3,1))
PY
done
cp /tmp/lib_keep3.so $R/enerf_amd/libenerf_hip.so
