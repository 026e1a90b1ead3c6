#!/bin/bash
# round-4 A/B 1: warp kernel variants (byte offsets + packed fp32, fast reciprocals) and XCD-banded render tiles:
# per-kernel durations (kernels alone), the per-frame-sync latency, and FETCH_SIZE of the render kernel (base vs banded).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_r04_1; mkdir -p $O
bash $R/tools/gpu_kvariants.sh ab_r04_1 "feature_volume|render_rays" base volpk volfast rxcd
cp $R/enerf_amd/libenerf_hip.so /tmp/lib_keep2.so
cd $R
for rep in 1 2; do for v in base volpk volfast rxcd; do
  cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  python bench.py --no-cpu-baseline --no-stages --steps 400 --warmup 50 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep$rep', round(d['value'],1), round(d['ms_per_step'],4))"
done; done
cd /tmp
for v in base rxcd; do
  cp $R/enerf_amd/_ab/lib_$v.so $R/enerf_amd/libenerf_hip.so
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcf_$v -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-stages --no-sync-per-frame --single-stream > $O/pmc_$v.log 2>&1
  f=$(find /tmp/pmcf_$v -name "*counter_collection.csv" | head -1)
  python - <<PY
import csv, collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open("$f")):
    if r.get("Counter_Name")=="FETCH_SIZE" and ("render_rays" in r["Kernel_Name"] or "feature_volume" in r["Kernel_Name"]):
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print("$v FETCH_SIZE(raw units, x64B x2 on gfx950)", k, "mean", sum(v)/len(v), "n", len(v), "-> MB fetched", 2*64*sum(v)/len(v)/1e6*0+sum(v)/len(v))
PY
done
cp /tmp/lib_keep2.so $R/enerf_amd/libenerf_hip.so
