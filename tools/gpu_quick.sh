#!/bin/bash
# One short gpurun call: GPU parity tests, the default dtu bench line, and per-kernel durations of the dtu frame with every
# kernel alone (--single-stream) and in the default frame.  usage: bash tools/gpu_quick.sh TAG [pytest-args]
export TMPDIR=/tmp
TAG=${1:-r03_q}; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q "$@" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_dtu.json 2> $O/bench_dtu.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_dtu.json").read().strip().splitlines()[-1])
print("value", round(d["value"],1), "ms", round(d["ms_per_step"],4), "lat", d.get("latency_ms"), "seq", d.get("sequential_fps"))
print("stages", d.get("stages_ms")); print("stages_default", d.get("stages_ms_default"))
print("roofline", {k: d["roofline"][k] for k in ("frac","avg_launch_ms","pmc_stale")} if "roofline" in d else None)
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline_config1",{}).get("value"), d.get("parity_vs_oracle"))
PY
cd /tmp
for mode in single default; do
  X=""; [ $mode = single ] && X="--single-stream"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$mode -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stages --no-sync-per-frame $X > $O/stats_$mode.log 2>&1
  cp $(find /tmp/ps_$mode -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$mode.csv
done
python - <<PY
import csv
for mode in ("single",):
    rows=list(csv.DictReader(open("$O/kernel_stats_%s.csv"%mode)))
    tot=0
    for r in rows:
        if "enerf" in r["Name"]:
            n=r["Name"].replace("void enerf::","").replace("enerf::","").split("(")[0]
            per=float(r["TotalDurationNs"])/20/1e3
            if per>1: print(f"{n:48s} calls/frame {int(r['Calls'])/20:4.1f}  us/frame {per:7.1f}  avg {float(r['AverageNs'])/1e3:7.1f}")
            tot+=per
    print("sum us/frame", round(tot,1))
PY
