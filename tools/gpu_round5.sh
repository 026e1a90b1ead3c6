#!/bin/bash
# round-5 evidence in one call: the driver's bench command (timed), pytest -m gpu, smoke().   usage: gpu_round5.sh TAG [pytest-k]
export TMPDIR=/tmp
TAG=$1; K=$2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
T0=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
echo "bench wall $(python -c "import time; print(round(time.time() - $T0, 1))") s"; tail -3 $O/bench_default.err
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "roofline.frac", d["roofline"]["frac"], "stages", d.get("stages_ms"))
print("stage_roofline", {k: v.get("frac") for k, v in d.get("stage_roofline", {}).items()})
print(json.dumps(d.get("workloads"), indent=1)[:6000])
PY
[ "$K" = "nobench-only" ] && exit 0
if [ -n "$K" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" > $O/pytest_gpu.log 2>&1; else timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; fi
echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
