#!/bin/bash
# A/B of the persistent producer/consumer conv3d kernel against the default paths (one box, one call).
export TMPDIR=/tmp
O=gpurun_out/ab_ws; mkdir -p $O
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stages"
ENERF_CONV_WS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "goldens or full" > $O/pytest_ws.log 2>&1; echo "pytest ws rc=$?" 
for cfg in "base" "ws2:ENERF_CONV_WS=1 ENERF_CONV_WS_BD=2" "ws4:ENERF_CONV_WS=1 ENERF_CONV_WS_BD=4" "ws2b512:ENERF_CONV_WS=1 ENERF_CONV_WS_BD=2 ENERF_CONV_WS_BLOCKS=512" "base2"; do
  name=${cfg%%:*}; envs=${cfg#*:}; [ "$name" = "$cfg" ] && envs=""
  env $envs timeout 300 $B > $O/bench_$name.json 2> $O/bench_$name.err
  echo "$name: $(python -c "import json,sys; d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>&1)"
done
for cfg in "base" "ws2:ENERF_CONV_WS=1 ENERF_CONV_WS_BD=2" "ws4:ENERF_CONV_WS=1 ENERF_CONV_WS_BD=4"; do
  name=${cfg%%:*}; envs=${cfg#*:}; [ "$name" = "$cfg" ] && envs=""
  (cd /tmp && env $envs timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-stages > /dev/null 2>&1)
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/stats_$name.csv 2>/dev/null
  echo "== $name"; grep -i "conv3d" $O/stats_$name.csv | cut -d, -f1-5 | head -20
done
