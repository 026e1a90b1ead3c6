#!/bin/bash
# per-kernel time of the config-5 training step.  usage: gpu_train_stats.sh TAG [extra bench flags]
# whole-process statistics (train_kernel_stats.csv) AND the per-step table of the last graph replays (train_step_kernels.csv,
# tools/train_step_profile.py: warm-up / capture / verification steps excluded)
export TMPDIR=/tmp
TAG=${1:-r04_train_ks}; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o p -- python $R/bench.py --train --steps 12 --warmup 2 --no-stages --no-cpu-baseline "$@" > $O/bench.log 2>&1
cp $(find /tmp/pt -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
tail -1 $O/bench.log | cut -c1-300
python $R/tools/train_step_profile.py $(find /tmp/pt -name "*kernel_trace.csv" | head -1) 8 $O/train_step_kernels.csv
