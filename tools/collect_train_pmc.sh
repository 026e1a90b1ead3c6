#!/bin/bash
# PMC counters of the TRAINING step's kernels (config 5, eager steps: the kernels of the captured step, one at a time):
# the four separate --pmc passes of tools/collect_profiles.sh over bench.py --train --train-eager, aggregated per kernel by
# tools/pmc_aggregate.py into gpurun_out/prof_<TAG>_train/<TAG>_train_pmc_per_kernel.csv.   usage: collect_train_pmc.sh TAG
export TMPDIR=/tmp
TAG=${1:-rXX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${TAG}_train; mkdir -p $O
cd /tmp
PB="python $R/bench.py --train --train-eager --no-perceptual --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc"
i=0
for ctr in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/p_tpmc${i}_$TAG -o p -- $PB > $O/pmc$i.log 2>&1
  cp $(find /tmp/p_tpmc${i}_$TAG -name "*counter_collection.csv" | head -1) $O/pmc$i.csv
done
cd $R && python tools/pmc_aggregate.py $O ${TAG}_train | tail -2
