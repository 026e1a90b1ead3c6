#!/bin/bash
# Collect the evidence DESIGN.md §7 / bench.py cite, on the GPU box, into gpurun_out/prof_<tag>/ :
#   bench line (default flags), rocprofv3 kernel stats of the same workload, and four separate --pmc passes
#   (SQ issue/busy, SQ wait/LDS, FETCH_SIZE, WRITE_SIZE — MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and
#   WRITE_SIZE do not fit one pass; --pmc is never combined with sys/hip/hsa traces).
#   usage: bash tools/collect_profiles.sh TAG [WORKLOAD=dtu] [pmc=1|0]
export TMPDIR=/tmp
TAG=${1:-rXX}; WL=${2:-dtu}; PMC=${3:-1}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd $R && python bench.py --workload $WL --no-secondary > $O/${TAG}_bench.json 2> $O/bench.err; tail -c 600 $O/${TAG}_bench.json; echo
cd /tmp
# frames enqueued back to back on one stream, nothing but the timed frames: clean per-kernel durations
BENCH="python $R/bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-stages --no-sync-per-frame"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats_$TAG -o p -- $BENCH > $O/stats.log 2>&1
cp $(find /tmp/p_stats_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
[ "$PMC" = "1" ] || exit 0
PB="python $R/bench.py --workload $WL --steps 3 --warmup 2 --no-cpu-baseline --no-stages --no-sync-per-frame --single-stream"   # kernels alone: no side lane
i=0
for ctr in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/p_pmc${i}_$TAG -o p -- $PB > $O/pmc$i.log 2>&1
  cp $(find /tmp/p_pmc${i}_$TAG -name "*counter_collection.csv" | head -1) $O/pmc$i.csv
done
cd $R && python tools/pmc_aggregate.py $O $TAG
