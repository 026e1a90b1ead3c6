# usage: gpu_micro_wgrad.sh TAG     kernel stats + two PMC passes of tools/micro_wgrad_tiled.py
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp
CMD="python $R/tools/micro_wgrad_tiled.py 5"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mw_stats -o p -- $CMD > $O/stats.log 2>&1
cp $(find /tmp/mw_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find /tmp/mw_stats -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv
i=0
for ctr in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/mw_pmc$i -o p -- $CMD > $O/pmc$i.log 2>&1
  cp $(find /tmp/mw_pmc$i -name "*counter_collection.csv" | head -1) $O/pmc$i.csv
done
cd $R && python tools/pmc_aggregate.py $O micro_wgrad > /dev/null 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/micro_wgrad_pmc_per_kernel.csv")))
for r in rows:
    if "wgrad" in r["kernel"] or "colsum" in r["kernel"]:
        print(r["kernel"][:44].ljust(44), "mfma_busy", r["mfma_busy_frac"], "wait_inst", r["SQ_WAIT_INST_ANY_frac"], "wait_any", r["SQ_WAIT_ANY_frac"],
              "valu", r["SQ_ACTIVE_INST_VALU_frac"], "lds_conf/idx", r["SQ_LDS_BANK_CONFLICT"], r["SQ_LDS_IDX_ACTIVE"], "gui", r["GRBM_GUI_ACTIVE"], "hbm MB", round(float(r["hbm_bytes_corrected"]) / 1e6, 1))
st = list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in st:
    if "wgrad" in r["Name"] or "colsum" in r["Name"]:
        print(r["Name"][:60].ljust(60), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), round(float(r["MinNs"]) / 1e3, 1), round(float(r["MaxNs"]) / 1e3, 1))
PY
