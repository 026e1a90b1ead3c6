export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --workload dtu --steps 20 --warmup 5 --no-cpu-baseline --no-stages --no-sync-per-frame --single-stream"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o p -- $B > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/ps/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]: print(f"{r['Name'][:70]:70s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.1f}")
PY
PB="python $R/bench.py --workload dtu --steps 3 --warmup 2 --no-cpu-baseline --no-stages --no-sync-per-frame --single-stream"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pp1 -o p -- $PB > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d /tmp/pp2 -o p -- $PB > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
for d in ('/tmp/pp1','/tmp/pp2'):
    f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'b4' not in k and 'pk8' not in k and 's1_lds' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items(): print(k[:60], {a:round(b/5) for a,b in v.items()})
PY
