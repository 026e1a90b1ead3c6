export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pz -o p -- python $R/bench.py --workload zju --steps 10 --warmup 3 --no-cpu-baseline --no-stages --no-sync-per-frame --single-stream > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/pz/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:26]: print(f"{r['Name'][:74]:74s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.1f}")
PY
