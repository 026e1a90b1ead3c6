export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for wl in dtu zju lego; do for g in 0 2 3 4; do
  python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-stages --no-live-pmc --options side_gate:$g 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl side_gate=$g', round(d['value'],1), round(d['ms_per_step'],4))"
done; done
