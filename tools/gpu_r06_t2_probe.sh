cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for o in "" "conv3d_t2_variant:1"; do
  for w in zju lego; do
    python bench.py --workload $w --no-secondary --steps 100 --warmup 10 --no-stages --no-cpu-baseline --no-live-pmc ${o:+--options $o} 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '[$o]', round(d['value'],1), 'fps', round(d['ms_per_step'],4), 'ms')"
  done
done
done
