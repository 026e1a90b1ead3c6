#!/bin/bash
# round-4 A/B: the render kernel with the MLP's dense layers on the bf16 matrix cores (enerf_options_t.render_precision = 2: bf16x3, 3: bf16x6; default = exact fp32)
# against the exact fp32 MFMA kernel: per-kernel duration (kernels alone), per-frame-sync fps, parity at the full DTU size.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_r04_bx3; mkdir -p $O
cd $R
timeout 600 python -m pytest "tests/test_gpu_parity.py::test_full_size_dtu_eval_vs_oracle_and_properties" -q -x > $O/pytest.log 2>&1; echo "parity rc=$?"; tail -2 $O/pytest.log
python - <<'PY'
import torch, numpy as np, sys
sys.path.insert(0, "tests")
from enerf_amd.config import EnerfConfig
from enerf_amd.synth import make_batch, make_zju_batch
from enerf_amd.lib import Options
from oracle import enerf_oracle as O
from golden_cases import load_weights
from test_gpu_parity import _net, _to, _rel
for wl in ("dtu", "zju"):
    if wl == "dtu":
        cfg = EnerfConfig.dtu_eval(); b = make_batch(512, 640, 3, cfg, seed=0, textured=True); human=False
    else:
        cfg = EnerfConfig().with_cas(volume_planes=(32, 8), render_if=(False, True)); b = make_zju_batch(1024, 1024, 4, cfg, seed=6); human=True
    batch = {k: torch.from_numpy(v) for k, v in b.items()}
    net = _net(cfg, human)
    with torch.no_grad(): ref = O.forward(cfg, load_weights(), batch)
    for name, opt in (("fp32", None), ("bx3", Options(render_precision=2)), ("bx6", Options(render_precision=3))):
        o = net._forward(_to(batch), opt) if opt is not None else net(_to(batch))
        print(wl, name, {k: f"{_rel(o[k].cpu(), ref[k]):.1e}" for k in ref}, "psnr", round(O.psnr(o["rgb_level1"].cpu(), ref["rgb_level1"]), 1))
PY
for wl in dtu zju lego; do for cfg in "" "--options render_precision:2" "--options render_precision:3"; do for rep in 1 2; do
  python bench.py --workload $wl --no-cpu-baseline --no-stages --steps 400 --warmup 50 $cfg 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl [$cfg] rep$rep', round(d['value'],1), round(d['ms_per_step'],4))"
done; done; done
cd /tmp
for tag in fp32 bx3 bx6; do
  X=""; [ $tag = bx3 ] && X="--options render_precision:2"; [ $tag = bx6 ] && X="--options render_precision:3"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbx_$tag -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stages --no-sync-per-frame --single-stream $X > $O/prof_$tag.log 2>&1
  f=$(find /tmp/pbx_$tag -name "*kernel_stats.csv" | head -1); cp "$f" $O/stats_$tag.csv
  echo "== $tag:"; grep -E "render_rays" $O/stats_$tag.csv | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   ', r[0][:80], 'avg us', round(float(r[3])/1e3,1), 'min', round(float(r[5])/1e3,1))"
done
