"""bench.py — rendered frames/sec of the MI355X-native ENeRF path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20 [--workload dtu|lego|zju]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one ``Network.forward(batch)`` = one rendered frame: the HIP FeatureNet + the whole cascade through ONE
``enerf_forward`` C call.  Default workload = BASELINE config 2, the configuration the reference quotes its 21.78 FPS on
(README.md:113: DTU 512x640, 3 source views, ``render_if False,True``, ``volume_planes 48,8``); ``--workload lego`` /
``zju`` are BASELINE configs 3 / 4 at their real shapes.  Inputs are synthetic, resident in HBM before
the timed region; weights are random-init with randomised BN statistics.

Protocol of ``value`` (SURVEY.md §8d = the reference's run.py:62-76): for each of the K timed frames
``synchronize(); network(batch); synchronize()`` on one stream with the default kernel variants — so
``value == 1 / latency_ms.mean`` — bracketed by a barrier + synchronize, MAX over ranks.  Rendering is frame-parallel
(frame f -> rank f mod G, no data-path collective: scaling = weak; enerf_amd/frame_parallel.py).  Extra keys, never
``value``: ``sequential_fps`` (same frames without the per-frame sync), ``pipelined_fps`` (frames in flight on several
HIP streams with the throughput kernel options), ``sustained`` (>= 2000 frames).  Rank 0 prints ONE JSON line.

``--gpus N`` with N > 1 and no ``WORLD_SIZE`` in the environment (a plain ``python bench.py --gpus 8``) re-executes itself
under ``torch.distributed.run`` with N ranks on 127.0.0.1; a line whose ``n_gpus`` differs from ``--gpus`` is never printed.
The timed frames rotate over ``--batches`` (default 4) distinct seeded batches, all resident in HBM before the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BASELINE_FPS_RTX3090 = 21.778975517304048      # BASELINE.md §1 / README.md:121 (DTU eval config only)
PEAK_L1_GATHER_TBS = 24.1          # measured: tools/micro/gather_rate.hip (L1-resident float4 gathers, requested bytes)
PEAK_F32_MFMA_TFLOPS = 157.3                   # MI355X_MICROARCH.md (v_mfma_f32_16x16x4_f32, 2.4 GHz)
PEAK_HBM_GBS = 8000.0


class StageTimer:
    """Per-stage HIP events: enerf_forward records the events this object hands out (same stream as the kernels)."""

    def __init__(self):
        self.frames = []

    def new_events(self, n):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        for e in evs:
            e.record()                                 # creates the underlying hipEvent_t; the C call re-records it
        return evs

    def frame(self, named_events):
        self.frames.append(named_events)

    def summary(self):
        acc = {}
        for fr in self.frames:
            for (_, e0), (n1, e1) in zip(fr[:-1], fr[1:]):
                acc.setdefault(n1, []).append(e0.elapsed_time(e1))
        return {k: sum(v) / len(v) for k, v in acc.items()}


# ---- algorithmic work --------------------------------------------------------------------------------
# reference-over-port time ratios of the CPU baseline (`kind: "port"`): the unmodified reference modules against oracle/enerf_oracle.py
# on the same inputs and threads in the build container (profiles/r03_cpu_ref_vs_port.txt, r05_cpu_ref_vs_port_train.txt; outputs /
# gradients bit-identical).  reference seconds = port seconds x this; the reference cannot travel to the GPU box.
REF_OVER_PORT = {"render_ft": round(1 / 0.820, 3), "render_tt": round(1 / 1.007, 3), "train": round(1 / 0.935, 3)}


def render_mfma_tiles_per_16(S, R):
    """16x16x4 fp32 MFMA tiles k_render_rays issues per 16 samples (render.hip): view_fc, global_fc (shared + per view),
    fc, lr0, color.0 (shared + per view)."""
    TR = (R + 3) // 4
    return S * TR + 4 * R + 2 * S * R + 8 + 24 + 88 + 4 * S * (R + 1)


def mlp_bwd_mfma_tiles_per_16(S, R):
    """16x16x4 fp32 MFMA tiles k_mlp_bwd issues per 16 points (mlp_train.hip): the forward recompute (= the render kernel's
    MLP), the colour layer's per-view half a second time, and the transposed-weight products b1..b7.  Equals the static
    v_mfma count of the kernel's ISA (tools/isa_count.py: 566 at S=3,R=3 with the in-kernel weight gradients, 482 without; 940 at R=9)."""
    TR, TX = (R + 3) // 4, (R + 4) // 4
    fwd = render_mfma_tiles_per_16(S, R)
    bwd = 4 * S * (R + 1) + 16 * S * TX + 96 + 32 + 8 + 16 * TR + 8 * S * TR + S * TX * R
    # round 6, R = 3 (F = 11): the per-view layers' weight gradients inside the kernel (k = point products): color.0's per-view columns
    # 16 per view; with S <= 3 also global_fc's `a` columns 8 and view_fc 4 per view (autograd.NerfMlpFn: level 2, else 1)
    wg = (28 * S if S <= 3 else 16 * S) if R == 3 else 0
    return fwd + bwd + wg


def render_dense_flop_per_sample(S, F):
    """The reference's dense count of the same maths (SURVEY.md §8a derivation, nerf.py:29-89)."""
    return 2 * 4 * F * S + 2 * 3 * F * 32 * S + 2 * 32 * S + 2 * 32 * 16 + 2 * 24 * 64 + 2 * 64 + (2 * (88 + F + 4) * 64 + 2 * 64) * S


def cost_reg_gflop(C, full, D, h, w):
    """Dense conv FLOPs of MinCostRegNet / CostRegNet (cost_reg_net.py) on a (C,D,h,w) volume."""
    n0 = D * h * w
    n1, n2, n3 = n0 // 8, n0 // 64, n0 // 512
    f = 27 * 2 * (C * 8 * n0 + 8 * 16 * n1 + 16 * 16 * n1 + 16 * 32 * n2 + 32 * 32 * n2 + 32 * 16 * n2 + 16 * 8 * n1 + 8 * 9 * n0)
    if full:
        f += 27 * 2 * (32 * 64 * n3 + 64 * 64 * n3 + 64 * 32 * n3)
    return f / 1e9


PMC_SENTINEL = "k_pack_rgb8"          # a library kernel no frame launches: brackets the measured frames of a workload in the PMC child


def pmc_sequence_child(names, dev):
    """`bench.py --pmc-sequence dtu,lego,zju` (the child of live_pmc): per workload, build the network, run two untimed frames,
    then a sentinel launch, three frames with every kernel alone on one stream, and a sentinel again.  The parent cuts the
    dispatch-ordered counter rows at the sentinels, so ONE child per counter pass covers every workload of the line."""
    from __graft_entry__ import _seeded_network
    from enerf_amd.lib import Options, get_lib
    lib = get_lib()
    dummy = torch.zeros((64, 3), device=dev)
    for name in names:
        cfg, batch_np, human, _ = make_workload(name, 0)
        net = _seeded_network(cfg, dev, human=human)
        net.options = Options(single_stream=1)
        batch = {k: torch.from_numpy(v).to(dev) for k, v in batch_np.items()}
        with torch.no_grad():
            for _ in range(2):
                net(batch)
            torch.cuda.synchronize()
            lib.pack_rgb8(dummy, 8, 8)
            for _ in range(3):
                net(batch)
            lib.pack_rgb8(dummy, 8, 8)
            torch.cuda.synchronize()
        del net, batch
        torch.cuda.empty_cache()


def live_pmc(workloads, child_flags=None, kernel_prefix=None):
    """HBM traffic + matrix-pipe busy fraction per kernel, measured NOW: three short rocprofv3 --pmc passes (FETCH_SIZE /
    WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; separate passes, --kernel-trace only — the MI355X_MICROARCH.md
    recipe, as tools/collect_profiles.sh) over a child run of this script (`--pmc-sequence`: three frames per workload, kernels
    alone on one stream).  Returns {workload: {kernel name: {hbm_bytes_per_launch, fetch_size_kb, write_size_kb, mfma_busy_frac}}}
    — with ``child_flags`` (the training line: eager steps of the same loss) {"": {...}} over the whole child — or None when
    rocprofv3 is missing or a pass fails: the caller then replays the committed figures."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    child = [sys.executable, os.path.abspath(__file__), "--pmc-sequence", ",".join(workloads)]
    if child_flags is not None:                           # (the training line: eager steps of the same loss, kernels alone)
        child = [sys.executable, os.path.abspath(__file__), *child_flags, "--no-cpu-baseline", "--no-live-pmc"]
    env = dict(os.environ, TMPDIR="/tmp")
    env = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    acc = {}                                              # (workload, kernel) -> counter -> [values]
    for counters in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"):
        d = tempfile.mkdtemp(prefix="enerf_pmc_", dir="/tmp")
        try:
            subprocess.run([rocprof, "--pmc", *counters.split(), "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", *child],
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            pmc_accumulate(csv.DictReader(open(files[0])), workloads if child_flags is None else None, kernel_prefix, acc)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for (w, k), c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        if not all(n in m for n in ("FETCH_SIZE", "WRITE_SIZE")):
            continue
        e = {"hbm_bytes_per_launch": (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0,      # gfx950: FETCH_SIZE doubled
             "fetch_size_kb": m["FETCH_SIZE"], "write_size_kb": m["WRITE_SIZE"], "launches_measured": len(c["FETCH_SIZE"])}
        if m.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_frac"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0), 4)
        out.setdefault(w, {})[k] = e
    return out or None


def pmc_accumulate(rows, workloads, kernel_prefix, acc):
    """One counter pass's rows (rocprofv3 counter_collection.csv: one row per dispatch and counter) -> acc[(workload, kernel)][counter]
    += [values].  ``workloads`` (the --pmc-sequence child): the dispatch-ordered rows are cut at the sentinel launches — the rows between
    the (2k+1)-th and (2k+2)-th sentinel dispatch belong to workloads[k]; everything else (network construction, warm-up frames) is
    dropped.  ``workloads is None`` (the training child): every kernel whose name starts with ``kernel_prefix``, under workload ""."""
    rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]))
    seg, inside, last_dispatch = -1, False, None
    for r in rows:
        k = r["Kernel_Name"].replace("void enerf::", "").replace("enerf::", "")
        if workloads is not None:
            if k.startswith(PMC_SENTINEL):
                if r["Dispatch_Id"] != last_dispatch:     # (one row per counter and dispatch)
                    inside = not inside
                    seg += 1 if inside else 0
                    last_dispatch = r["Dispatch_Id"]
                continue
            if not inside or seg >= len(workloads):
                continue
            key = (workloads[seg], k)
        else:
            if kernel_prefix is not None and not k.startswith(kernel_prefix):
                continue
            key = ("", k)
        acc.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return acc


def pmc_lookup(pmc, workload, kernel_prefix):
    """The entry of the first kernel of `workload` whose name starts with `kernel_prefix` (live_pmc's result), or None."""
    if not pmc or workload not in pmc:
        return None
    for k, e in pmc[workload].items():
        if k.startswith(kernel_prefix):
            return e
    return None


def kernel_time_table(run, frames=8):
    """Per-kernel device time of `frames` calls of run(), measured in-process (torch.profiler = roctracer: every HIP kernel of
    this process, the library's ctypes launches included).  Returns [(kernel name, launches, avg us, share of kernel time)],
    largest total first.  Callers run it with enerf_options_t.single_stream = 1 so that every kernel is alone on the device."""
    from torch.profiler import ProfilerActivity, profile
    run(2)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run(frames)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages():
        t = getattr(e, "device_time_total", None)
        if t is None:
            t = getattr(e, "cuda_time_total", 0.0)
        if t and e.count:
            rows.append((e.key.replace("void enerf::", "").replace("enerf::", ""), int(e.count), float(t)))
    total = sum(r[2] for r in rows) or 1.0
    rows.sort(key=lambda r: -r[2])
    return [(k, c // 1, t / c, t / total) for k, c, t in rows]


def kernel_work_model(name, cfg, S, H, W, n_rays_by_level):
    """Algorithmic work of one launch of a library kernel at this workload's shapes: (bound, work, unit) or None.
    FLOP counts are the dense counts of the layers a kernel evaluates (no halo recompute), the render kernel's the issued MFMA
    tiles (its view-independent halves evaluated once per point)."""
    import re
    cas = cfg.cas
    vox = [cas.volume_planes[i] * int(H * cas.volume_scale[i]) * int(W * cas.volume_scale[i]) for i in range(cas.num)]
    m = re.match(r"k_render_rays<(\d+), (\d+)", name)
    if m:
        R, Sv = int(m.group(1)), int(m.group(2))
        lvl = [i for i in range(cas.num) if cas.render_if[i] and (cas.nerf_model_feat_ch[i] + 3 + 3) // 4 == R]
        if not lvl:
            return None
        n_samples = n_rays_by_level[lvl[0]] * cas.num_samples[lvl[0]]
        return "mfma", render_mfma_tiles_per_16(Sv, R) * 2 * 16 * 16 * 4 / 16.0 * n_samples, "flop"
    px = S * H * W
    if name.startswith("k_smooth0"):
        return "mfma", px * (2 * 9 * 32 * 8 + 2 * 8 * 32), "flop"                  # smooth0 3x3 32->8 + lat0 1x1 8->32
    if name.startswith("k_conv0_fused"):
        return "mfma", px * (2 * 9 * 3 * 8 + 2 * 9 * 8 * 8), "flop"
    m = re.match(r"k_conv3d_s1_b4[gc]?<(\d+), \d+, (true|false)>", name)
    if m:
        cin, heads = int(m.group(1)), m.group(2) == "true"
        if heads:                                                                   # the fused heads of every level share a name
            return "mfma", sum(vox) / len(vox) * 2 * 27 * 8 * 9, "flop"
        lv = [i for i in range(cas.num) if (32 >> i) == cin]
        return ("mfma", vox[lv[0]] * 2 * 27 * cin * 8, "flop") if lv else None
    m = re.match(r"k_conv2d<(\d+), (\d+), (\d+), (\d+)", name)
    if m:                                                                           # FeatureNet layers (conv2d.hip launch table)
        cinp, rt, k, st = (int(m.group(i)) for i in range(1, 5))
        scale = {(8, 5, 2): 4, (16, 3, 1): 4, (16, 5, 2): 16, (32, 3, 1): 16 if rt == 2 else 4, (16, 1, 1): 4}.get((cinp, k, st))
        if scale is None:
            return None
        cout = 16 * rt if not (cinp == 32 and rt == 1) else 16
        fl = 2 * k * k * cinp * cout + (2 * 32 * 32 if (cinp == 32 and rt == 2) else 0)   # conv2.1 carries the toplayer
        return "mfma", px / scale * fl, "flop"
    return None


def secondary_workload(name, dev, frames=100, pmc=None):
    """BASELINE configs 3 / 4 inside the default run: the reference's per-frame-sync protocol on `frames` frames of
    the workload (4 distinct resident batches, >= 100 untimed frames first), the per-stage times, and the kernel that holds the
    largest share of the workload's kernel time (kernels alone: single_stream) with its roofline fraction."""
    from __graft_entry__ import _seeded_network
    from enerf_amd.lib import Options
    cfg, batch_np, human, workload = make_workload(name, 0)
    net = _seeded_network(cfg, dev, human=human)
    S, H, W = batch_np["src_inps"].shape[1], batch_np["src_inps"].shape[3], batch_np["src_inps"].shape[4]
    batches = [{k: torch.from_numpy(v).to(dev) for k, v in (batch_np if j == 0 else make_workload(name, 1000 * j)[1]).items()}
               for j in range(4)]
    no = [0]

    def step():
        no[0] += 1
        return net(batches[no[0] % 4])
    import gc
    # the default run has built and dropped two networks by now: collect that garbage BEFORE the timed frames and keep the survivors
    # out of later collections (a generation-2 pass over them in the middle of 100 frames cost up to 30 ms in one run: 545 -> 468
    # frames/s on zju); the distribution is reported so that such an outlier is visible.  The collection itself comes BEFORE the
    # warm-up: an idle gap in front of the timed frames takes the device out of its steady clocks (see main()).
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()
    try:                                                  # (an exception in the timed frames must not leave the collector frozen)
        t_w, n_w, recent = time.perf_counter(), 0, []
        while True:
            t_f = time.perf_counter()
            step()
            torch.cuda.synchronize()
            now = time.perf_counter()
            recent.append(now - t_f)
            n_w += 1
            if n_w < 150 or now - t_w < 0.5:
                continue
            a, b = sum(recent[-20:-10]), sum(recent[-10:])
            if abs(a - b) <= 0.015 * a or now - t_w > 2.0:                  # settled (or give up after 2 s)
                break
        lat = []
        for _ in range(frames):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out = step()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
    finally:
        gc.unfreeze()
    ms = 1e3 * sum(lat) / len(lat)
    ls = sorted(lat)
    res = {"workload": workload, "fps": round(1e3 / ms, 1), "ms": round(ms, 4), "frames": frames,
           "latency_ms": {"p50": round(1e3 * ls[len(ls) // 2], 4), "p95": round(1e3 * ls[int(len(ls) * 0.95)], 4), "max": round(1e3 * ls[-1], 4)},
           "protocol": "per-frame synchronize (run.py:62-76), default kernel options, 4 resident batches; fps = 1 / mean latency"}
    cas = cfg.cas
    n_rays = {i: int(out[f"depth_level{i}"].shape[1]) for i in range(cas.num) if cas.render_if[i]}
    saved = net.options
    net.options = Options(single_stream=1)
    timer = StageTimer()
    net._timer = timer
    for _ in range(12):
        step()
    torch.cuda.synchronize()
    net._timer = None
    res["stages_ms"] = {k: round(v, 4) for k, v in timer.summary().items()}
    try:
        table = kernel_time_table(lambda n: [step() for _ in range(n)])
    except Exception as e:                                  # profiler missing / refused: keep the line
        table = []
        res["kernel_table_error"] = str(e)[:160]
    net.options = saved
    if table:
        res["top_kernels"] = [{"kernel": k[:80], "launches_per_frame": round(c / 8, 2), "avg_us": round(a, 1), "share": round(sh, 4)}
                              for k, c, a, sh in table[:5]]
        k, c, a, sh = table[0]
        roof = {"kernel": k[:120], "share_of_kernel_time": round(sh, 4), "avg_launch_ms": round(a / 1e3, 4),
                "avg_launch_ms_source": "torch.profiler (roctracer) device time over 8 frames with enerf_options_t.single_stream = 1 "
                                        "(every kernel alone)"}
        wm = kernel_work_model(k, cfg, S, H, W, n_rays)
        if wm is not None:
            ach = wm[1] / (a * 1e-6) / 1e12
            roof.update({"bound": wm[0], "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "algorithmic_flops_per_launch": wm[1], "traffic": None})
        else:
            roof.update({"bound": None, "frac": None, "note": "no work model for this kernel (latency-bound gather / glue)"})
        e = pmc_lookup(pmc, name, k.split("(")[0])
        if e is not None:                                   # HBM bytes of one launch from the line's live PMC child (live_pmc)
            roof.update({"traffic": e["hbm_bytes_per_launch"], "traffic_measured_live": True, "mfma_pipe_busy_frac_pmc": e.get("mfma_busy_frac"),
                         "traffic_source": "this run's rocprofv3 --pmc child (bench.py --pmc-sequence: FETCH_SIZE, WRITE_SIZE, "
                                           "SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE in separate passes); bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB"})
        res["roofline"] = roof
    t = res["stages_ms"].get("feature_net")
    if t:
        gf = 14.64 * (S * H * W) / (3 * 512 * 640)
        res["feature_net_frac"] = round(gf / t / PEAK_F32_MFMA_TFLOPS, 4)
    del net, batches
    torch.cuda.empty_cache()
    return res


def train_child(timeout_s, live_pmc_ok=False, perceptual=False):
    """Config 5 inside the default run: `bench.py --train --no-perceptual --steps 20 --warmup 3` as a child process (its own
    line: ms_per_step, roofline of k_mlp_bwd, cpu_baseline of the oracle's train_step).  perceptual=True: the same step with the
    reference's full loss (the VGG16 term on, `bench.py --train`), 20 replays, ms_per_step only."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--train", "--steps", "20", "--warmup", "3"]
    if perceptual:
        cmd += ["--no-stages", "--no-cpu-baseline", "--no-live-pmc"]
    else:
        cmd.append("--no-perceptual")
    if not live_pmc_ok and not perceptual:                # (the rocprofv3 --pmc child passes of roofline.traffic cost ~25 s more)
        cmd.append("--no-live-pmc")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": f"{type(e).__name__}: {str(e)[:160]}"}
    return {"ms_per_step": round(d["ms_per_step"], 3), "samples_per_s": round(d["value"], 2), "steps": d["steps"],
            "roofline": d.get("roofline"), "cpu_baseline": d.get("cpu_baseline"), "step_launch": d["config"]["step_launch"],
            "loss": ("MSE + 0.01 x VGG16 perceptual L1 at both levels (losses/enerf.py:16-38; the VGG16 architecture with seeded random-init "
                     "weights: no pretrained weights offline)") if perceptual else
                    ("MSE at both levels (the VGG16 perceptual term off; `workloads.train_perceptual` is the step with the term on)"),
            "final_loss": d.get("final_loss"), "child_wall_s": round(time.perf_counter() - t0, 1),
            "workload": d["config"]["workload"]}


def make_workload(name, seed):
    from enerf_amd.config import EnerfConfig
    from enerf_amd.synth import make_batch, make_lego_batch, make_zju_batch
    rank = seed
    if name == "tiny":                                 # launcher tests on the CPU lane emulator only (--emu)
        cfg = EnerfConfig().with_cas(volume_planes=(8, 8), render_if=(False, True))
        return cfg, make_batch(32, 64, 3, cfg, seed=rank, textured=True), False, "32x64 launcher-test frame (CPU lane emulator)"
    if name == "dtu":
        cfg = EnerfConfig.dtu_eval()
        return cfg, make_batch(512, 640, 3, cfg, seed=rank, textured=True), False, \
            "DTU generalizable eval (dtu_pretrain.yaml, render_if False,True, volume_planes 48,8), 512x640, 3 src views, one target view per step"
    if name == "lego":
        cfg = EnerfConfig()
        return cfg, make_lego_batch(800, 800, 4, cfg, seed=rank), False, \
            "NeRF-Synthetic lego (configs/enerf/nerf/lego.yaml: render_if True,True, volume_planes 64,8), 800x800, 4 src views"
    if name == "zju":
        cfg = EnerfConfig().with_cas(volume_planes=(32, 8), render_if=(False, True))
        return cfg, make_zju_batch(1024, 1024, 4, cfg, seed=rank), True, \
            "ZJU-MoCap CoreView_313 (zjumocap_eval.yaml at input_ratio 1.0: network_human, volume_planes 32,8, render_if False,True), 1024x1024, 4 src views, mask_at_box"
    raise SystemExit(f"unknown workload {name}")


def emit_line(result):
    """Rank 0's ONE JSON line, as the LAST line of stdout: RCCL prints a version banner through C stdio (block-buffered on a
    pipe, so it would surface at process exit, after Python's line) — flush C stdio first, then print."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(result), flush=True)


class _PerceptualVGG16(torch.nn.Module):
    """The trainer's perceptual term (losses/vgg_perceptual_loss.py:4-46 as used by losses/enerf.py:30-38): L1 distances
    between the VGG16 ``features[:4] / [4:9] / [9:16] / [16:23]`` activations (conv1_2, conv2_2, conv3_3, conv4_3 after
    ReLU) of the rendered and the target image, both normalised with the ImageNet mean/std.  torchvision and its pretrained
    weights are not available offline: the architecture is restated here with RANDOM-INIT (seeded, frozen) weights — the
    cost of the term (forward of both images, backward into the rendering) is config 5's, its value is not.
    PyTorch-ROCm/MIOpen, like the trainer's; ``reduce`` is the sum used for the L1 means (train_graph.tree_sum under
    capture)."""

    def __init__(self, reduce):
        super().__init__()
        cfgs = [[(3, 64), (64, 64)], [(64, 128), (128, 128)], [(128, 256), (256, 256), (256, 256)],
                [(256, 512), (512, 512), (512, 512)]]
        gen = torch.Generator().manual_seed(16)
        blocks = []
        for bi, convs in enumerate(cfgs):
            layers = [torch.nn.MaxPool2d(2, 2)] if bi else []
            for cin, cout in convs:
                c = torch.nn.Conv2d(cin, cout, 3, padding=1)
                with torch.no_grad():
                    c.weight.copy_(torch.randn(c.weight.shape, generator=gen) * (2.0 / (cin * 9)) ** 0.5)
                    c.bias.zero_()
                layers += [c, torch.nn.ReLU(inplace=True)]
            blocks.append(torch.nn.Sequential(*layers))
        self.blocks = torch.nn.ModuleList(blocks).eval()
        for p in self.parameters():
            p.requires_grad = False
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))
        self.reduce = reduce

    def forward(self, inp, tar):
        x, y = (inp - self.mean) / self.std, (tar - self.mean) / self.std
        loss = 0.0
        for blk in self.blocks:
            x = blk(x)
            with torch.no_grad():
                y = blk(y)
            loss = loss + self.reduce((x - y).abs()) / x.numel()          # F.l1_loss, mean reduction
        return loss


def train_extras(args, net, batch, loss_fn, cfg, dev, ms_per_step):
    """`roofline` + `cpu_baseline` of the training line (rank 0, after the timed region).
    roofline: the step's dominant kernel, k_mlp_bwd<R=3,S> of the last cascade level (MFMA-bound: fused recompute-forward +
    backward of the Agg/NeRF MLP).  Its launch arguments are recorded from one eager step of this very batch and the kernel is
    then re-launched alone on the bench's stream between HIP events (inside the captured step no event can bracket a node);
    the per-step rocprofv3 table (profiles/r04_train_step_kernels_*.csv) holds the same kernel's duration inside the replays.
    cpu_baseline: oracle.train_step (forward + MSE + backward of the reference-equivalent CPU restatement, BatchNorm batch
    statistics; pinned to the reference's gradients by tests/test_oracle_golden.py) on the host cores, ONE step."""
    lib = net.lib
    rec = {}
    orig = lib.nerf_mlp_bwd

    def spy(*a, **kw):                   # (vox, x, g_raw, packed, bimg, offsets, S, F, level=...)
        rec[a[7]] = (a, kw)
        return orig(*a, **kw)
    lib.nerf_mlp_bwd = spy
    try:
        for p_ in net.parameters():
            p_.grad = None
        loss_fn(net(batch), batch).backward()
    finally:
        lib.nerf_mlp_bwd = orig
    for p_ in net.parameters():
        p_.grad = None
    out = {}
    last = cfg.cas.num - 1
    F = cfg.cas.nerf_model_feat_ch[last] + 3
    if F in rec:
        a, kw = rec[F]
        P, S, R = int(a[0].shape[0]), int(a[6]), (F + 3) // 4
        for _ in range(3):
            orig(*a, **kw)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        ev[0].record()
        for i in range(10):
            orig(*a, **kw)
            ev[i + 1].record()
        torch.cuda.synchronize()
        t_ms = sum(ev[i].elapsed_time(ev[i + 1]) for i in range(10)) / 10
        tiles = mlp_bwd_mfma_tiles_per_16(S, R)
        flops = tiles * 2 * 16 * 16 * 4 * ((P + 15) // 16)
        ach = flops / (t_ms * 1e-3) / 1e12
        out["roofline"] = {
            "kernel": f"k_mlp_bwd<{R},{S}> (level-{last} Agg + NeRF MLP: fused recompute-forward + backward" +
                      (" + the per-view layers' weight gradients" if R == 3 else "") + f", {P} points)",
            "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None, "avg_launch_ms": round(t_ms, 4),
            "avg_launch_ms_source": "HIP events around 10 stand-alone launches on the bench's stream with the launch arguments "
                                    "of this batch's step (a node inside the captured step cannot be bracketed)",
            "algorithmic_flops_per_launch": flops, "mfma_tiles_per_16_points": tiles,
            "share_of_step": round(t_ms / ms_per_step, 4),
            "note": "issued fp32 16x16x4 MFMA work (static v_mfma count of the kernel = tiles per 16 points); one wave per SIMD at "
                    "256 VGPRs + AGPR traffic is why the fraction is low (DESIGN.md)",
            "weight_gradients_in_kernel": kw.get("level", 0)}
        if not args.no_live_pmc and not args.emu:
            # HBM bytes of one launch, measured now: rocprofv3 --pmc passes over two EAGER steps of a child run (the same kernels
            # as the captured step; counters serialise the kernels either way)
            flags = ["--train", "--train-eager", "--steps", "2", "--warmup", "1"] + (["--no-perceptual"] if args.no_perceptual else [])
            live = live_pmc([], kernel_prefix=f"k_mlp_bwd<{R}, {S}", child_flags=flags)       # (<R, S> or <R, S, level>)
            live = next(iter(live[""].values())) if live and live.get("") else None
            if live is not None:
                out["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = ("live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over a child run of "
                                                     "bench.py --train --train-eager; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB")
                if "mfma_busy_frac" in live:
                    out["roofline"]["mfma_busy_frac"] = live["mfma_busy_frac"]
    if not args.no_cpu_baseline:
        import numpy as np   # noqa: F401
        from oracle import enerf_oracle as O
        ncores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(ncores)
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        cb = {k: v.detach().cpu() for k, v in batch.items()}
        t0 = time.perf_counter()
        O.train_step(cfg, sd, cb)
        cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "samples/s", "cores": ncores, "kind": "port",
                               "ref_over_port": REF_OVER_PORT["train"],
                               "sample": f"1 training step (forward + MSE loss + backward, no optimizer update) of this batch through "
                                         f"oracle/enerf_oracle.py::train_step (torch CPU, {ncores} of {os.cpu_count()} host threads)"}
    return out


def collective_stack_note():
    """RCCL version and the peer-access matrix of the visible devices (hipDeviceCanAccessPeer: xGMI / PCIe P2P), for stderr."""
    import torch
    try:
        ver = ".".join(map(str, torch.cuda.nccl.version()))
    except Exception as e:                                               # noqa: BLE001 — a note, never a failure
        ver = f"unknown ({e})"
    n = torch.cuda.device_count()
    rows = []
    for i in range(n):
        row = ""
        for j in range(n):
            try:
                row += "-" if i == j else ("1" if torch.cuda.can_device_access_peer(i, j) else "0")
            except Exception:                                            # noqa: BLE001
                row += "?"
        rows.append(row)
    return f"RCCL {ver}; {n} visible device(s); peer access (hipDeviceCanAccessPeer) rows = from, columns = to: {' '.join(rows)}"


def train_bench(args, rank, world, dev, dist, emu_lib=None):
    """Config 5: trainer.py:56-63 on the drop-in network — forward (train mode, BN batch statistics),
    the MSE part of losses/enerf.py:21-24, backward (DDP gradient all-reduce over RCCL when world > 1),
    clip_grad_value_(40), Adam step.  Data-parallel: one sample per GPU per step (dtu_pretrain.yaml:60), weak scaling."""
    import numpy as np
    import torch.nn.functional as F
    from __graft_entry__ import _seeded_network
    from enerf_amd.config import EnerfConfig
    from enerf_amd.synth import make_batch
    cfg = EnerfConfig()                                                  # dtu_pretrain.yaml: planes 64,8, render_if True,True
    emu = emu_lib is not None
    H, W = (32, 64) if emu else (512, 640)
    if emu:     # launcher / step-structure dry run on the CPU lane emulator over gloo: the step the graph would capture, run eagerly
        cfg = cfg.with_cas(volume_planes=(8, 8))
    net = _seeded_network(cfg, dev, lib=emu_lib).train()
    model = net
    graphed = not args.train_eager and not emu                           # one hipGraph replay per step (train_graph.py)
    device_sync = (lambda: None) if emu else torch.cuda.synchronize
    dp = world > 1 or dist is not None                                   # --train-dp1: the data-parallel step on a 1-rank group
    if dp and world == 1:
        from enerf_amd import autograd as _A
        _A.SYNC_SINGLE_RANK = True
    if dp:   # (every BatchNorm runs inside the HIP training functions, whose statistics exchange works on gloo as well)
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)         # trainer.py:16
        model = net
        if not graphed and not emu:                                      # trainer.py:17-22 as written: the eager DDP step
            from torch.nn.parallel import DistributedDataParallel as DDP
            model = DDP(net, device_ids=[dev.index], output_device=dev.index, find_unused_parameters=True)
    # torch.optim.Adam as the trainer builds it (lib/train/optimizer.py); on the GPU its fused multi-tensor implementation (the
    # same update rule in ONE kernel per dtype/device group; the default capturable foreach path issues ~220 scalar-base pow
    # kernels per step — 1 ms of a 19 ms step, profiles/r04_train_step_profile_foreach_adam.txt)
    adam_kw = {"capturable": graphed}
    if not emu and not args.adam_foreach:
        adam_kw["fused"] = True
    opt = torch.optim.Adam(net.parameters(), lr=5e-4, **adam_kw)
    b = make_batch(H, W, 3, cfg, seed=rank, textured=True)
    rng = np.random.default_rng(rank)
    for i in range(2):
        b[f"rgb_{i}"] = rng.uniform(0, 1, size=(1, b[f"rays_{i}"].shape[1], 3)).astype(np.float32)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}

    from enerf_amd.train_graph import GraphedTrainStep, GraphMismatch, mse_loss as tree_mse
    mse = tree_mse if graphed else F.mse_loss          # same value; the tree form keeps memset nodes out of the graph

    from enerf_amd.train_graph import tree_sum
    perceptual = None if (args.no_perceptual or emu) else _PerceptualVGG16(tree_sum if graphed else torch.sum).to(dev)

    def loss_fn(out, bt):                                                # losses/enerf.py:16-38 (train_img True,True)
        loss = 0.0
        for i, w in enumerate((0.1, 1.0)):
            loss = loss + w * mse(bt[f"rgb_{i}"], out[f"rgb_level{i}"])
            if perceptual is not None:
                hi, wi = int(H * cfg.cas.render_scale[i]), int(W * cfg.cas.render_scale[i])
                img = lambda t: t.reshape(-1, hi, wi, 3).permute(0, 3, 1, 2)
                loss = loss + 0.01 * w * perceptual(img(out[f"rgb_level{i}"]), img(bt[f"rgb_{i}"]))
        return loss

    def step():
        out = model(batch)
        loss = loss_fn(out, batch)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_value_(net.parameters(), 40)
        opt.step()
        return loss
    launch_note = "eager"
    if emu:         # the step the graph captures (train_graph.train_step + FlatGradSync), enqueued eagerly on the emulator
        from enerf_amd.train_graph import FlatGradSync, train_step
        sync = FlatGradSync(net) if world > 1 else None
        if sync is not None:
            sync.broadcast()
        step = lambda: train_step(net, opt, loss_fn, batch, 40.0, sync)
        launch_note = "DRY RUN on the CPU lane emulator over gloo: train_graph.train_step with the flat gradient all-reduce, not captured"
    if graphed:
        # fallback="eager": a capture the stack refuses (first contact with N > 1: >= 35 RCCL nodes per step) or a replay that fails
        # the verification on ANY rank costs the graph, not the job — the verdict is one MAX all-reduce, so every rank keeps
        # training with the same eager train_step (flat gradient all-reduce + per-layer SyncBatchNorm exchanges) and says so
        gstep = GraphedTrainStep(net, opt, loss_fn, batch, clip_value=40.0, distributed=dp, fallback="eager")
        step = lambda: gstep(batch)                                      # copies the batch in, camera tables, one replay
        launch_note = gstep.step_launch + (" (enerf_amd/train_graph.py)" if gstep.graph is not None else "") + \
            (": the flat gradient all-reduce and the SyncBatchNorm statistics exchanges are graph nodes" if dp and gstep.graph is not None else "")
    import gc
    gc.collect()                                                         # before the warm-up: no idle gap in front of the timed region,
    gc.disable()                                                         # no collector pause inside it (see the rendering region in main())
    try:
        for _ in range(args.warmup):
            step()
        device_sync()
        if dist is not None:
            dist.barrier()
        device_sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        device_sync()
        if dist is not None:
            dist.barrier()
        device_sync()
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    extra = {}
    if rank == 0 and not emu and not args.no_stages:
        extra = train_extras(args, net, batch, loss_fn, cfg, dev, 1e3 * elapsed / args.steps)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        emit_line(({**extra,
            "metric": "training samples/sec (dtu_pretrain, 512x640, 3 src views, full-image rays at both levels)",
            "value": world * args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not emu else "synthetic (CPU lane emulator, 32x64: launcher check, not a measurement)", "final_loss": float(loss.detach()),
            "collective_ranks_seen": args.binding["ranks_seen"], "collective_backend": args.binding["backend"],
            "rank_devices": [{k: b.get(k) for k in ("rank", "device", "visible", "hw", "affinity")} for b in args.binding["bindings"]],
            "config": {"workload": "BASELINE config 5: DTU dtu_pretrain training, one sample per GPU per step, MSE loss "
                                   "(losses/enerf.py:21-24)" + (" + 0.01 x VGG16 perceptual L1 at both levels (losses/enerf.py:30-38; the "
                                   "architecture with seeded random-init weights: no pretrained weights offline)" if perceptual is not None
                                   else " (the VGG perceptual term switched off)") + ", Adam, clip_grad_value_ 40", "parallelism": (f"data-parallel x{world} + SyncBatchNorm over {'gloo' if emu else 'RCCL'} (" + ("DistributedDataParallel" if model is not net else "one flat gradient all-reduce per step") + ")") if dp else "single GPU",
                       "step_launch": launch_note,
                       "optimizer": "torch.optim.Adam(" + ", ".join(f"{k}={v}" for k, v in adam_kw.items()) + ")",
                       "backward": "HIP forward+backward for every stage of the network: FeatureNet (MFMA conv / dgrad incl. the stride-2 5x5 layers / wgrad, BN-train, upsampling adjoint, channels-last), cost-reg nets (MFMA conv/dgrad/wgrad, BN-train), get_depth_values / build_rays / sample_along_depth, camera tables, Agg+NeRF MLP (fused), warp+variance, depth regression, compositing, render-side fetches; PyTorch ops left in a step: the trainer's loss, clip_grad_value_, Adam, and autograd's own gradient accumulation"}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["dtu", "lego", "zju"], default="dtu")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true", help="skip everything after the timed region (profiling runs)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the rocprofv3 --pmc child passes for roofline.traffic (replay profiles/pmc_render_<workload>.json)")
    ap.add_argument("--no-sync-per-frame", action="store_true",
                    help="profiling runs: enqueue the timed frames back to back (value is then sequential_fps)")
    ap.add_argument("--in-flight", type=int, default=6, help="frames in flight for the pipelined_fps extra")
    ap.add_argument("--sustained-frames", type=int, default=2000)
    ap.add_argument("--graph", action="store_true", help="time whole-frame HIP graph replays (enerf_amd/graph.py)")
    ap.add_argument("--single-stream", action="store_true",
                    help="enerf_options_t.single_stream=1: no side lane inside the frame (per-kernel PMC passes want kernels alone)")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE config 5 instead of rendering: one step = forward + MSE loss + backward + Adam step of "
                         "dtu_pretrain (512x640, 3 views, full-image rays at both levels, bs 1 per GPU), DDP over RCCL for N > 1")
    ap.add_argument("--train-eager", action="store_true", help="--train: enqueue every step eagerly instead of one graph replay")
    ap.add_argument("--no-perceptual", action="store_true", help="--train: leave the VGG16 perceptual term out of the loss")
    ap.add_argument("--adam-foreach", action="store_true", help="--train: torch.optim.Adam's foreach implementation instead of the fused one")
    ap.add_argument("--train-dp1", action="store_true",
                    help="--train --gpus 1: take the DATA-PARALLEL step on a 1-rank RCCL group (SyncBatchNorm conversion, statistics "
                         "exchanges and the flat gradient all-reduce captured as graph nodes) — what a 1-GPU box can measure of the N > 1 step")
    ap.add_argument("--feature-backend", choices=["hip", "torch"], default="hip",
                    help="FeatureNet on the HIP matrix-core path (default) or in PyTorch-ROCm/MIOpen (north_star's split)")
    ap.add_argument("--options", default="",
                    help="A/B and profiling runs: enerf_options_t fields as 'field:value,field:value' (recorded in config.options); "
                         "the default line is measured with all-zero options")
    ap.add_argument("--pmc-sequence", default="",
                    help="internal (live_pmc's child): render three frames of each listed workload, kernels alone, bracketed by sentinel launches")
    ap.add_argument("--batches", type=int, default=4,
                    help="distinct seeded input batches the timed frames rotate over (all uploaded before the timed region)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `workloads` extra (lego, zju, training step) of the default dtu run")
    ap.add_argument("--time-budget", type=float, default=105.0,
                    help="wall-clock budget (s) of the default run: extras that would start after it are skipped and say so")
    ap.add_argument("--emu", action="store_true",
                    help="launcher/CI check WITHOUT a GPU: run the same bench flow on the CPU lane emulator of the kernel "
                         "sources (tests/emu) over gloo with a 32x64 frame; the numbers mean nothing")
    args = ap.parse_args()
    t_process = time.perf_counter()

    # ---- N > 1 without a launcher: become one.  `python bench.py --gpus 8` must not silently measure one GPU. ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        if not args.emu and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible: refusing to "
                             f"report an n_gpus={args.gpus} line")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}: refusing to print a line whose n_gpus is not --gpus")
    emu_lib = None
    if args.emu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu_lib import emu_lib as _emu
        emu_lib = _emu()
        args.workload, args.no_stages, args.no_cpu_baseline = "tiny", True, True
        os.environ.setdefault("ENERF_EMU_THREADS", "2")
        torch.set_num_threads(1)
        dev = torch.device("cpu")
        device_sync = lambda: None
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        device_sync = torch.cuda.synchronize
    dist = None
    if world > 1 or (args.train and args.train_dp1):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))   # (set by the launcher for N > 1)
        if args.emu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL

    from __graft_entry__ import _seeded_network
    from enerf_amd.frame_parallel import rank_bindings, render_sharded

    # one process per GPU, checked: every rank's device binding is exchanged, duplicates are refused on all ranks, and the
    # size of the communicator is counted on the devices (`collective_ranks_seen`, next to n_gpus in the line)
    try:
        args.binding = rank_bindings(rank, world, local, dev, pin=world > 1)     # N > 1: one core set per rank (inherited by children)
    except RuntimeError as e:
        raise SystemExit(f"rank {rank}: {e}: refusing to print an n_gpus={world} line")
    print(f"[bench] rank {rank}/{world} local_rank {local} -> {args.binding['bindings'][rank if world > 1 else 0]}", file=sys.stderr, flush=True)
    if world > 1 and rank == 0 and dev.type == "cuda":                   # first contact with a multi-GPU node: what the job runs on
        print(f"[bench] {collective_stack_note()}", file=sys.stderr, flush=True)
    if args.binding["ranks_seen"] != world:
        raise SystemExit(f"communicator holds {args.binding['ranks_seen']} ranks, --gpus says {world}")

    if args.train:
        return train_bench(args, rank, world, dev, dist, emu_lib)
    if args.pmc_sequence:
        return pmc_sequence_child([w for w in args.pmc_sequence.split(",") if w], dev)
    nb = max(1, args.batches)
    cfg, batch_np, human, workload = make_workload(args.workload, rank)
    cas = cfg.cas
    net = _seeded_network(cfg, dev, human=human, feature_backend=args.feature_backend, lib=emu_lib)
    _, S, _, H, W = batch_np["src_inps"].shape
    batch = {k: torch.from_numpy(v).to(dev) for k, v in batch_np.items()}
    # the timed frames rotate over nb distinct batches (seed = rank + 1000*j), all resident before the timed region: a frame
    # never finds its own inputs / rays of the previous frame in MALL or L2
    batches = [batch] + [{k: torch.from_numpy(v).to(dev) for k, v in make_workload(args.workload, rank + 1000 * j)[1].items()}
                         for j in range(1, nb)]
    last = cas.num - 1
    opt_fields = {k: int(v) for k, v in (f.split(":") for f in args.options.split(",") if f)}
    if args.single_stream:
        opt_fields["single_stream"] = 1
    if opt_fields:
        from enerf_amd.lib import Options
        net.options = Options(**opt_fields)

    frame_no = [0]
    if args.graph:
        from enerf_amd.graph import GraphedFrame
        net.static_shapes = True
        frame = GraphedFrame(net, batch)

        def step():
            frame_no[0] += 1
            return frame(batches[frame_no[0] % nb])
    else:
        def step():
            frame_no[0] += 1
            return net(batches[frame_no[0] % nb])

    out = None

    window = []                                         # per-frame wall times of the timed window (diagnostics: `timed_window_ms`)

    def timed_frame(_f):                                # run.py:62-67: sync, network(batch), sync
        nonlocal out
        t_f = time.perf_counter()
        out = step()
        if not args.no_sync_per_frame:
            device_sync()
        window.append(time.perf_counter() - t_f)
        return None

    for _ in range(args.warmup):
        step()
    # steady state before the timed region whatever --warmup was (the driver's 20-step / small-warm-up run
    # timed the clock ramp and first-touch effects: 1094 vs 1144 frames/s): at least 0.3 s and 100 frames of untimed work in total
    # The timed region of the driver's command is 20 frames = 15 ms.  Two host-side effects were measured to move it by 6 - 25 % while
    # the kernels did not change (profiles/r05_run13 .. run15: 1240 / 1218 / 998 frames/s in the window against a 200-frame latency
    # p50 of 0.760 ms = 1316/s in the same process): a pass of Python's collector inside the window, and — worse — ANY idle gap in
    # front of it (a collection, a first-time import): the device leaves its steady clocks and the next ~15 frames run 1.11 -> 0.96
    # of a millisecond on their way back.  So: collect BEFORE the warm-up, keep the collector off until the window closes, import
    # what the window needs now, and let the warm-up run until the frame time has settled (the last 10 frames within 1.5 % of the 10
    # before them; at most 2 s).  Nothing of a frame is skipped; the warm-up is untimed by contract.
    import gc
    import torch.distributed as _td  # noqa: F401  (render_sharded imports it: not inside the gap)
    gc.collect()
    gc.disable()
    try:                                                  # (whatever happens in the timed region: the collector comes back)
        internal_warmup = 0
        if not args.emu:
            device_sync()
            t_w = time.perf_counter()
            recent = []
            while True:
                t_f = time.perf_counter()
                step()
                device_sync()
                now = time.perf_counter()
                recent.append(now - t_f)
                internal_warmup += 1
                if internal_warmup + args.warmup < 100 or now - t_w < 0.3:
                    continue
                a, b = sum(recent[-20:-10]), sum(recent[-10:])
                if abs(a - b) <= 0.015 * a or now - t_w > 2.0:
                    break
        device_sync()
        if dist is not None:
            # N > 1: the ranks leave their warm-ups at different moments, and the first to reach the timed region's barrier would idle
            # there (the same gap, seen from its device).  Meet once HERE, run a few more untimed frames from a common start, and the
            # barrier in front of the window finds every rank within a frame or two of the others.
            dist.barrier()
            for _ in range(2 if args.emu else 40):
                step()
                device_sync()
        # frame f -> rank f mod world: every rank renders `steps` frames; barrier + sync both sides, MAX over ranks
        t_rank = time.perf_counter()
        _, fps, elapsed = render_sharded(timed_frame, args.steps * world, rank, world, sync=device_sync)
        t_rank = time.perf_counter() - t_rank
    finally:
        gc.enable()
    assert bool(torch.isfinite(out[f"rgb_level{last}"]).all()), "non-finite render"
    per_rank = [args.steps / t_rank]
    if dist is not None:                                # every rank's own rate next to the aggregate (reporting only)
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered

    result = None
    if rank == 0:
        result = {
            "metric": f"rendered frames/sec @{H}x{W} {S}-src-view", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": (fps / BASELINE_FPS_RTX3090) if args.workload == "dtu" else None,
            "dtype": "f32", "data": "synthetic" if not args.emu else "synthetic (CPU lane emulator: launcher check, not a measurement)",
            "per_rank_fps": [round(v, 2) for v in per_rank], "internal_warmup_frames": internal_warmup,
            "timed_window_ms": [round(1e3 * v, 4) for v in window[:64]],
            "collective_ranks_seen": args.binding["ranks_seen"], "collective_backend": args.binding["backend"],
            "rank_devices": [{k: b.get(k) for k in ("rank", "device", "visible", "hw", "affinity")} for b in args.binding["bindings"]],
            "config": {"workload": workload, "distinct_batches": nb, "feature_net": args.feature_backend, "hip_graph": bool(args.graph),
                       "single_stream": bool(args.single_stream), "options": opt_fields,
                       "protocol": "per-frame synchronize (run.py:62-76), one frame at a time, default kernel options "
                                   "(inside the frame the FeatureNet's top-down half runs on the library's side stream); untimed "
                                   "warm-up = --warmup frames + frames until the last 10 frame times are within 1.5 % of the 10 before "
                                   "them (>= 100 frames and 0.3 s, at most 2 s; N > 1: 40 more from a common barrier) — "
                                   "`internal_warmup_frames`"
                                   if not args.no_sync_per_frame else "frames enqueued back to back on one stream",
                       "frames_per_step_per_gpu": 1, "parallelism": f"frame-parallel x{world} (no collectives)"},
        }
        if args.workload == "dtu":
            result["vs_baseline_note"] = ("value / 21.78 FPS (README.md:121, one RTX 3090, the same per-frame-sync protocol); "
                                          "for N > 1 the numerator is the whole-job aggregate")

    secondary = (rank == 0 and world == 1 and args.workload == "dtu" and not args.no_stages and not args.no_secondary
                 and not args.graph and not args.emu and args.feature_backend == "hip" and not opt_fields)
    pmc_all = None
    # ---- extras on rank 0 (not in the timed region) ----
    if rank == 0 and not args.no_stages and not args.graph:
        t0 = time.perf_counter()                      # host-side enqueue cost (no device sync inside)
        for _ in range(50):
            step()
        result["host_enqueue_ms_per_step"] = round(1e3 * (time.perf_counter() - t0) / 50, 4)
        torch.cuda.synchronize()
        # the reference's protocol once more, as a distribution
        lat = []
        for _ in range(200):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            lat.append(1e3 * (time.perf_counter() - t1))
        lat.sort()
        result["latency_ms"] = {"p50": round(lat[len(lat) // 2], 4), "p95": round(lat[int(len(lat) * 0.95)], 4),
                                "mean": round(sum(lat) / len(lat), 4), "protocol": "sync per frame (run.py:62-76), 200 frames"}
        # the same frames strictly one after the other on one stream, no per-frame sync
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(200):
            step()
        torch.cuda.synchronize()
        result["sequential_fps"] = round(200 / (time.perf_counter() - t1), 1)
        # per-stage HIP events recorded inside enerf_forward.  The default frame forks the FeatureNet's top-down half onto
        # the library's side lane (enerf_options_t.single_stream = 0), so stage intervals on the caller's stream overlap it:
        # `stages_ms` is the attribution pass with single_stream = 1 (every kernel in order on one stream), and
        # `stages_ms_default` the intervals of the frame `value` is timed on (the render interval starts after the join,
        # so the FINAL render is the kernel alone either way — the roofline below uses its default-mode figure).
        from enerf_amd.lib import Options

        def stage_pass(opt):
            timer = StageTimer()
            net._timer = timer
            saved = net.options
            net.options = opt
            for _ in range(30):
                step()
            torch.cuda.synchronize()
            net._timer, net.options = None, saved
            return timer.summary()
        stages_default = stage_pass(net.options)
        stages = stage_pass(Options(single_stream=1))
        result["stages_ms"] = {k: round(v, 4) for k, v in stages.items()}
        result["stages_ms_default"] = {k: round(v, 4) for k, v in stages_default.items()}
        result["stages_note"] = ("stages_ms: enerf_options_t.single_stream=1 (sequential attribution); stages_ms_default: the "
                                 "default frame, FeatureNet top-down half overlapped with level 0 on the side lane")
        # the final render launch is measured in the mode `value` runs in (it starts after every join: the kernel alone);
        # a non-final level's render overlaps the next level in that mode, so its single-stream figure is the kernel's own
        stages[f"render_{last}"] = stages_default[f"render_{last}"]

        # sustained legs (>= 2000 frames: long enough for utilisation sampling) and the pipelined throughput
        n_sus = max(0, args.sustained_frames)
        if n_sus:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_sus):
                step()
            torch.cuda.synchronize()
            sus = {"frames": n_sus, "sequential_fps": round(n_sus / (time.perf_counter() - t1), 1)}
            from enerf_amd.pipeline import FramePipeline
            if human:
                net.static_shapes = True              # no count readback: frames can be in flight
            pipe = FramePipeline(net, depth=max(1, args.in_flight), throughput_tuning=True)
            for _ in range(20):
                pipe.submit(batch)
            pipe.join()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_sus):
                pipe.submit(batch)
            pipe.join()
            torch.cuda.synchronize()
            sus["pipelined_fps"] = round(n_sus / (time.perf_counter() - t1), 1)
            pipe.close()
            net.static_shapes = False
            result["sustained"] = sus
            result["pipelined_fps"] = {"value": sus["pipelined_fps"], "frames_in_flight": max(1, args.in_flight),
                                       "options": "enerf_options_t{single_stream=1}",
                                       "note": "throughput of frames in flight on separate HIP streams; NOT the reference's "
                                               "protocol, never `value`"}
            if args.workload == "dtu":
                result["vs_baseline_pipelined"] = round(sus["pipelined_fps"] / BASELINE_FPS_RTX3090, 2)

        # ---- north_star's split kept visible: the same frames with the FeatureNet in PyTorch-ROCm (MIOpen), NCHW maps handed
        #      through the C ABI (feature_backend="torch"); an extra key, never `value` ----
        if args.feature_backend == "hip" and not human:
            try:
                net_t = _seeded_network(cfg, dev, human=human, feature_backend="torch")
                for _ in range(10):
                    net_t(batches[0])
                torch.cuda.synchronize()
                lt = []
                for f in range(60):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    net_t(batches[f % nb])
                    torch.cuda.synchronize()
                    lt.append(time.perf_counter() - t1)
                result["feature_backend_torch"] = {"value": round(len(lt) / sum(lt), 1), "unit": "frames/s",
                                                   "note": "same protocol, FeatureNet in PyTorch-ROCm/MIOpen (north_star's split); 60 frames"}
                del net_t
            except Exception as e:                      # an extra: never let it take the line down
                result["feature_backend_torch"] = {"error": str(e)[:200]}

        # ---- the opt-in fast precision of the render MLP (enerf_options_t.render_precision = 2, "bf16x3": every fp32 operand as two
        #      bf16 pieces on the bf16 matrix cores; ~7e-6 of max|ref| on ordinary inputs, NOT robust to large head gains —
        #      tests/test_adversarial.py): an extra key, never `value` ----
        if not args.graph:
            try:
                from enerf_amd.lib import Options as _O
                saved = net.options
                net.options = _O(render_precision=2, **{k: v for k, v in opt_fields.items()})
                for _ in range(20):
                    step()
                torch.cuda.synchronize()
                lt = []
                for _ in range(200):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    o_fast = step()
                    torch.cuda.synchronize()
                    lt.append(time.perf_counter() - t1)
                key = f"rgb_level{last}"
                o_fast = {key: o_fast[key].clone()}
                net.options = saved
                o_exact = net(batches[frame_no[0] % nb])
                torch.cuda.synchronize()
                result["render_bf16x3"] = {"value": round(len(lt) / sum(lt), 1), "unit": "frames/s",
                                           "max_abs_rgb_vs_exact_kernel": float((o_fast[key] - o_exact[key]).abs().max()),
                                           "note": "same protocol, 200 frames, render_precision = 2 (opt-in: ~1e-5 operand error, "
                                                   "amplified by large head gains); the default and `value` are the exact fp32 kernel"}
            except Exception as e:
                result["render_bf16x3"] = {"error": str(e)[:200]}

        # ---- rooflines: per stage, and the dominant kernel as the contract's `roofline` object ----
        sr = {}
        for i in range(cas.num):
            D, h, w = cas.volume_planes[i], int(H * cas.volume_scale[i]), int(W * cas.volume_scale[i])
            C = 32 >> i
            t = stages.get(f"cost_reg_{i}")
            if t:
                gf = cost_reg_gflop(C, i != 0, D, h, w)
                sr[f"cost_reg_{i}"] = {"bound": "mfma", "achieved": round(gf / t, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                                       "unit": "TFLOP/s", "frac": round(gf / t / PEAK_F32_MFMA_TFLOPS, 4),
                                       "algorithmic_gflop": round(gf, 3)}
            t = stages.get(f"volume_{i}")
            if t:
                Hs, Ws = H >> (2 - i), W >> (2 - i)
                mb = (S * Hs * Ws * C + D * h * w * C + D * h * w) * 4 / 1e6       # features once + volume once + depth planes
                # The stage's roof is the vector-memory REQUEST rate (tools/micro/gather_rate.hip, profiles/r06_gather_rate_micro.txt): 4 bilinear
                # taps of C channels per voxel and view, 24.1 TB/s of requested bytes when the texels sit in L1 (8.5 TB/s from L2; loads in
                # flight per wave do not change it).  achieved = S * 4 * C * 4 bytes * voxels / time.
                req_gb = D * h * w * S * 4 * C * 4 / 1e9
                sr[f"volume_{i}"] = {"bound": "l1-gather", "achieved": round(req_gb / t, 2), "peak": PEAK_L1_GATHER_TBS, "unit": "TB/s requested",
                                     "frac": round(req_gb / t / PEAK_L1_GATHER_TBS, 4), "voxels_per_us": round(D * h * w / (t * 1e3), 1),
                                     "requested_gather_gbytes": round(req_gb, 4), "algorithmic_hbm_mbytes": round(mb, 2),
                                     "note": "k_feature_volume_mp: requested gather bytes against the measured L1-hit request rate of this access "
                                             "shape (64/CQ texels x CQ lanes x 16 B per wave instruction); the L2-hit rate is 8.5 TB/s, so the "
                                             "kernel runs on L1 line sharing between neighbouring voxels"}
            t = stages.get(f"render_{i}")
            if t and cas.render_if[i]:
                F = cas.nerf_model_feat_ch[i] + 3
                R = (F + 3) // 4
                n_rays = int(out[f"depth_level{i}"].shape[1])
                n_samples = n_rays * cas.num_samples[i]
                tiles = render_mfma_tiles_per_16(S, R)
                fl = tiles * 2 * 16 * 16 * 4 / 16.0 * n_samples
                ach = fl / (t * 1e-3) / 1e12
                shape = ("12,3 (lean: 3 waves/SIMD)" if cas.num_samples[i] <= 2 else "4,2") if R == 3 else "8,2"
                sr[f"render_{i}"] = {"kernel": f"k_render_rays<{R},{S},{shape}>", "bound": "mfma",
                                     "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "avg_launch_ms": round(t, 4),
                                     "mfma_tiles_per_16_samples": tiles, "samples": n_samples,
                                     "algorithmic_flops_per_launch": fl,
                                     "reference_dense_flops_per_launch": render_dense_flop_per_sample(S, F) * n_samples}
        t = stages.get("feature_net")
        if t and args.feature_backend == "hip":
            gf = 14.64 * (S * H * W) / (3 * 512 * 640)                                  # SURVEY.md §2.2 probe, per pixel
            sr["feature_net"] = {"bound": "mfma", "achieved": round(gf / t, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                                 "unit": "TFLOP/s", "frac": round(gf / t / PEAK_F32_MFMA_TFLOPS, 4),
                                 "algorithmic_gflop": round(gf, 3)}
        result["stage_roofline"] = sr
        dom = max((k for k in sr if k.startswith("render_")), key=lambda k: stages[k], default=None)
        if dom is not None:
            d = sr[dom]
            # PMC evidence (separate rocprofv3 --pmc passes, tools/collect_profiles.sh) is replayed from profiles/ and tied to
            # the kernel sources it was collected on: a kernel edit without a PMC refresh shows up as pmc_stale = true
            from enerf_amd.build import source_digest
            traffic, pmc_note, busy, pmc_digest = None, None, None, None
            lib_digest = source_digest()
            live = None
            if world == 1 and not args.no_live_pmc and dom == f"render_{last}" and cas.num_samples[last] <= 2:
                torch.cuda.synchronize()
                # ONE child per counter pass renders this workload and (default dtu run) the secondary ones: their rooflines read
                # `pmc_all` further down
                pmc_all = live_pmc([args.workload] + (["lego", "zju"] if secondary else []))
                live = pmc_lookup(pmc_all, args.workload, f"k_render_rays<{(cas.nerf_model_feat_ch[last] + 3 + 3) // 4}, {S}")
            if live is not None:
                traffic, busy, pmc_digest = live["hbm_bytes_per_launch"], live.get("mfma_busy_frac"), lib_digest
                pmc_note = ("measured by this run: rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE) "
                            "over a child run of bench.py --pmc-sequence (3 frames per workload, every kernel alone on one stream); "
                            "bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB")
            for cand in (() if live is not None else (f"pmc_render_{args.workload}.json", f"r02_pmc_render_{args.workload}.json")):
                pmc_path = os.path.join(ROOT, "profiles", cand)
                if os.path.exists(pmc_path):
                    pmc = json.load(open(pmc_path))
                    traffic, pmc_note, busy = pmc.get("hbm_bytes_per_launch"), pmc.get("source"), pmc.get("mfma_busy_frac")
                    pmc_digest = pmc.get("source_digest")
                    break
            result["roofline"] = {
                "kernel": d["kernel"] + f" (level-{dom[-1]} fused render: sample placement + gathers + Agg/NeRF MLP + compositing)",
                "bound": "mfma", "achieved": d["achieved"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": d["frac"], "traffic": traffic, "avg_launch_ms": d["avg_launch_ms"],
                "avg_launch_ms_source": "HIP events recorded around the launch inside enerf_forward, on its stream, default "
                                        "kernel options (the variant `value` is timed on)",
                "algorithmic_flops_per_launch": d["algorithmic_flops_per_launch"],
                "reference_dense_flops_per_launch": d["reference_dense_flops_per_launch"],
                "note": f"achieved = {d['mfma_tiles_per_16_samples']} fp32 16x16x4 MFMA tiles per 16 samples (the MLP with the "
                        "view-independent halves of global_fc/color.0 evaluated once per point) x samples / launch time",
                "traffic_measured_live": live is not None, "mfma_pipe_busy_frac_pmc": busy, "pmc_measured_live": live is not None,
                "traffic_source": pmc_note, "library_source_digest": lib_digest, "pmc_source_digest": pmc_digest,
                "pmc_stale": pmc_digest != lib_digest}

    # ---- the other BASELINE configs in the SAME driver-visible line: lego, zju (100 frames each, the same
    #      protocol) and the config-5 training step; `value` / `config` above are untouched ----
    if secondary:
        wl = {}
        for name in ("lego", "zju"):
            if time.perf_counter() - t_process > args.time_budget - 45:
                wl[name] = {"skipped": "time budget"}
                continue
            try:
                wl[name] = secondary_workload(name, dev, pmc=pmc_all)
            except Exception as e:
                wl[name] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        result["workloads"] = wl

    # ---- CPU baseline: the oracle (torch CPU restatement of the reference) on this box's host cores + parity ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import enerf_oracle as O
        from enerf_amd.synth import make_batch
        # torch CPU ops stop scaling (and regress) far below 256 threads: 8 threads rendered a DTU frame in 8.2 s in the
        # build container, 256 threads took 70 s on the GPU box.  Use at most 32 and say so.
        ncores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(ncores)
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        cb = {k: torch.from_numpy(v) for k, v in batch_np.items()}
        with torch.no_grad():
            small = {k: torch.from_numpy(v) for k, v in make_batch(64, 96, 3, cfg, seed=0).items()}
            O.forward(cfg, sd, small)                      # page-in / thread-pool warm-up (untimed)
            n_frames, t0 = 0, time.perf_counter()
            while n_frames < 1 or (time.perf_counter() - t0 < 10.0 and n_frames < 8):
                ref = O.forward(cfg, sd, cb)
                n_frames += 1
            cpu_s = (time.perf_counter() - t0) / n_frames
        result["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "frames/s", "cores": ncores, "kind": "port",
                                  "ref_over_port": REF_OVER_PORT["render_ft" if not cas.render_if[0] else "render_tt"],
                                  "sample": f"{n_frames} full {H}x{W} {S}-view frame(s) of this workload through "
                                            f"oracle/enerf_oracle.py (torch CPU, {ncores} of {os.cpu_count()} host threads), "
                                            "first small frame untimed"}
        if args.workload == "dtu":
            # BASELINE config 1 (configs[0]: the reference's own CPU-runnable case, render_if True,True, planes 48,8 — the
            # configuration BASELINE.md's 0.122 FPS / 8 cores is quoted on): one frame of the same port, same threads
            cfg1 = cfg.with_cas(render_if=(True, True))
            with torch.no_grad():
                t0 = time.perf_counter()
                O.forward(cfg1, sd, cb)
                c1 = time.perf_counter() - t0
            result["cpu_baseline_config1"] = {"value": 1.0 / c1, "unit": "frames/s", "cores": ncores, "kind": "port",
                                              "ref_over_port": REF_OVER_PORT["render_tt"],
                                              "sample": "1 frame, render_if True,True, volume_planes 48,8 (BASELINE configs[0])"}
        net.static_shapes = False
        o = net(batch)
        key = f"rgb_level{last}"
        err = float((o[key].cpu() - ref[key]).abs().max())
        result["parity_vs_oracle"] = {f"{key}_max_abs": err, "psnr_db": O.psnr(o[key].cpu(), ref[key])}

    if secondary and not args.no_cpu_baseline:
        left = args.time_budget - (time.perf_counter() - t_process)
        # the training child needs ~35 s (graph capture + verification, 20 replays, one CPU step of the oracle)
        result["workloads"]["train"] = train_child(left, live_pmc_ok=left > 58 and not args.no_live_pmc) if left > 40 else \
            {"skipped": f"time budget ({left:.0f} s left)"}
        # the reference's REAL loss (losses/enerf.py:30-38: + 0.01 x VGG16 perceptual at both levels, dtu_pretrain.yaml:41
        # train_img True,True) beside the MSE-only step: 20 replays, no extras
        left = args.time_budget - (time.perf_counter() - t_process)
        result["workloads"]["train_perceptual"] = train_child(left, perceptual=True) if left > 25 else \
            {"skipped": f"time budget ({left:.0f} s left)"}
        result["workloads"]["wall_s"] = round(time.perf_counter() - t_process, 1)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        emit_line(result)


if __name__ == "__main__":
    main()
