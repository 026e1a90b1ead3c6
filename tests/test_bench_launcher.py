"""bench.py must be launchable the way the driver calls it: a plain ``python bench.py --gpus N`` (no torchrun, no
WORLD_SIZE) has to become N ranks by itself, and a line whose ``n_gpus`` differs from ``--gpus`` must never be printed
(VERDICT r02 weak #9).  Runs the real bench flow on the CPU lane emulator over gloo (``--emu``): the numbers mean nothing,
the launch / sharding / reduction path is the one the GPU run takes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(env_extra or {})
    env["OMP_NUM_THREADS"] = "1"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, cwd=ROOT, text=True,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def test_plain_gpus2_invocation_self_spawns_two_ranks():
    from emu_lib import emu_lib
    emu_lib()                                           # build the emulator twin once, outside the ranks
    p = _run(["--gpus", "2", "--emu", "--steps", "3", "--warmup", "1", "--workload", "zju"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout                    # rank 0 prints ONE line
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert len(d["per_rank_fps"]) == 2 and all(v > 0 for v in d["per_rank_fps"])
    # whole-job aggregate: 2 ranks x 3 frames over the slowest rank's time
    assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 1e-6
    assert d["config"]["distinct_batches"] == 4
    # every rank is pinned to its own cores (frame_parallel.pin_rank_to_cores) and the line says which
    aff = [r["affinity"] for r in d["rank_devices"]]
    assert len(aff) == 2 and all(a and a["n"] >= 1 and a["cores"] for a in aff), aff
    assert aff[0]["cores"] != aff[1]["cores"]


def test_single_rank_emu_line_and_refusal_of_mismatched_world():
    p = _run(["--gpus", "1", "--emu", "--steps", "2", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = _json_lines(p.stdout)[0]
    assert d["n_gpus"] == 1 and len(d["per_rank_fps"]) == 1
    # a launcher that gives one rank while --gpus says two must not produce a line
    p = _run(["--gpus", "2", "--emu", "--steps", "2"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not _json_lines(p.stdout)
    assert "refusing" in p.stderr


def test_train_gpus2_dry_run_takes_the_flat_data_parallel_step():
    """``bench.py --train --gpus 2``: two self-spawned ranks run the step the hipGraph captures on a
    GPU — train_graph.train_step with ONE flat gradient all-reduce and the cost-volume networks' SyncBatchNorm exchange —
    eagerly on the emulator over gloo.  Both ranks must finish and agree on one line with n_gpus 2."""
    from emu_lib import emu_lib
    emu_lib()
    p = _run(["--train", "--gpus", "2", "--emu", "--steps", "1", "--warmup", "0"], timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak" and d["unit"] == "samples/s"
    assert "one flat gradient all-reduce per step" in d["config"]["parallelism"]
    assert "DRY RUN" in d["config"]["step_launch"]
    assert d["final_loss"] == d["final_loss"] and 0 < d["final_loss"] < 10


def test_pmc_rows_are_cut_at_the_sentinel_launches():
    """bench.py's live PMC child renders dtu, lego and zju in ONE process per counter pass, bracketing the measured frames of each
    workload with sentinel launches (`--pmc-sequence`); `pmc_accumulate` must attribute the dispatch-ordered counter rows to the
    right workload — both lego and zju launch k_render_rays<3, 4, ...> and every workload launches k_smooth0_cb — and drop
    everything outside the brackets (network construction, warm-up frames)."""
    sys.argv = sys.argv[:1]
    import bench
    rows, did = [], [0]

    def disp(name, val):
        did[0] += 1
        rows.append({"Dispatch_Id": str(did[0]), "Kernel_Name": name, "Counter_Name": "FETCH_SIZE", "Counter_Value": str(val)})
    S = f"enerf::{bench.PMC_SENTINEL}(float const*, int, int, int, unsigned char*)"
    for w, (render, base) in enumerate((("void enerf::k_render_rays<3, 3, 12, 3, false, true, 0>(x)", 100.0),
                                        ("void enerf::k_render_rays<3, 4, 12, 3, false, true, 0>(x)", 200.0),
                                        ("void enerf::k_render_rays<3, 4, 12, 3, false, true, 0>(x)", 300.0))):
        disp("enerf::k_conv2d_pack(x)", 1.0)                         # construction: outside the brackets
        for _ in range(2):                                           # warm-up frames: outside
            disp("enerf::k_smooth0_cb(x)", 9999.0); disp(render, 9999.0)
        disp(S, 0.0)
        for f in range(3):
            disp("enerf::k_smooth0_cb(x)", base + f); disp(render, 10 * base + f)
        disp(S, 0.0)
    rows.reverse()                                                   # the parser sorts by dispatch id itself
    acc = bench.pmc_accumulate(rows, ["dtu", "lego", "zju"], None, {})
    assert sorted({w for w, _ in acc}) == ["dtu", "lego", "zju"]
    for w, base in (("dtu", 100.0), ("lego", 200.0), ("zju", 300.0)):
        assert acc[(w, "k_smooth0_cb(x)")]["FETCH_SIZE"] == [base, base + 1, base + 2]
        (rk,) = [k for (ww, k) in acc if ww == w and k.startswith("k_render_rays")]
        assert acc[(w, rk)]["FETCH_SIZE"] == [10 * base, 10 * base + 1, 10 * base + 2]
    assert not any(k.startswith("k_conv2d_pack") for _, k in acc)
    # the training child: no sentinels, one kernel prefix, workload ""
    acc2 = bench.pmc_accumulate([{"Dispatch_Id": "2", "Kernel_Name": "void enerf::k_mlp_bwd<3, 3>(a)", "Counter_Name": "WRITE_SIZE", "Counter_Value": "5"},
                                 {"Dispatch_Id": "1", "Kernel_Name": "void enerf::k_mlp_fwd<3, 3>(a)", "Counter_Name": "WRITE_SIZE", "Counter_Value": "7"}],
                                None, "k_mlp_bwd<3, 3>", {})
    assert acc2 == {("", "k_mlp_bwd<3, 3>(a)"): {"WRITE_SIZE": [5.0]}}
    assert bench.pmc_lookup({"lego": {"k_render_rays<3, 4, 12, 3, false, true, 0>(x)": {"hbm_bytes_per_launch": 1.0}}}, "lego", "k_render_rays<3, 4")
    assert bench.pmc_lookup(None, "lego", "k") is None and bench.pmc_lookup({"dtu": {}}, "lego", "k") is None
