"""World-size-8 runs on gloo: what an 8-GPU node will execute, checked without one.

  * the frame partition ``f mod G`` and the aggregate over an ODD frame count (19 frames on 8 ranks: ranks 0-2 render three);
  * ``FlatGradSync``: one flat all-reduce scaled by 1/world == the mean over 8 ranks, parameters without a gradient skipped;
  * SyncBatchNorm on the HIP training kernels (autograd._BatchNormTrain on the lane emulator): statistics over 8 ranks with
    DIFFERENT position counts per rank == one process normalising the concatenation, also on a 4-rank SUBGROUP
    (``process_group``: ADVICE r03) while the other four ranks form their own;
  * ``GraphedTrainStep(distributed=True, fallback=...)``: the agreement protocol with a capture refused on ONE rank and with a
    replay that fails the verification on ONE rank — every rank lands in the same branch (eager steps, or the same exception),
    the model is handed back untouched, and the eager steps that follow keep the ranks' parameters identical;
  * ``rank_bindings``: every rank listed once, communicator size counted by a collective, duplicates refused on all ranks;
  * ``bench.py --gpus 8`` and ``bench.py --train --gpus 8`` self-spawn eight ranks (CPU lane emulator, tiny frame).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
WORLD = 8


def _counts():
    return [40 + 7 * r for r in range(WORLD)]           # positions per rank: all different


def _bn_data(relu):
    g = torch.Generator().manual_seed(5)
    zs = [torch.randn(n, 8, generator=g) * 2 + 1 for n in _counts()]
    gs = [torch.randn(n, 8, generator=g) for n in _counts()]
    return zs, gs


class _ToyNet(torch.nn.Module):
    """Stands in for the network in the protocol test: batch dict in, output dict out, `invalidate_packed` like Network."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.body = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))

    def forward(self, batch):
        return {"y": self.body(batch["x"])}

    def invalidate_packed(self):
        pass


def _cpu_step_class(rank, inject):
    """GraphedTrainStep with its GPU hooks replaced (no library, no stream, a 'graph' whose replay runs the step eagerly): the
    collective protocol around the hooks is the product's own code.  inject = (kind, rank) makes ONE rank fail."""
    from enerf_amd.train_graph import GraphedTrainStep

    class CpuStep(GraphedTrainStep):
        def _require_library(self):
            pass

        @staticmethod
        def _device_sync():
            pass

        def _warm_up(self, warmup):
            for _ in range(warmup):
                self._eager_step()

        def _capture(self):
            if inject == ("capture", rank):
                raise RuntimeError("injected: operation not permitted when stream is capturing")
            step = self

            class Graph:
                def replay(self):
                    step.loss = step._eager_step()
                    if inject == ("verify", rank):       # a replay that computes something else on this rank only
                        with torch.no_grad():
                            next(p for p in step.net.parameters() if p.grad is not None).grad.mul_(3.0)
            return Graph()                               # like a real capture: nothing executes (no collective runs) here
    return CpuStep


def _fallback_scenarios(rank):
    from enerf_amd.train_graph import GraphMismatch
    out = {}
    for name, inject, fallback in (("clean", None, "eager"), ("capture_r5", ("capture", 5), "eager"), ("verify_r2", ("verify", 2), "eager"),
                                   ("capture_r5_raise", ("capture", 5), "raise"), ("verify_r2_raise", ("verify", 2), "raise")):
        net = _ToyNet().train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-2)
        g = torch.Generator().manual_seed(200 + rank)
        batch = {"x": torch.randn(6, 5, generator=g), "t": torch.randn(6, 3, generator=g)}
        loss_fn = lambda o, b: (o["y"] - b["t"]).square().mean()
        before = [p.detach().clone() for p in net.parameters()]
        try:
            step = _cpu_step_class(rank, inject)(net, opt, loss_fn, batch, distributed=True, fallback=fallback, warmup=1, verify_steps=2)
        except (GraphMismatch, RuntimeError) as e:
            out[name] = ("raised", type(e).__name__, all(torch.equal(a, p) for a, p in zip(before, net.parameters())))
            continue
        untouched = all(torch.equal(a, p) for a, p in zip(before, net.parameters()))
        for _ in range(3):
            loss = step(batch)
        out[name] = (step.step_launch, step.graph is None, untouched, [p.detach().numpy().copy() for p in net.parameters()], float(loss.detach()))
    return out


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, HERE)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ENERF_EMU_THREADS="1")
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from enerf_amd.frame_parallel import rank_bindings, render_sharded
        from enerf_amd.train_graph import FlatGradSync
        res = {}
        # ---- frame partition, odd frame count
        outs, fps, tmax = render_sharded(lambda f: f * f, 19, rank, world)
        res["frames"] = (sorted(outs), fps, tmax)
        # ---- bindings
        b = rank_bindings(rank, world, rank, torch.device("cpu"))
        res["bind"] = (b["ranks_seen"], [x["rank"] for x in b["bindings"]], b["backend"])
        # ---- flat gradient sync: mean over 8 ranks; `unused` has no gradient on any rank
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
        m.unused = torch.nn.Parameter(torch.ones(4))
        if rank == 3:
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(1.0)                          # broadcast() must overwrite this
        sync = FlatGradSync(m)
        sync.broadcast()
        x = torch.randn(6, 5, generator=torch.Generator().manual_seed(100 + rank))
        m(x).square().sum().backward()
        local = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        sync()
        res["flat"] = ({n: p.grad.numpy().copy() for n, p in m.named_parameters() if p.grad is not None},
                       {n: v.numpy() for n, v in local.items()}, m.unused.grad is None,
                       {n: p.detach().numpy().copy() for n, p in m.named_parameters()})
        # ---- SyncBatchNorm statistics on the HIP training kernels (lane emulator), whole world then 4-rank subgroups
        from emu_lib import emu_lib
        from enerf_amd.autograd import _BatchNormTrain
        lib = emu_lib()
        groups = [dist.new_group(list(range(0, 4))), dist.new_group(list(range(4, 8)))]
        zs, gs = _bn_data(True)
        for tag, grp in (("world", None), ("sub", groups[rank // 4])):
            bn = torch.nn.SyncBatchNorm(8, process_group=grp).train()
            with torch.no_grad():
                bn.weight.copy_(torch.linspace(0.5, 1.5, 8)); bn.bias.copy_(torch.linspace(-1, 1, 8))
            blk = _BatchNormTrain(lib, bn, True)
            out = blk.forward(zs[rank].clone())
            dz, dgamma, dbeta = blk.backward(gs[rank].clone())
            res[tag] = tuple(t.detach().numpy().copy() for t in (out, dz, dgamma, dbeta, bn.running_mean, bn.running_var))
        res["fallback"] = _fallback_scenarios(rank)
        q.put((rank, res))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, "ERROR: " + traceback.format_exc()))


@pytest.fixture(scope="module")
def world8():
    from emu_lib import emu_lib
    emu_lib()                                            # build the emulator twin once, outside the ranks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    [p.join(120) for p in procs]
    for r in res:
        assert not isinstance(r[1], str), r[1]
    return [r[1] for r in res]


def test_odd_frame_count_is_partitioned_f_mod_g(world8):
    seen = []
    for rank, r in enumerate(world8):
        frames, fps, tmax = r["frames"]
        assert frames == list(range(rank, 19, WORLD))
        assert fps == pytest.approx(19 / tmax)           # whole-job aggregate: ALL frames over the slowest rank's time
        seen += frames
    assert sorted(seen) == list(range(19))
    assert len({r["frames"][1] for r in world8}) == 1    # every rank holds the same aggregate


def test_rank_bindings_count_the_communicator(world8):
    for r in world8:
        assert r["bind"] == (WORLD, list(range(WORLD)), "gloo")


def test_flat_gradient_sync_is_the_mean_over_eight_ranks(world8):
    mean = {n: sum(r["flat"][1][n] for r in world8) / WORLD for n in world8[0]["flat"][1]}
    for r in world8:
        synced, _, unused_none, params = r["flat"]
        assert unused_none
        for n, v in synced.items():
            assert np.abs(v - mean[n]).max() <= 1e-6 * max(np.abs(mean[n]).max(), 1e-12) + 1e-9, n
        for n, v in params.items():                      # rank 3's edited weights were replaced by rank 0's
            assert np.array_equal(v, world8[0]["flat"][3][n]), n


def test_graphed_step_falls_back_on_every_rank_together(world8):
    """One rank refuses the capture (or fails the replay verification): with fallback="eager" EVERY rank ends up stepping
    eagerly from the untouched model and the ranks stay in lock step; with fallback="raise" every rank raises."""
    for r in world8:
        f = r["fallback"]
        launch, eager, untouched, _, _ = f["clean"]
        assert launch.startswith("one hipGraph replay per step") and "verified" in launch and not eager and untouched
        for name, why in (("capture_r5", "capture failed"), ("verify_r2", "graph replay failed verification")):
            launch, eager, untouched, _, _ = f[name]
            assert eager and untouched and launch.startswith("eager steps on every rank") and "one flat all-reduce" in launch, (name, launch)
        # the rank that failed names its error, the others say another rank failed — same branch either way
        assert f["capture_r5_raise"][0] == "raised" and f["capture_r5_raise"][1] == "RuntimeError" and f["capture_r5_raise"][2]
        assert f["verify_r2_raise"][:2] == ("raised", "GraphMismatch") and f["verify_r2_raise"][2]
    assert "injected" in world8[5]["fallback"]["capture_r5"][0] and "another rank" in world8[0]["fallback"]["capture_r5"][0]
    for name in ("clean", "capture_r5", "verify_r2"):      # three steps later: identical parameters on all ranks, and the three
        ref = world8[0]["fallback"][name][3]               # ways of stepping (graph double / eager fallback) agree
        for r in world8:
            for a, b in zip(r["fallback"][name][3], ref):
                assert np.array_equal(a, b), name
        for a, b in zip(ref, world8[0]["fallback"]["clean"][3]):
            assert np.abs(a - b).max() < 1e-6, name


def _bn_reference(ranks):
    zs, gs = _bn_data(True)
    z = torch.cat([zs[r] for r in ranks]).requires_grad_(True)
    ref = torch.nn.BatchNorm1d(8).train()
    with torch.no_grad():
        ref.weight.copy_(torch.linspace(0.5, 1.5, 8)); ref.bias.copy_(torch.linspace(-1, 1, 8))
    out = torch.relu(ref(z))
    out.backward(torch.cat([gs[r] for r in ranks]))
    return out.detach(), z.grad, ref


@pytest.mark.parametrize("tag,groups", [("world", [list(range(8))]), ("sub", [[0, 1, 2, 3], [4, 5, 6, 7]])])
def test_syncbn_statistics_on_hip_kernels_equal_one_process(world8, tag, groups):
    """Global mean/var from per-rank sums with DIFFERENT counts (the count travels with the sums and stays on the device);
    d gamma / d beta stay rank-local like torch's SyncBatchNorm (their mean over ranks is DDP's job)."""
    for ranks in groups:
        out, dz, ref = _bn_reference(ranks)
        offs = np.cumsum([0] + [_counts()[r] for r in ranks])
        dgamma = sum(world8[r][tag][2] for r in ranks)
        dbeta = sum(world8[r][tag][3] for r in ranks)
        assert np.abs(dgamma - ref.weight.grad.numpy()).max() <= 2e-5 * float(ref.weight.grad.abs().max())
        assert np.abs(dbeta - ref.bias.grad.numpy()).max() <= 2e-5 * float(ref.bias.grad.abs().max())
        for i, r in enumerate(ranks):
            o, g, _, _, rm, rv = world8[r][tag]
            sl = slice(offs[i], offs[i + 1])
            assert np.abs(o - out[sl].numpy()).max() <= 2e-6 * float(out.abs().max())
            assert np.abs(g - dz[sl].numpy()).max() <= 2e-5 * float(dz.abs().max())
            assert np.abs(rm - ref.running_mean.numpy()).max() <= 1e-6 + 1e-5 * float(ref.running_mean.abs().max())
            assert np.abs(rv - ref.running_var.numpy()).max() <= 1e-6 + 1e-5 * float(ref.running_var.abs().max())


def test_duplicate_device_bindings_are_refused_on_every_rank():
    """Two ranks reporting the same device: rank_bindings raises on ALL ranks after the exchange (nobody hangs)."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from enerf_amd import frame_parallel as fp
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
real = fp.torch.cuda.get_device_properties
class P: name = "fake"; uuid = None; pci_bus_id = None
fp.torch.cuda.get_device_properties = lambda d: P()
try:
    fp.rank_bindings(rank, world, 0, torch.device("cuda", 0))       # both ranks claim cuda:0 of the same visible list
    sys.stdout.write("NOT-REFUSED\n")
except RuntimeError as e:
    sys.stdout.write("refused-" + str("share a device" in str(e)) + "\n")     # one write per rank: the two ranks share the pipe
sys.stdout.flush()
dist.destroy_process_group()
''' % ROOT
    env = dict(os.environ, OMP_NUM_THREADS="1")
    # torch.distributed.run has no -c: write the snippet to a temp file
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(code)
    try:
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                            "127.0.0.1", "--master-port", str(32000 + os.getpid() % 2000), f.name], env=env, text=True,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    finally:
        os.unlink(f.name)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.count("refused-True") == 2 and "NOT-REFUSED" not in p.stdout, p.stdout


def _bench(args, timeout):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, cwd=ROOT, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                     # rank 0 prints ONE line
    return lines[0], p.stderr


def test_bench_gpus8_self_spawns_eight_ranks_and_reports_them():
    from emu_lib import emu_lib
    emu_lib()
    d, err = _bench(["--gpus", "8", "--emu", "--steps", "1", "--warmup", "0", "--batches", "1"], 900)
    assert d["n_gpus"] == 8 and d["collective_ranks_seen"] == 8 and d["collective_backend"] == "gloo"
    assert [b["rank"] for b in d["rank_devices"]] == list(range(8)) and len(d["per_rank_fps"]) == 8
    assert abs(d["value"] - 8 * 1 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6      # whole-job aggregate
    assert sum(f"[bench] rank {r}/8" in err for r in range(8)) == 8                     # every rank prints its binding


def test_bench_train_gpus8_dry_run():
    from emu_lib import emu_lib
    emu_lib()
    d, _ = _bench(["--train", "--gpus", "8", "--emu", "--steps", "1", "--warmup", "0"], 1500)
    assert d["n_gpus"] == 8 and d["collective_ranks_seen"] == 8 and d["unit"] == "samples/s" and d["scaling"] == "weak"
    assert "data-parallel x8" in d["config"]["parallelism"] and "one flat gradient all-reduce per step" in d["config"]["parallelism"]
    assert d["final_loss"] == d["final_loss"] and 0 < d["final_loss"] < 10
