"""One training step as ONE hipGraph replay (trainer.py:56-63).

A dtu_pretrain step is ~700 kernel launches (FeatureNet, two cost-volume networks, two fused render levels, their backward
kernels, clip, Adam) whose GPU time (~35 ms) is no longer than the time Python + the dispatcher need to enqueue them
(~34 ms): eager steps are host-bound.  Shapes are static over an epoch (fixed image size, fixed ray count), so the step is
captured once — forward, loss, backward, gradient clip and optimizer update — and replayed with new batch contents copied
into the captured input buffers.

Nothing of the step stays outside the graph: the camera-only tables (train_path.camera_tables: 4x4 inverses) are device
kernels since ABI v7 (enerf_get_proj_mats, enerf_camera_tables) and are captured with it.

DATA-PARALLEL (trainer.py:15-22; ``distributed=True``): the same ONE graph per rank, with the collectives inside it.
DistributedDataParallel's reducer (autograd hooks, bucket views, a host-side bookkeeping pass per step) is an eager-step
device; what it computes is the mean over ranks of 436,012 floats.  ``FlatGradSync`` does exactly that as ONE all-reduce of
ONE flat 1.7 MB buffer enqueued after the backward pass — RCCL's kernel is a graph node like any other — and the
SyncBatchNorm statistics exchanges (autograd._BatchNormTrain, for the FeatureNet and the cost-volume networks alike: one
C-sized all-reduce per layer and direction, the global count kept on the device) are captured the same way.
(BatchNorm layer k+1 normalises what layer k produced from the GLOBAL statistics of layer k, so those ~46 exchanges are a
dependency chain: they cannot be merged into one buffer without changing SyncBatchNorm's arithmetic; captured, each costs
its ~10 us of xGMI latency and no host time.)  Every rank must construct the step at the same point (the constructor runs
collectives and agrees on the verification verdict collectively).

MEMSET NODES.  On this stack (ROCm 7.2 / PyTorch 2.10, MI355X) a hipMemsetAsync captured into a large graph does not replay
reliably: tools/micro/graph_reduce_check.py captures nothing but torch reductions (whose multi-block kernels reset their
semaphores with a memset) and gets wrong sums in about half of the replays.  Consequences here:
  * the library zeroes its accumulators with a fill kernel (csrc/capi.hip zero_async), never with hipMemsetAsync;
  * every large reduction of the step runs on the library's own kernels (bias gradients: enerf_channel_sums / the
    enerf_gemm_wgrad pass; BatchNorm: enerf_channel_sums);
  * a loss function should reduce with ``tree_sum`` / ``mse_loss`` below (single-block reductions: no semaphores);
  * ``GraphedTrainStep(verify=True)`` (the default) replays a few steps against eager steps from the same state and
    compares every gradient before the graph is trusted; a mismatch raises ``GraphMismatch`` (bench.py then runs eager).
"""
from typing import Callable, Dict, Iterable, Optional

import torch



class GraphMismatch(RuntimeError):
    pass


def tree_sum(x: torch.Tensor) -> torch.Tensor:
    """Sum of all elements as a tree of 256-wide row sums: every stage is a one-block-per-output reduction, i.e. none of
    torch's multi-block reduction kernels (semaphore memset) ends up in a captured graph.  Differentiable."""
    x = x.reshape(-1)
    while x.numel() > 256:
        pad = (-x.numel()) % 256
        if pad:
            x = torch.cat([x, x.new_zeros(pad)])
        x = x.view(-1, 256).sum(1)
    return x.sum()


def mse_loss(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """F.mse_loss(a, b) (mean reduction) on ``tree_sum``."""
    d = a - b
    return tree_sum(d * d) / d.numel()


class FlatGradSync:
    """DistributedDataParallel's gradient averaging (trainer.py:17-22) as ONE collective: the gradients of all parameters
    that have one are packed into one flat buffer, all-reduced once, scaled by 1/world and unpacked in place.  Parameters
    whose gradient is None (not on the loss's path — what ``find_unused_parameters=True`` tolerates) are skipped; with a
    static configuration that set is the same on every rank.  Works eagerly on any backend (gloo in the CPU tests) and
    inside a hipGraph capture on RCCL.  ``broadcast()`` is DDP's construction-time parameter/buffer broadcast from rank 0."""

    def __init__(self, module: torch.nn.Module, group=None):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("FlatGradSync needs an initialised torch.distributed process group")
        self.dist, self.group, self.module = dist, group, module
        self.world = dist.get_world_size(group)
        self.params = [p for p in module.parameters() if p.requires_grad]

    def broadcast(self):
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                self.dist.broadcast(t, 0, group=self.group)

    def __call__(self):
        grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        self.dist.all_reduce(flat, group=self.group)
        flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])


def train_step(net, optimizer, loss_fn, batch, clip_value: Optional[float] = 40.0, grad_sync: Optional[Callable] = None,
               zero: bool = True, params: Optional[Iterable] = None):
    """trainer.py:56-63: forward, loss, backward, (gradient averaging over ranks,) clip_grad_value_, optimizer step."""
    out = net(batch)
    loss = loss_fn(out, batch)
    if zero:
        optimizer.zero_grad(set_to_none=True)
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    if clip_value is not None:
        torch.nn.utils.clip_grad_value_(list(params) if params is not None else [p for p in net.parameters() if p.requires_grad], clip_value)
    optimizer.step()
    return loss


class GraphedTrainStep:
    """``step = GraphedTrainStep(net, optimizer, loss_fn, example_batch); loss = step(batch)``.

    ``loss_fn(outputs, batch) -> scalar``.  ``optimizer`` must be capture-safe (``torch.optim.Adam(..., capturable=True)``).
    Every batch passed later must have the tensors (keys, shapes, dtypes) of ``example_batch``.  The returned loss tensor is
    the graph's own output buffer: read it (``.item()``) before the next call.

    Construction is side-effect free on the model (also when it FAILS — GraphMismatch, a refused capture): the warm-up, the capture and the verification steps DO run optimizer
    steps on ``example_batch`` (optimizer state must exist before the capture), but parameters, buffers (BatchNorm running
    statistics, ``num_batches_tracked``) are snapshotted first and restored in place afterwards, and the optimizer's state
    tensors are zeroed in place (step 0, zero moments: exactly a fresh Adam/AdamW; the graph keeps their addresses).  An
    optimizer that already carried state before construction gets that state back instead."""

    def __init__(self, net, optimizer, loss_fn: Callable, example_batch: Dict[str, torch.Tensor], clip_value: float = 40.0,
                 warmup: int = 3, verify: bool = True, verify_steps: int = 4, distributed: bool = False, group=None,
                 fallback: str = "raise"):
        """``distributed=True``: data-parallel step (see DATA-PARALLEL above) — ``net`` is the plain network (SyncBatchNorm
        converted, NOT wrapped in DistributedDataParallel); parameters and buffers are broadcast from rank 0 first.

        ``fallback``: what happens when the stack refuses the capture or a replay fails the verification — on ANY rank; the
        verdict is one MAX all-reduce over the group, so all ranks take the same branch at the same point and nobody is left
        alone inside a collective.  ``"raise"`` (default): ``RuntimeError`` / ``GraphMismatch`` on every rank.  ``"eager"``:
        every rank keeps training with eager steps — the same ``train_step`` the graph would have captured, ``FlatGradSync``'s
        one flat all-reduce and the per-layer SyncBatchNorm exchanges enqueued per step — and ``step_launch`` says so.  (First
        contact with a multi-GPU node is a capture of >= 35 RCCL nodes that no 1-GPU lease can rehearse: a refused capture
        there must cost the graph, not the job.)"""
        if not net.training:
            raise ValueError("GraphedTrainStep captures a training step: call net.train() first")
        if warmup < 1:
            raise ValueError("at least one eager warm-up step: optimizer state must exist before the capture")
        if fallback not in ("raise", "eager"):
            raise ValueError("fallback: 'raise' or 'eager'")
        self.net, self.opt, self.loss_fn, self.clip, self.fallback = net, optimizer, loss_fn, clip_value, fallback
        self.graph, self.loss = None, None
        self.step_launch = "not constructed"
        self.sync = FlatGradSync(net, group) if distributed else None
        if self.sync is not None:
            self.sync.broadcast()
        self.static = {k: v.clone() for k, v in example_batch.items() if torch.is_tensor(v)}
        self.extra = {k: v for k, v in example_batch.items() if not torch.is_tensor(v)}
        # (the camera tables — 4x4 inverses — are device kernels of the library inside the step: captured with it)
        self._require_library()
        self._params = [p for p in net.parameters() if p.requires_grad]
        pristine_net = [(t, t.clone()) for t in net.state_dict().values()]
        pristine_opt = {id(t): t.clone() for st in optimizer.state.values() for t in st.values() if torch.is_tensor(t)}
        try:
            self._construct(warmup, verify, verify_steps)
        finally:
            # undo the training the construction did (warm-up + capture + verification steps on example_batch) — on the
            # failure paths too (GraphMismatch, a refused capture): a caller that falls back to eager steps (bench.py)
            # must start from the weights, BatchNorm buffers and optimizer state it handed in
            self._device_sync()
            with torch.no_grad():
                self._restore(pristine_net)
                for st in optimizer.state.values():
                    for t in st.values():
                        if torch.is_tensor(t):
                            t.copy_(pristine_opt[id(t)]) if id(t) in pristine_opt else t.zero_()
            net.invalidate_packed()

    # ---- device hooks: everything that needs a GPU, so that the collective protocol around them (capture -> one verdict for
    # all ranks -> verification -> one verdict -> graph or eager steps) runs unchanged on gloo in tests/test_world8_gloo.py ----
    def _require_library(self):
        from .train_path import _hip_lib
        if _hip_lib(self.net, next(iter(self.static.values()))) is None:
            raise RuntimeError("GraphedTrainStep: the HIP library is required (the training path has no eager fallback)")

    @staticmethod
    def _device_sync():
        torch.cuda.synchronize()

    def _warm_up(self, warmup: int):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on a side stream: lazy initialisation, caches, allocator
            for _ in range(warmup):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    def _capture(self):
        """Capture one step; returns the graph (``.replay()``).  Raises RuntimeError when the stack refuses the capture."""
        graph = torch.cuda.CUDAGraph()
        capture_kw = {}
        if self.sync is not None:
            # The process group's watchdog thread polls the events of the collectives the warm-up enqueued until it has seen
            # them complete; an event query from another thread while THIS thread captures in the default ("global") mode is
            # an illegal call that takes the process down (seen once in ~10 runs on ROCm 7 / PyTorch 2.10: abort inside the
            # capture).  The guard is the capture MODE: in "thread_local" mode only this thread's calls are policed, so a
            # watchdog query — whenever it comes — cannot abort the capture (kernels other threads, the autograd engine's,
            # launch into the capturing stream are captured all the same; the price: an illegal call from those threads is
            # not diagnosed).  In front of it the warm-up's collectives are drained DETERMINISTICALLY (round 4 slept 0.5 s
            # here): a fence collective is enqueued behind them on the communicator — collectives of one communicator
            # complete in order — and its Work handle is waited on and polled to completion on every rank, so every event
            # the watchdog still holds has fired before the capture starts; then all ranks meet once more.
            fence = self.sync.dist.all_reduce(torch.zeros(1, device=next(iter(self.static.values())).device),
                                              group=self.sync.group, async_op=True)
            fence.wait()
            torch.cuda.synchronize()
            while not fence.is_completed():
                pass
            self.sync.dist.barrier(group=self.sync.group)
            torch.cuda.synchronize()
            capture_kw["capture_error_mode"] = "thread_local"
        with torch.cuda.graph(graph, **capture_kw):
            self.loss = self._eager_step(zero=False)
        return graph

    def _agree(self, failed: bool) -> bool:
        """One verdict for all ranks (MAX over the group): nobody is left alone inside a collective."""
        if self.sync is None:
            return failed
        flag = torch.tensor([1.0 if failed else 0.0], device=next(iter(self.static.values())).device)
        self.sync.dist.all_reduce(flag, op=self.sync.dist.ReduceOp.MAX, group=self.sync.group)
        return bool(flag.item())

    def _construct(self, warmup: int, verify: bool, verify_steps: int):
        self._warm_up(warmup)
        self.opt.zero_grad(set_to_none=True)               # gradients are (re)allocated inside the graph's memory pool
        error = None
        try:
            self.graph = self._capture()
        except RuntimeError as e:                          # a refused capture executes nothing: every rank can still talk
            error = e
        if self._agree(error is not None):
            self.graph, self.loss = None, None
            why = f"capture failed: {str(error)[:200]}" if error is not None else "the capture failed on another rank"
            if self.fallback != "eager":
                raise error if error is not None else RuntimeError("GraphedTrainStep: " + why)
            self._fall_back(why)
            return
        self.net.invalidate_packed()
        self.step_launch = "one hipGraph replay per step" + ("" if verify else " (replays NOT verified against eager steps)")
        if verify:
            try:
                self._verify(verify_steps)                 # raises on EVERY rank or on none (the verdict inside is collective)
                self.step_launch += " (replays verified against eager steps)"
            except GraphMismatch as e:
                self.graph, self.loss = None, None
                if self.fallback != "eager":
                    raise
                self._fall_back(f"graph replay failed verification: {str(e)[:200]}")

    def _fall_back(self, why: str):
        self.opt.zero_grad(set_to_none=True)               # drop the graph pool's gradient buffers
        self.step_launch = f"eager steps on every rank ({why})" + \
            ("; gradient mean = one flat all-reduce per step, SyncBatchNorm exchanges per layer" if self.sync is not None else "")

    # ---- replay-vs-eager check (see MEMSET NODES above) ----
    def _snapshot(self):
        opt_state = [(t, t.clone()) for st in self.opt.state.values() for t in st.values() if torch.is_tensor(t)]
        return [(t, t.clone()) for t in self.net.state_dict().values()] + opt_state

    @staticmethod
    def _restore(snap):
        for t, saved in snap:
            t.copy_(saved)                                  # in place: the graph holds these addresses

    def _verify(self, steps: int):
        """From the same state: one replay vs one eager step, ``steps`` times; every parameter gradient must agree to
        2e-3 of its largest element (fp32 atomics reorder; a broken replay is off by orders of magnitude)."""
        for k in range(steps):
            snap = self._snapshot()
            self.graph.replay()
            g_graph = {n: p.grad.clone() for n, p in self.net.named_parameters() if p.grad is not None}
            loss_graph = float(self.loss.detach())
            self._restore(snap)
            grads_kept = {n: p.grad for n, p in self.net.named_parameters()}      # the graph's own gradient buffers
            loss_eager = float(self._eager_step().detach())           # zero_grad(set_to_none) detaches the graph's buffers: put them back
            bad = []
            for n, p in self.net.named_parameters():
                ge, gg = p.grad, g_graph.get(n)
                if (ge is None) != (gg is None):
                    bad.append((n, "missing"))
                elif ge is not None:
                    err, scale = float((ge - gg).abs().max()), float(ge.abs().max())
                    if not (err <= 2e-3 * scale + 1e-9):
                        bad.append((n, err, scale))
                p.grad = grads_kept[n]
            failed = bool(bad) or not abs(loss_graph - loss_eager) <= 1e-3 * abs(loss_eager) + 1e-7
            if self._agree(failed):
                raise GraphMismatch(f"replay {k}: loss {loss_graph} vs eager {loss_eager}; gradient mismatches "
                                    f"(name, max err, max |g|): {bad[:6]}" if failed else
                                    f"replay {k}: this rank's replay matched its eager step, another rank's did not")
        self.net.invalidate_packed()

    def _eager_step(self, zero: bool = True):
        batch = dict(self.static, **self.extra)
        return train_step(self.net, self.opt, self.loss_fn, batch, self.clip, self.sync, zero, self._params)

    def __call__(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        for k, dst in self.static.items():
            src = batch[k]
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        if self.graph is None:                             # fallback="eager" after a refused capture / failed verification
            return self._eager_step()
        self.graph.replay()
        self.net.invalidate_packed()                       # the inference weight images are stale after every update
        return self.loss


class GraphedTrainSteps:
    """One captured step per batch SHAPE.  The reference's DTU training draws the number of source views per sample
    (``dtu_pretrain.yaml:71-72``: S in {2, 3, 4} with probabilities .1/.8/.1), so a run sees a handful of distinct shape
    signatures, each static: ``GraphedTrainSteps(net, optimizer, loss_fn, [batch_S2, batch_S3, batch_S4])`` captures one
    hipGraph for each up front (same parameters, same optimizer state tensors; every graph owns its activation pool and
    gradient buffers) and ``step(batch)`` replays the one that matches.  Data-parallel: ranks may replay DIFFERENT graphs in
    the same step — the sequence and sizes of the collectives (C-sized SyncBatchNorm exchanges, the flat gradient buffer) do
    not depend on S — but every rank must construct the set from example batches in the same order."""

    def __init__(self, net, optimizer, loss_fn: Callable, example_batches, **kwargs):
        self.steps = {}
        for b in example_batches:
            k = self.key(b)
            if k not in self.steps:
                self.steps[k] = GraphedTrainStep(net, optimizer, loss_fn, b, **kwargs)

    @staticmethod
    def key(batch):
        return tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in batch.items() if torch.is_tensor(v)))

    def __call__(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        try:
            step = self.steps[self.key(batch)]
        except KeyError:
            raise KeyError("GraphedTrainSteps: no graph was captured for this batch's shapes "
                           f"({[(k, s) for k, s, _ in self.key(batch)][:4]} ...); pass an example of it to the constructor") from None
        return step(batch)
