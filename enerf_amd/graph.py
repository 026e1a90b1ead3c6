"""Whole-frame HIP graph: capture one ``Network.forward`` — ~38 kernel launches and their
ctypes/torch bookkeeping (0.26 ms of host time per 1.09 ms frame) — once, replay it per frame.

    frame = GraphedFrame(net, example_batch)      # captures on a side stream; static input/output buffers
    out = frame(batch)                            # copies the batch into the static inputs, one hipGraphLaunch

The graph holds the shapes and the weights' packed images of the capture; re-capture after load_state_dict /
a shape change.  Works because the HIP path never synchronises, allocates only through torch's (graph-aware)
caching allocator and takes every pointer from tensors that stay alive inside the graph's private pool."""
from __future__ import annotations

from typing import Dict

import torch


class GraphedFrame:
    def __init__(self, net, batch: Dict[str, torch.Tensor], warmup: int = 2):
        if net.training:
            raise RuntimeError("GraphedFrame: call net.eval() first")
        if getattr(net, "human", False) and not net.static_shapes:
            raise RuntimeError("GraphedFrame: the human variant needs static_shapes=True (no count readback inside a graph)")
        self.net = net
        self.static_in = {k: v.clone() if torch.is_tensor(v) else v for k, v in batch.items()}
        with torch.no_grad():
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(max(1, warmup)):          # packs weights, sizes workspaces, warms the allocator
                    net(self.static_in)
            torch.cuda.current_stream().wait_stream(s)
            self.graph = torch.cuda.CUDAGraph()
            # capture on the warm-up stream: the library's per-stream side lane (frame.hip) then exists before the capture
            # starts, and the lane's fork/join (event waits) is captured as a two-branch graph
            with torch.cuda.graph(self.graph, stream=s):
                self.static_out = net(self.static_in)
        # the graph holds raw addresses of the packed weight images and the FeatureNet scratch, which live OUTSIDE the
        # graph's private pool: keep them alive here and refuse to replay once the network has replaced them
        self._held = {k: v[0] for k, v in net._packed.items()}
        self._held_ws = {k: v["ws"] for k, v in net._frames.items()}     # the frame workspaces

    def __call__(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        for k, t in self._held.items():
            ent = self.net._packed.get(k)
            if ent is None or ent[0] is not t:
                raise RuntimeError("GraphedFrame: the network's weights changed (load_state_dict / .to()) after capture; "
                                   "re-capture the frame")
        for k, v in batch.items():
            if torch.is_tensor(v):
                dst = self.static_in[k]
                if dst.shape != v.shape:
                    raise RuntimeError(f"GraphedFrame: '{k}' changed shape {tuple(dst.shape)} -> {tuple(v.shape)}; re-capture")
                if dst.data_ptr() != v.data_ptr():
                    dst.copy_(v, non_blocking=True)
        self.graph.replay()
        return self.static_out                      # static buffers: valid until the next call
