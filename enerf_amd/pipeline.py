"""Frames in flight on several HIP streams (serving-throughput mode).

One frame is ~36 dependent launches, a third of them small cascade layers that leave most of the 256 CUs idle;
with two frames in flight on two streams the GPU fills those holes with the other frame's work.  Per-frame latency
goes up, frames/s goes up.  Same kernels, same results per frame.

    pipe = FramePipeline(net, depth=2)
    for batch in batches:
        out, done = pipe.submit(batch)        # enqueued on stream (i % depth); `done` is a torch.cuda.Event
        ...                                   # consume `out` after done.synchronize() / stream.wait_event(done)
    pipe.join()                               # current stream waits for everything submitted

Buffers: each frame allocates from its own stream's pool of torch's caching allocator; the network keeps one
FeatureNet workspace per stream.  A submitted batch must stay alive and unmodified until its event has fired."""
from __future__ import annotations

from typing import Dict, Tuple

import torch


class FramePipeline:
    def __init__(self, net, depth: int = 2):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if getattr(net, "overlap", False):
            raise RuntimeError("FramePipeline: use overlap=False (each frame already owns a stream)")
        self.net = net
        self.streams = [torch.cuda.Stream() for _ in range(depth)]
        self._i = 0

    def submit(self, batch: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], torch.cuda.Event]:
        s = self.streams[self._i % len(self.streams)]
        self._i += 1
        s.wait_stream(torch.cuda.current_stream())       # inputs produced on the caller's stream
        with torch.cuda.stream(s):
            out = self.net(batch)
            done = s.record_event()
        return out, done

    def join(self):
        cur = torch.cuda.current_stream()
        for s in self.streams:
            cur.wait_stream(s)
