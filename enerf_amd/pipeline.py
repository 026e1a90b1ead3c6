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

import os
from typing import Dict, Tuple

import torch

# Kernel-variant choices that differ between "one frame as fast as possible" and "most frames per second": with
# several frames in flight the matrix pipes are shared, so variants that issue fewer MFMAs win even where they are
# slower in isolation (measured on MI355X, 6 frames in flight: 1216 -> 1234 -> 1244 frames/s; one frame at a time the
# same switches cost 2 %).  The library reads these per launch; they are set only while a pipeline is open and only
# if the user has not set them.  Results change at the 1e-6 level (different summation order), not bit for bit.
THROUGHPUT_KNOBS = {"ENERF_CONV_PK8": "2",       # tap-packed conv3d for every Cout=8(+1) layer, not just where it is faster alone
                    "ENERF_RENDER_OCC": "2"}     # render kernel at 2 blocks/CU (no spills) instead of 3


class FramePipeline:
    def __init__(self, net, depth: int = 2, throughput_tuning: bool = False):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if getattr(net, "overlap", False):
            raise RuntimeError("FramePipeline: use overlap=False (each frame already owns a stream)")
        self.net = net
        self.streams = [torch.cuda.Stream() for _ in range(depth)]
        self._i = 0
        self._restore = {}
        if throughput_tuning:
            for k, v in THROUGHPUT_KNOBS.items():
                if k not in os.environ:
                    self._restore[k] = None
                    os.environ[k] = v

    def close(self):
        """Wait for everything submitted and undo the throughput knobs."""
        self.join()
        for k in self._restore:
            os.environ.pop(k, None)
        self._restore = {}

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
