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

from typing import Dict, Optional, Tuple

import torch

from .lib import Options, throughput_options

# Kernel-variant choices that differ between "one frame as fast as possible" and "most frames per second": with
# several frames in flight the matrix pipes are shared, so variants that issue fewer MFMAs win even where they are
# slower in isolation (measured on MI355X, 6 frames in flight: 1216 -> 1234 -> 1244 frames/s; one frame at a time the
# same switches cost 2 %).  They are an explicit ``enerf_options_t`` (lib.throughput_options(): tap-packed conv3d for
# every Cout=8 layer) handed to every launch of the frames submitted HERE — nothing
# process-global is touched, other users of the same Network keep its own ``net.options``.  Results change at the
# 1e-6 level (different summation order), not bit for bit.


class FramePipeline:
    def __init__(self, net, depth: int = 2, throughput_tuning: bool = False, options: Optional[Options] = None):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.net = net
        self.options = options if options is not None else (throughput_options() if throughput_tuning else net.options)
        # weight images are packed lazily by device kernels: do it now on the caller's stream, then let every pipeline
        # stream start after it (a cold pipeline would otherwise pack on streams[0] while streams[1] reads)
        net.prepare()
        self.streams = [torch.cuda.Stream() for _ in range(depth)]
        cur = torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(cur)
        self._i = 0

    def close(self):
        """Wait for everything submitted."""
        self.join()

    def submit(self, batch: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], torch.cuda.Event]:
        """Outputs are allocated from the pipeline stream's pool; they are marked as used by the caller's current
        stream (``record_stream``), so consuming them there after ``wait_event(done)`` and dropping them is safe.
        A consumer on yet another stream must call ``record_stream`` itself."""
        cur = torch.cuda.current_stream()
        s = self.streams[self._i % len(self.streams)]
        self._i += 1
        s.wait_stream(cur)                               # inputs produced on the caller's stream
        with torch.cuda.stream(s):
            out = self.net._forward(batch, self.options)
            done = s.record_event()
        for v in out.values():
            v.record_stream(cur)
        return out, done

    def join(self):
        cur = torch.cuda.current_stream()
        for s in self.streams:
            cur.wait_stream(s)
