"""Do torch reductions (semaphore memset + multi-block reduce) replay correctly inside a captured graph on this stack?"""
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
gy = torch.randn(3, 8, 512, 640, device=dev)
big = torch.randn(1 << 22, device=dev)
filler = torch.randn(64, 1 << 16, device=dev)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
def work():
    outs = []
    for k in range(6):
        t = gy * (1.0 + k)
        outs.append(t.sum((0, 2, 3)))
        outs.append((big * (k + 1.0)).sum())
        outs.append((filler * 2).sum(1))
    return outs
with torch.cuda.stream(side):
    ref = [o.clone() for o in work()]
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    outs = work()
bad = 0
for rep in range(50):
    g.replay(); torch.cuda.synchronize()
    for o, r in zip(outs, ref):
        if not torch.allclose(o, r, rtol=1e-4, atol=1e-2):
            bad += 1
    # eager noise between replays (allocations, other kernels)
    _ = (torch.randn(1 << 20, device=dev) * 2).sum().item()
print("bad comparisons over 50 replays:", bad, "of", 50 * len(ref))
