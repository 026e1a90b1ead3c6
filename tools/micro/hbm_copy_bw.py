import torch, time
x=torch.empty(64*1024*1024, dtype=torch.float32, device='cuda'); y=torch.empty_like(x)
for n in (4,16,64):
    a=x[:n*1024*1024]; b=y[:n*1024*1024]
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(f"copy {n*4} MB: {ms*1e3:.1f} us -> {2*n*4/ms/1e3:.2f} TB/s (read+write)")
    for _ in range(5): b.mul_(1.0001)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): b.zero_()
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/20
    print(f"fill {n*4} MB: {ms*1e3:.1f} us -> {n*4/ms/1e3:.2f} TB/s (write)")
