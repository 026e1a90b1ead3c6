import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch_twins as T
from enerf_amd.autograd import nerf_mlp
from enerf_amd.network import NerfParams
from enerf_amd.lib import get_lib
dev = torch.device("cuda:0"); lib = get_lib()
g = torch.Generator().manual_seed(11); torch.manual_seed(11)
for F, S, P in ((11, 3, 37), (11, 4, 16), (35, 2, 21)):
    m = NerfParams(F, True).to(dev)
    vox = torch.randn(1, P, 8, generator=g).to(dev).requires_grad_(True)
    x = torch.randn(1, P, S, F + 4, generator=g).to(dev).requires_grad_(True)
    gout = torch.randn(1, P, 4, generator=g).to(dev)
    ref = T.nerf_forward(m, vox, x); ref.backward(gout)
    gx = x.grad.clone(); gv = vox.grad.clone(); x.grad = vox.grad = None
    want = {n: p.grad.clone() for n, p in m.named_parameters()}
    for p in m.parameters(): p.grad = None
    out = nerf_mlp(lib, m, T.nerf_forward, vox, x); out.backward(gout)
    d = (x.grad - gx).abs()[0]
    bad = (d > 1e-4).nonzero()
    print(F, S, P, "bad x entries", bad.shape[0], bad[:12].tolist(), "max", float(d.max()))
    for n, p in m.named_parameters():
        e = float((p.grad - want[n]).abs().max() / (want[n].abs().max() + 1e-12))
        if e > 1e-3: print("   param", n, e)
