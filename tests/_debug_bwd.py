import sys, os, torch, numpy as np
sys.path.insert(0, '.')
from tests.test_gpu_kernels import _net, dv, NETS
from loner_amd import ops
from oracle import network as NW
name = sys.argv[1] if len(sys.argv) > 1 else "default"
spec_o, spec_h, params = _net(name, seed=2, table_gain=3000.0)
gen = torch.Generator().manual_seed(4)
n = 777
pts = (torch.rand(n, 3, generator=gen) * 1.9 - 0.95)
d_sigma = torch.randn(n, generator=gen)
grad = torch.zeros(int(spec_h.n_params), device="cuda")
ops.density_backward(spec_h, dv(params), dv(d_sigma), grad, pts=dv(pts), want_d_pts=False)
p32 = params.clone().requires_grad_(True)
(NW.density(spec_o, p32, pts) * d_sigma).sum().backward()
g = grad.cpu(); r = p32.grad
nm = int(spec_h.n_mlp_params)
print("mlp err", float((g[:nm]-r[:nm]).abs().max()), float(r[:nm].abs().max()))
F = int(spec_h.n_features)
for l in range(int(spec_h.n_levels)):
    lo = nm + int(spec_h.level_offset[l]) * F; hi = lo + int(spec_h.level_size[l]) * F
    d = (g[lo:hi]-r[lo:hi]).abs()
    top = torch.topk(d, 6).indices
    print("   worst idx", top.tolist(), "hip", g[lo:hi][top].tolist(), "ref", r[lo:hi][top].tolist())
    print(l, "size", hi-lo, "scale", float(spec_h.level_scale[l]), "err", float(d.max()), "ref max", float(r[lo:hi].abs().max()), "nz hip", int((g[lo:hi]!=0).sum()), "nz ref", int((r[lo:hi]!=0).sum()), "sum hip", float(g[lo:hi].sum()), "sum ref", float(r[lo:hi].sum()))
