"""Development probe: per-matrix error of the fp16-mode weight gradient against the oracle (which layer of a general network is off)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from loner_amd import hip, ops
from oracle import network as NW
from test_gpu_kernels import NETS
name = sys.argv[1] if len(sys.argv) > 1 else "freq_tanh2"
enc, net = NETS[name]
net16 = dict(net, precision="fp16")
spec_o, spec_h = NW.NetworkSpec.from_config(enc, net16), hip.make_net_spec(enc, net16)
params = NW.init_params(spec_o, 3)
gen = torch.Generator().manual_seed(8)
n = 1000
pts = torch.rand(n, 3, generator=gen) * 1.9 - 0.95
d_sigma = torch.randn(n, generator=gen)
grad = torch.zeros(int(spec_h.n_params), device="cuda")
d_pts = ops.density_backward(spec_h, params.cuda(), d_sigma.cuda(), grad, pts=pts.cuda(), want_d_pts=True)
p = params.clone().requires_grad_(True); x = pts.clone().requires_grad_(True)
(NW.density(spec_o, p, x) * d_sigma).sum().backward()
H, I, NHid = spec_o.n_neurons, spec_o.in_dim, spec_o.n_hidden
g, r = grad.cpu(), p.grad
off = 0
for nm, sz in [("W1", H * I)] + [(f"Wh{l}", H * H) for l in range(NHid - 1)] + [("Wo", 16 * H)]:
    a, b = g[off:off + sz], r[off:off + sz]
    print(f"{name} {nm}: max|ref| {float(b.abs().max()):.3e}  max|diff| {float((a - b).abs().max()):.3e}")
    if nm == "W1":
        d = (a - b).abs().reshape(H, I)
        print("   worst W1 columns:", torch.topk(d.max(0).values, 5))
    off += sz
print("dpts", float((d_pts.cpu() - x.grad).abs().max()), float(x.grad.abs().max()))
