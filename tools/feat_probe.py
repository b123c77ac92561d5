import sys, torch, numpy as np
sys.path.insert(0, '.')
from loner_amd import hip, ops
from oracle import network as NW
enc = dict(otype="Frequency", n_frequencies=12); net = dict(activation="None", n_neurons=16, n_hidden_layers=1)
spec_h = hip.make_net_spec(enc, net); spec_o = NW.NetworkSpec.from_config(enc, net)
gen = torch.Generator().manual_seed(1); pts = torch.rand(4000, 3, generator=gen) * 1.98 - 0.99
feat = NW.encode_frequency(spec_o, (pts + 1) / 2)
feat64 = NW.encode_frequency(spec_o, (pts.double() + 1) / 2)
worst = []
for k in [0, 1, 20, 21, 22, 23, 46, 47, 70, 71]:
    p = torch.zeros(int(spec_h.n_params)); p[0 * spec_h.in_dim + k] = 1.0; p[16 * spec_h.in_dim + 0] = 1.0
    s = ops.density_forward(spec_h, p.cuda(), pts=pts.cuda()).cpu()
    print(k, 'hip-vs-torch32', float((s - feat[:, k]).abs().max()), 'hip-vs-64', float((s.double() - feat64[:, k]).abs().max()),
          'torch32-vs-64', float((feat[:, k].double() - feat64[:, k]).abs().max()))
