"""Development probe: one backward of the north-star network (freq12 -> 128 ReLU x 2, fp16 mode) at 2.1 M samples, for rocprofv3
--kernel-trace (per-dispatch durations of the three mlp_backward_f16_gen_kernel launches)."""
import sys, torch
sys.path.insert(0, '.')
from loner_amd import hip, ops
enc, net = dict(otype="Frequency", n_frequencies=12), dict(activation="ReLU", n_neurons=128, n_hidden_layers=2, precision="fp16")
if len(sys.argv) > 1 and sys.argv[1] == "siren":
    enc, net = dict(otype="Frequency", n_frequencies=8), dict(activation="Sine", n_neurons=64, n_hidden_layers=3, precision="fp16")
spec = hip.make_net_spec(enc, net)
N, S = 4096, 512
rays = torch.zeros(N, 13, device='cuda'); rays[:, 0:3] = torch.rand(N, 3, device='cuda') * 0.2 - 0.1
rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(N, 3, device='cuda'), dim=1); rays[:, 11] = 0.0117; rays[:, 12] = 0.58
z = torch.sort(torch.rand(N, S, device='cuda') * 0.57 + 0.0117, dim=1).values
ds = torch.randn(N, S, device='cuda'); dr = torch.zeros(N, 13, device='cuda')
p = torch.rand(int(spec.n_params), device='cuda') - 0.5
g = torch.zeros_like(p)
for _ in range(4):
    ops.density_forward(spec, p, rays=rays, z=z)
    ops.density_backward(spec, p, ds, g, rays=rays, z=z, d_rays=dr)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record(); ops.density_backward(spec, p, ds, g, rays=rays, z=z, d_rays=dr); b.record(); torch.cuda.synchronize()
print("backward ms", a.elapsed_time(b))
