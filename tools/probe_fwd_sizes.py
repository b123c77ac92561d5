"""Development probe: forward of the north-star network class (freq12 -> 128 ReLU x 2, fp16 mode) at several sample counts, for
rocprofv3 --kernel-trace: fixed cost per launch (weight staging) against the per-tile cost.  tools/fwd_sizes_report.py reads the trace."""
import sys, torch
sys.path.insert(0, '.')
from loner_amd import hip, ops
spec = hip.make_net_spec(dict(otype="Frequency", n_frequencies=12), dict(activation="ReLU", n_neurons=128, n_hidden_layers=2, precision="fp16"))
p = torch.rand(int(spec.n_params), device='cuda') - 0.5
for N in (64, 512, 2048, 4096, 8192):
    S = 512
    rays = torch.zeros(N, 13, device='cuda'); rays[:, 0:3] = torch.rand(N, 3, device='cuda') * 0.2 - 0.1
    rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(N, 3, device='cuda'), dim=1); rays[:, 11] = 0.0117; rays[:, 12] = 0.58
    z = torch.sort(torch.rand(N, S, device='cuda') * 0.57 + 0.0117, dim=1).values
    for _ in range(6):
        ops.density_forward(spec, p, rays=rays, z=z)
    torch.cuda.synchronize()
