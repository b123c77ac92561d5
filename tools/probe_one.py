"""time forward/backward of one network shape (experiment helper): python tools/probe_one.py <enc json> <net json>"""
import json, sys, torch
sys.path.insert(0, '.')
from loner_amd import hip, ops
enc, net = json.loads(sys.argv[1]), json.loads(sys.argv[2])
N, S = 4096, 512
rays = torch.zeros(N, 13, device='cuda'); rays[:, 0:3] = torch.rand(N, 3, device='cuda') * 0.2 - 0.1
rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(N, 3, device='cuda'), dim=1); rays[:, 11] = 0.0117; rays[:, 12] = 0.58
z = torch.sort(torch.rand(N, S, device='cuda') * 0.57 + 0.0117, dim=1).values
ds = torch.randn(N, S, device='cuda'); dr = torch.zeros(N, 13, device='cuda')
spec = hip.make_net_spec(enc, net)
p = torch.rand(int(spec.n_params), device='cuda') - 0.5
g = torch.zeros_like(p)
for _ in range(3):
    ops.density_forward(spec, p, rays=rays, z=z)
    ops.density_backward(spec, p, ds, g, rays=rays, z=z, d_rays=dr, reuse_features=True)
torch.cuda.synchronize()
