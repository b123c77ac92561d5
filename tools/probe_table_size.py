"""Development probe: does the hash table's size (L2 residency of a level's table) price the encode kernels?  Forward / backward of the
default network at 4096 x 512 samples with log2_hashmap_size 18 (default), 17, 16, 15."""
import sys, json
sys.path.insert(0, '.')
import torch
from loner_amd import hip, ops
N, S = 4096, 512
g = torch.Generator().manual_seed(5)
rays = torch.zeros(N, 13); rays[:, 0:3] = torch.rand(N, 3, generator=g) * 0.2 - 0.1
rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1); rays[:, 11] = 0.0117; rays[:, 12] = 0.58
z = torch.sort(torch.rand(N, S, generator=g) * 0.57 + 0.0117, dim=1).values
ds = torch.randn(N, S, generator=g)
rays, z, ds = rays.cuda(), z.cuda(), ds.cuda()


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for T in (18, 17, 16, 15):
    enc = dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=T, base_resolution=16)
    spec = hip.make_net_spec(enc, dict(activation="ReLU", n_neurons=64, n_hidden_layers=1))
    p = (torch.rand(int(spec.n_params), generator=g) - 0.5).cuda()
    grad = torch.zeros_like(p); dr = torch.zeros(N, 13, device="cuda")
    fwd = timed(lambda: ops.density_forward(spec, p, rays=rays, z=z))
    bwd = timed(lambda: ops.density_backward(spec, p, ds, grad, rays=rays, z=z, reuse_features=True, d_rays=dr))
    print(json.dumps({"log2_hashmap_size": T, "table_MB_per_fine_level": round(2 ** T * 8 / 2 ** 20, 2), "forward_ms": round(fwd, 3), "backward_ms": round(bwd, 3)}))
