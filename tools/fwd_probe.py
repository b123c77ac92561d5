import sys, torch
sys.path.insert(0, '.')
from loner_amd import hip, ops
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
N, S = 4096, 512
rays = torch.zeros(N, 13, device='cuda'); rays[:, 0:3] = torch.rand(N, 3, device='cuda') * 0.2 - 0.1
d = torch.nn.functional.normalize(torch.randn(N, 3, device='cuda'), dim=1); rays[:, 3:6] = d; rays[:, 11] = 0.0117; rays[:, 12] = 0.58
z = torch.sort(torch.rand(N, S, device='cuda') * 0.57 + 0.0117, dim=1).values
for log2 in (18,):
    spec = hip.make_net_spec(dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=log2, base_resolution=16),
                             dict(n_neurons=64, n_hidden_layers=1))
    p = (torch.rand(int(spec.n_params), device='cuda') - 0.5)
    ds = torch.randn(N, S, device='cuda'); g = torch.zeros_like(p)
    f = t(lambda: ops.density_forward(spec, p, rays=rays, z=z))
    b0 = t(lambda: ops.density_backward(spec, p, ds, g, rays=rays, z=z, want_d_pts=False), 5)
    b1 = t(lambda: ops.density_backward(spec, p, ds, g, rays=rays, z=z, want_d_pts=True), 5)
    print(f"log2_T={log2}: table {int(spec.n_params)*4/1e6:6.1f} MB  fwd {f:.3f} ms  bwd(no dx) {b0:.3f} ms  bwd(dx) {b1:.3f} ms")
