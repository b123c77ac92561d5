"""Host-side cost of the density calls (enqueue time with an idle GPU), to find launch-path overheads."""
import sys, time, torch
sys.path.insert(0, '.')
from loner_amd import hip, ops
spec = hip.make_net_spec(dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=18, base_resolution=16),
                         dict(n_neurons=64, n_hidden_layers=1))
N, S = 32, 64
rays = torch.zeros(N, 13, device='cuda'); rays[:, 3] = 1.0; rays[:, 11] = 0.01; rays[:, 12] = 0.5
z = torch.sort(torch.rand(N, S, device='cuda') * 0.4 + 0.01, dim=1).values
p = torch.rand(int(spec.n_params), device='cuda') - 0.5
g = torch.zeros_like(p); ds = torch.randn(N, S, device='cuda'); dr = torch.zeros(N, 13, device='cuda')
def timeit(fn, n=200):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
print("density_forward  host us:", round(timeit(lambda: ops.density_forward(spec, p, rays=rays, z=z)), 1))
print("density_backward host us:", round(timeit(lambda: (ops.density_forward(spec, p, rays=rays, z=z), ops.density_backward(spec, p, ds, g, rays=rays, z=z, reuse_features=True, d_rays=dr))), 1), "(incl. forward)")
print("workspace query  host us:", round(timeit(lambda: ops._workspace(spec, p.device, N * S)), 1))
print("torch.empty      host us:", round(timeit(lambda: torch.empty(N, S, device='cuda')), 1))
ops.profile_enable(True)
print("density_forward (profiling on) host us:", round(timeit(lambda: ops.density_forward(spec, p, rays=rays, z=z)), 1))
ops.profile_read(); ops.profile_enable(False)
