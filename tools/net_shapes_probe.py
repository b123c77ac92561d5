"""Forward / backward time of the density network for the supported shape classes at the default batch (2.1 M points)."""
import sys, torch
sys.path.insert(0, '.')
from loner_amd import hip, ops
SHAPES = {
    "default hash16x2 -> 64 ReLU x1": (dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=18, base_resolution=16), dict(activation="ReLU", n_neurons=64, n_hidden_layers=1)),
    "hash16x2 -> 64 ReLU x2": (dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=18, base_resolution=16), dict(activation="ReLU", n_neurons=64, n_hidden_layers=2)),
    "freq8 -> 64 Sine x3 (SIREN)": (dict(otype="Frequency", n_frequencies=8), dict(activation="Sine", n_neurons=64, n_hidden_layers=3)),
    "freq12 -> 128 ReLU x2": (dict(otype="Frequency", n_frequencies=12), dict(activation="ReLU", n_neurons=128, n_hidden_layers=2)),
    "freq6 -> 256 ReLU x1": (dict(otype="Frequency", n_frequencies=6), dict(activation="ReLU", n_neurons=256, n_hidden_layers=1)),
}
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
N, S = 4096, 512
rays = torch.zeros(N, 13, device='cuda'); rays[:, 0:3] = torch.rand(N, 3, device='cuda') * 0.2 - 0.1
rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(N, 3, device='cuda'), dim=1); rays[:, 11] = 0.0117; rays[:, 12] = 0.58
z = torch.sort(torch.rand(N, S, device='cuda') * 0.57 + 0.0117, dim=1).values
ds = torch.randn(N, S, device='cuda'); dr = torch.zeros(N, 13, device='cuda')
PRECS = sys.argv[1:] or ["fp32", "fp16"]
for name, (enc, net) in [(n + " [" + pr + "]", (e, dict(k, precision=pr))) for n, (e, k) in SHAPES.items() for pr in PRECS]:
    try:
        spec = hip.make_net_spec(enc, net)
        ops.density_forward(spec, torch.zeros(int(spec.n_params), device='cuda'), pts=torch.zeros(64, 3, device='cuda'))
    except RuntimeError as e:
        print(f"{name:44s} unsupported: {str(e)[-90:]}")
        continue
    p = torch.rand(int(spec.n_params), device='cuda') - 0.5
    g = torch.zeros_like(p)
    f = t(lambda: ops.density_forward(spec, p, rays=rays, z=z))
    b = t(lambda: ops.density_backward(spec, p, ds, g, rays=rays, z=z, d_rays=dr))
    mac = spec.n_neurons * spec.in_dim + (spec.n_hidden - 1) * spec.n_neurons ** 2
    print(f"{name:44s} fwd {f:7.3f} ms ({N*S*2*mac/f/1e9:6.1f} TFLOP/s)   bwd (incl. re-encode) {b:7.3f} ms ({N*S*6*mac/b/1e9:6.1f} TFLOP/s)")
