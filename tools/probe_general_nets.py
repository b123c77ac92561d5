"""Forward / backward of lnr_density_* in fp16 mode for the general network shapes of docs/HISTORY.md 4.6, at 4096 rays x 512 samples
(backward incl. the encoding's backward, features reused from the forward: the training loop's route)."""
import argparse, sys, json
sys.path.insert(0, '.')
import torch
from loner_amd import hip, ops

NETS = {
    "freq12 -> 128 ReLU x 2": (dict(otype="Frequency", n_frequencies=12), dict(activation="ReLU", n_neurons=128, n_hidden_layers=2)),
    "freq8 -> 64 Sine x 3 (SIREN)": (dict(otype="Frequency", n_frequencies=8), dict(activation="Sine", n_neurons=64, n_hidden_layers=3)),
    "hash16x2 -> 64 ReLU x 2": (dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=18, base_resolution=16),
                                dict(activation="ReLU", n_neurons=64, n_hidden_layers=2)),
    "freq6 -> 256 ReLU x 1": (dict(otype="Frequency", n_frequencies=6), dict(activation="ReLU", n_neurons=256, n_hidden_layers=1)),
    "freq10 -> 32 Tanh x 2": (dict(otype="Frequency", n_frequencies=10), dict(activation="Tanh", n_neurons=32, n_hidden_layers=2)),
}
ap = argparse.ArgumentParser()
ap.add_argument("--only", default="", help="substring of the network name")
ap.add_argument("--precision", default="", help="fp16 | fp32 (default: both)")
args = ap.parse_args()
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


for name, (enc, net) in NETS.items():
    if args.only not in name:
        continue
    for prec in ((args.precision,) if args.precision else ("fp16", "fp32")):
        spec = hip.make_net_spec(enc, dict(net, precision=prec))
        p = (torch.rand(int(spec.n_params), generator=g) - 0.5).cuda()
        grad = torch.zeros_like(p); dr = torch.zeros(N, 13, device="cuda")
        fwd = timed(lambda: ops.density_forward(spec, p, rays=rays, z=z))
        bwd = timed(lambda: ops.density_backward(spec, p, ds, grad, rays=rays, z=z, reuse_features=True, d_rays=dr))
        print(json.dumps({"network": name, "precision": prec, "samples": N * S, "forward_ms": round(fwd, 3), "backward_ms": round(bwd, 3)}))
