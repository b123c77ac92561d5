"""Forward / backward time of the 256 x 2..3 networks (the layer-by-layer route, lnr_density_wide.hip) at the bench's sample count, both
precisions, beside the widest fused shapes (128 x 2, 256 x 1) for scale.      python tools/probe_wide_nets.py [--rays 4096 --samples 512]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loner_amd import hip, ops   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--samples", type=int, default=512)
ap.add_argument("--only", default="", help="substring of the network names to run (e.g. '256 x 2'); default: all")
ap.add_argument("--prec", default="fp32,fp16")
a = ap.parse_args()
g = torch.Generator().manual_seed(5)
n_rays, S = a.rays, a.samples
rays = torch.zeros(n_rays, 13); rays[:, 0:3] = torch.rand(n_rays, 3, generator=g) * 0.2 - 0.1
rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=1); rays[:, 11] = 0.0117; rays[:, 12] = 0.58
z = torch.sort(torch.rand(n_rays, S, generator=g) * 0.57 + 0.0117, dim=1).values
rays, z = rays.cuda(), z.cuda()
ds = torch.randn(n_rays, S, generator=g).cuda(); dr = torch.zeros(n_rays, 13, device="cuda")


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


freq12 = dict(otype="Frequency", n_frequencies=12)
for name, enc, net in (("freq12 -> 128 x 2", freq12, dict(activation="ReLU", n_neurons=128, n_hidden_layers=2)),
                       ("freq12 -> 256 x 1", freq12, dict(activation="ReLU", n_neurons=256, n_hidden_layers=1)),
                       ("freq12 -> 256 x 2", freq12, dict(activation="ReLU", n_neurons=256, n_hidden_layers=2)),
                       ("freq12 -> 256 x 3", freq12, dict(activation="ReLU", n_neurons=256, n_hidden_layers=3)),
                       ("hash16x2 -> 256 x 2", dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=18, base_resolution=16),
                        dict(activation="ReLU", n_neurons=256, n_hidden_layers=2))):
    if a.only and a.only not in name:
        continue
    for prec in a.prec.split(","):
        spec = hip.make_net_spec(enc, dict(net, precision=prec))
        p = (torch.rand(int(spec.n_params), generator=g) - 0.5).cuda(); grad = torch.zeros_like(p)
        fwd = timed(lambda: ops.density_forward(spec, p, rays=rays, z=z))
        bwd = timed(lambda: ops.density_backward(spec, p, ds, grad, rays=rays, z=z, reuse_features=True, d_rays=dr))
        mac = spec.n_neurons * spec.in_dim + (spec.n_hidden - 1) * spec.n_neurons ** 2 + spec.n_neurons
        pts = n_rays * S
        print(f"{name:22s} {prec}: forward {fwd:8.3f} ms ({pts * 2.0 * mac / fwd / 1e9:7.1f} TFLOP/s)   backward {bwd:8.3f} ms ({pts * 6.0 * mac / bwd / 1e9:7.1f} TFLOP/s)", flush=True)
