"""Development probe: lnr_render_forward at 2048 samples per ray (depth only): the staged-row path (no noise) against the direct loads
(a zero noise tensor forces them), and the same at 512 samples."""
import sys
sys.path.insert(0, '.')
import torch
from loner_amd import ops
g = torch.Generator().manual_seed(0)
for n, S in ((8192, 2048), (8192, 1024), (32768, 512)):
    sigma = torch.rand(n, S, generator=g).cuda() * 3; z = torch.sort(torch.rand(n, S, generator=g), dim=1).values.cuda()
    rays = torch.zeros(n, 13); rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1); rays[:, 12] = 1.0
    rays = rays.cuda(); zero = torch.zeros(n, S, device="cuda")
    for name, kw in (("staged", {}), ("direct (zero noise tensor)", {"noise": zero}), ("generated noise (std 1)", {"noise_std": 1.0, "seed": 123})):
        for _ in range(2):
            ops.render_forward(sigma, z, rays, want_weights=False, **kw)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            d = ops.render_forward(sigma, z, rays, want_weights=False, **kw)[0]
        b.record(); torch.cuda.synchronize()
        print(f"{n} x {S} {name}: {a.elapsed_time(b) / 5:.3f} ms  depth sum {float(d.sum()):.6f}")
