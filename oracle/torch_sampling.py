"""Ray samplers as plain torch ops -- the reference's own op sequence, device-agnostic (test / baseline infrastructure).

`sampling.py` emulates the ROUNDING of torch's CPU kernels in numpy so that sample indices can be pinned bit for bit;
this module instead issues the torch ops the reference issues (src/models/ray_sampling.py:22-92, sample_pdf
src/models/rendering_tcnn.py:18-67), which is what `bench.py` times as the "PyTorch op for op" baseline on whatever
device the tensors live on (host cores, or the MI355X through PyTorch-ROCm).
"""
import torch


def sample_pdf(bins, weights, n_importance, u, eps=1e-5):
    weights = weights + eps
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, cdf.shape[-1] - 1)
    inds_sampled = torch.stack([below, above], -1).view(u.shape[0], 2 * n_importance)
    cdf_g = torch.gather(cdf, 1, inds_sampled).view(u.shape[0], n_importance, 2)
    bins_g = torch.gather(bins, 1, inds_sampled).view(u.shape[0], n_importance, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom[denom < eps] = 1
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])


def sample_uniform(rays, n_samples, perturb, u_jitter):
    near, far = rays[:, -2:-1], rays[:, -1:]
    z_steps = torch.linspace(0, 1, n_samples, device=rays.device)
    z = near * (1 - z_steps) + far * z_steps
    z = z.expand(rays.shape[0], n_samples)
    if perturb > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper = torch.cat([mid, z[:, -1:]], -1)
        lower = torch.cat([z[:, :1], mid], -1)
        z = lower + (upper - lower) * (perturb * u_jitter)
    return z


def sample_occupancy(rays, grid, n_samples, perturb, u_jitter, u_pdf):
    """rays [N,13], grid [1,1,V,V,V] -> sorted z [N, n_samples] (no gradient)"""
    with torch.no_grad():
        half = n_samples // 2
        z = sample_uniform(rays, half, perturb, u_jitter)
        pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
        n, b, _ = pts.shape
        logits = torch.nn.functional.grid_sample(grid, pts.reshape(1, 1, n, b, 3), mode="bilinear", align_corners=False).reshape(n, b)
        probs = 1.0 / (1 + torch.exp(-logits))
        probs = 2 * (probs.clamp(min=0.5, max=1.0) - 0.5)
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        fine = sample_pdf(mid, probs[:, 1:-1], half, u_pdf)
        return torch.sort(torch.cat([z, fine], -1), -1).values
