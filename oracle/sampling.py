"""Ray samplers -- CPU oracle (test infrastructure), NumPy, bit-faithful.

Follows the reference:
  * UniformRaySampler.get_samples   src/models/ray_sampling.py:22-43
  * OccGridRaySampler.get_samples   src/models/ray_sampling.py:53-92
  * sample_pdf                      src/models/rendering_tcnn.py:18-67

Random draws are *inputs* (u_jitter, u_pdf) so that the HIP kernels and this
oracle consume identical numbers (the reference draws them with torch.rand at
ray_sampling.py:38/72 and rendering_tcnn.py:48).

One deliberate difference from torch: the occupancy sigmoid uses a correctly
rounded exp (computed in float64, rounded once to float32).  torch CPU's
float32 exp is correctly rounded for ~98.9 % of inputs only, which is not
reproducible on another machine, let alone a GPU; with the correctly rounded
form the oracle and the HIP kernel agree bit-for-bit and the reference agrees
wherever its own exp happened to round correctly
(tests/test_oracle_golden.py measures the residual mismatch rate).
"""
import numpy as np
import torch

from .occupancy import trilinear_lookup_np
from .torch_rounding import cumsum_lastdim_f32, sum_lastdim_f32

f32 = np.float32


def unit_steps(count: int) -> np.ndarray:
    """torch.linspace(0, 1, count) as float32 (host table, also fed to the kernels)."""
    return torch.linspace(0, 1, count).numpy().astype(f32)


def stratified_depths(near: np.ndarray, far: np.ndarray, count: int,
                      perturb: float, u_jitter) -> np.ndarray:
    """near/far [N] -> z [N,count]; ray_sampling.py:29-41 / :59-73."""
    s = unit_steps(count)[None, :]
    near = np.asarray(near, f32)[:, None]
    far = np.asarray(far, f32)[:, None]
    z = near * (f32(1) - s) + far * s
    if perturb > 0:
        mid = f32(0.5) * (z[:, :-1] + z[:, 1:])
        upper = np.concatenate([mid, z[:, -1:]], axis=1)
        lower = np.concatenate([z[:, :1], mid], axis=1)
        jitter = f32(perturb) * np.asarray(u_jitter, f32)
        z = lower + (upper - lower) * jitter
    return z.astype(f32)


def occupancy_probs(logits: np.ndarray) -> np.ndarray:
    """ray_sampling.py:80-81 with a correctly rounded exp (see module doc)."""
    e = np.exp(-logits.astype(np.float64)).astype(f32)
    p = f32(1) / (f32(1) + e)
    return (f32(2) * (np.clip(p, f32(0.5), f32(1.0)) - f32(0.5))).astype(f32)


def inverse_cdf(bins: np.ndarray, weights: np.ndarray, u: np.ndarray, eps: float = 1e-5):
    """sample_pdf, rendering_tcnn.py:18-67.  bins [N,K+1], weights [N,K], u [N,M].

    Returns (samples [N,M] float32, inds [N,M] int64, cdf [N,K+1] float32)."""
    bins = np.asarray(bins, f32)
    u = np.ascontiguousarray(u, f32)
    w = np.asarray(weights, f32) + f32(eps)
    k = w.shape[1]
    pdf = w / sum_lastdim_f32(w)[:, None]
    cdf = np.concatenate([np.zeros((w.shape[0], 1), f32), cumsum_lastdim_f32(pdf)], axis=1)
    # searchsorted(right=True): number of cdf entries <= u
    inds = (cdf[:, None, :] <= u[:, :, None]).sum(-1).astype(np.int64)
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, k)
    c0 = np.take_along_axis(cdf, below, 1)
    c1 = np.take_along_axis(cdf, above, 1)
    b0 = np.take_along_axis(bins, below, 1)
    b1 = np.take_along_axis(bins, above, 1)
    denom = c1 - c0
    denom = np.where(denom < f32(eps), f32(1), denom)
    samples = b0 + (u - c0) / denom * (b1 - b0)
    return samples.astype(f32), inds, cdf


def sample_uniform(rays: np.ndarray, n_samples: int, perturb: float, u_jitter) -> np.ndarray:
    """UniformRaySampler: rays [N,13] -> z [N,n_samples]."""
    return stratified_depths(rays[:, -2], rays[:, -1], n_samples, perturb, u_jitter)


def sample_occupancy(rays: np.ndarray, grid: np.ndarray, n_samples: int, perturb: float,
                     u_jitter, u_pdf, probs_override=None, return_stages: bool = False):
    """OccGridRaySampler: rays [N,13], grid [V,V,V] -> sorted z [N,n_samples].

    `probs_override` lets a test substitute the reference's own point_probs so
    that index identity can be asserted stage-wise."""
    rays = np.asarray(rays, f32)
    half = n_samples // 2
    o = rays[:, 0:3]
    d = rays[:, 3:6]
    z = stratified_depths(rays[:, -2], rays[:, -1], half, perturb, u_jitter)
    pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    if probs_override is None:
        probs = occupancy_probs(trilinear_lookup_np(grid, pts))
    else:
        probs = np.asarray(probs_override, f32)
    mids = f32(0.5) * (z[:, :-1] + z[:, 1:])
    fine, inds, cdf = inverse_cdf(mids, probs[:, 1:-1], u_pdf)
    merged = np.sort(np.concatenate([z, fine], axis=1), axis=1)
    if return_stages:
        return merged, dict(coarse=z, probs=probs, cdf=cdf, inds=inds, fine=fine)
    return merged
