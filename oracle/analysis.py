"""Inference path + matched-quality metric -- CPU oracle (test infrastructure).

Follows the reference:
  * Model.forward(testing=True)   src/models/model_tcnn.py:70-105 (N_samples_test samples, perturb 0; the importance draw of
                                  sample_pdf AND the density noise stay random: raw_noise_std is passed in test mode too, :92)
  * compute_l1_depth              analysis/compute_l1_depth.py:42-64 (metres; rays with ray_range[0] < range < ray_range[1]-0.25)
Pinned by tests/golden/g12_checkpoint_l1_depth.npz (captured from the reference, tests/golden/make_golden2.py).
"""
import torch

from . import network as NW
from . import poses as P
from . import rays as R
from . import render as RD
from . import sampling as SP
from . import torch_sampling as TS


def render_depth(spec, params, grid, rays, n_samples, u_pdf, noise=None, sampler="numpy"):
    """rays [N,13] -> rendered depth [N] (cube units) as Model.forward(testing=True) returns in 'depth_fine'.
    grid [V,V,V] occupancy logits; u_pdf [N, n_samples/2]; noise [N, n_samples] already scaled by raw_noise_std, or None.
    sampler: "numpy" = rounding-exact emulation (CPU), "torch" = the reference's torch op sequence (any device)."""
    if sampler == "torch":
        z = TS.sample_occupancy(rays.detach(), grid[None, None], n_samples, 0.0, None, u_pdf)
    else:
        z = torch.from_numpy(SP.sample_occupancy(rays.detach().numpy(), grid.numpy(), n_samples, 0.0, None, u_pdf.numpy()))
    xyz = RD.sample_points(rays, z)
    sigma = NW.density(spec, params, xyz.reshape(-1, 3)).reshape(z.shape)
    return RD.composite(sigma, z, rays[:, 3:6], rays[:, -1:], noise)["depth"], z


def l1_depth(spec, params, grid, directions, distances, pose6, scale, shift, ray_range, n_samples, u_pdf, noise=None, sampler="numpy"):
    """-> (mean |rendered depth - measured range| in metres, per-ray depth in cube units)"""
    T = P.transform_from_pose6(pose6)
    idx = torch.arange(directions.shape[1], device=directions.device)
    rays, depths, keep = R.lidar_ray_records(directions, distances, idx, T, ray_range, scale, shift)
    assert bool(keep.all()), "the reference's chunk bookkeeping assumes that no ray is dropped"
    depth, _ = render_depth(spec, params, grid, rays.float(), n_samples, u_pdf, noise, sampler)
    good = (distances > ray_range[0]) & (distances < ray_range[1] - 0.25)
    return float(torch.nn.functional.l1_loss(depth[good] * scale, distances[good])), depth
