"""L1 depth error of a trained map - the "matched L1 depth" gate of the benchmark metric.

Mirrors compute_l1_depth of the reference (analysis/compute_l1_depth.py:42-64): render every ray of a scan
through Model.forward(testing=True) (N_samples_test samples, no jitter, importance draws still random), convert
depth to metres, and average |depth - measured range| over rays with ray_range[0] < range < ray_range[1] - 0.25.
"""
import torch

from ..common.ray_utils import LidarRayDirections


@torch.no_grad()
def compute_l1_depth(lidar_pose, ray_directions: LidarRayDirections, model, ray_sampler, world_cube, ray_range, device,
                     max_rays=None, stride=None, indices=None):
    """-> mean L1 depth error in metres (float).  `lidar_pose`: a Pose; `stride`/`max_rays` subsample the scan, `indices` names the
    rays to score outright (the reference evaluates every ray; subsampling keeps tests short)."""
    scale = world_cube.scale_factor
    n = len(ray_directions)
    if indices is not None:
        idx = torch.as_tensor(indices, dtype=torch.int64).reshape(-1)
    else:
        idx = torch.arange(0, n, stride or 1)
        if max_rays is not None and idx.numel() > max_rays:
            idx = idx[torch.linspace(0, idx.numel() - 1, max_rays).long()]
    size = ray_directions._chunk_size
    err_sum, count = 0.0, 0
    T = lidar_pose.get_transformation_matrix().detach()
    for lo in range(0, idx.numel(), size):
        chunk = idx[lo:lo + size]
        rays, depths = ray_directions.build_lidar_rays(chunk, ray_range, world_cube, T)
        if rays.shape[0] == 0:
            continue
        rendered = model.render_depth(rays, ray_sampler, scale, testing=True) * float(scale)   # = forward(testing=True)["depth_fine"]
        gt = depths * float(scale)
        good = (gt > float(ray_range[0])) & (gt < float(ray_range[1]) - 0.25)
        err_sum += float((rendered[good] - gt[good]).abs().sum())
        count += int(good.sum())
    return err_sum / max(count, 1)
