"""LiDAR ray records -- CPU oracle (test infrastructure), torch (autograd-capable).

Follows the reference:
  * get_far_val                           src/common/ray_utils.py:31-60
  * LidarRayDirections.build_lidar_rays   src/common/ray_utils.py:269-322
  * KeyFrame.build_lidar_rays (sky rays)  src/mapping/keyframe.py:71-101
  * LidarScan.get_sky_scan                src/common/sensors.py:162-167

Ray record (13 float32): [origin(3) dir(3) viewdir(3) 0 0 near far].
"""
import torch


def cube_exit_distance(origins: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """Distance along each ray to the exit of the cube [-1,1]^3; [n,3] -> [n,1].

    ray_utils.py:55-58 with no_nan=True: the direction is offset by 1e-15
    before dividing; per axis take the larger of the two clamped plane
    distances, then the smallest over axes."""
    d = dirs + 1e-15
    t_lo = ((-1.0 - origins) / d).clamp(min=0)
    t_hi = ((1.0 - origins) / d).clamp(min=0)
    per_axis = torch.maximum(t_lo, t_hi)
    return per_axis.min(dim=1, keepdim=True).values


def lidar_ray_records(directions: torch.Tensor, distances: torch.Tensor, index: torch.Tensor,
                      transform: torch.Tensor, ray_range, scale, shift,
                      keep_all: bool = False):
    """directions [3,n] (sensor frame), distances [n], index [m] int64,
    transform [4,4] (lidar->world, may require grad).

    Returns (rays [k,13], depths [k], keep [m] bool) with k = keep.sum()."""
    depths = distances[index] / scale
    local = directions[:, index]
    origin = (transform[:3, 3] + shift) / scale
    origins = origin.tile(index.shape[0], 1)
    world = (transform[:3, :3] @ local).T
    unit = world / torch.linalg.vector_norm(world, dim=1, keepdim=True)
    ones = torch.ones_like(origins[:, :1])
    near = ray_range[0] / scale * ones
    far_range = ray_range[1] / scale * ones
    far = torch.minimum(far_range, cube_exit_distance(origins, unit))
    rays = torch.cat([origins, unit, -unit, torch.zeros_like(origins[:, :2]), near, far], dim=1)
    if keep_all:
        keep = torch.ones(index.shape[0], dtype=torch.bool)
        return rays, depths, keep
    keep = (far > near + 1.0 / scale)[:, 0]
    return rays[keep], depths[keep], keep


def keyframe_ray_records(directions, distances, index, transform, ray_range, scale, shift,
                         sky_directions=None, sky_index=None):
    """keyframe.py:71-101: lidar rays followed (optionally) by sky rays whose
    'measured' depth is ray_range[1]+1 and whose pose is detached."""
    rays, depths, keep = lidar_ray_records(directions, distances, index, transform,
                                           ray_range, scale, shift)
    if sky_index is None:
        return rays, depths
    sky_dist = torch.full_like(sky_directions[0], float(ray_range[1]) + 1.0)
    s_rays, s_depths, _ = lidar_ray_records(sky_directions, sky_dist, sky_index,
                                            transform.detach(), ray_range, scale, shift)
    return torch.cat([rays, s_rays]), torch.cat([depths, s_depths])
