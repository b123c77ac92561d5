"""LiDAR ray construction (mirrors src/common/ray_utils.py:31-60 and :252-322 of the reference).

`LidarRayDirections.build_lidar_rays` keeps the reference's signature and return value but runs on
the MI355X: the scan (the keyframe point buffer) is kept resident in HBM, rays are built by
`lnr_build_lidar_rays`, and the gradient with respect to the 4x4 lidar pose is produced by
`lnr_lidar_rays_backward`, so that the 6-vector pose tail stays in stock torch autograd
(tensor_to_transform) exactly as in the reference.
"""
import weakref

import torch

from .. import ops
from .pose_utils import WorldCube
from .sensors import LidarScan


def mapping_device():
    """The HIP device the mapper runs on (the reference hard-codes device 0, mapper.py:62-66)."""
    if not torch.cuda.is_available():
        raise RuntimeError("loner_amd: no MI355X/HIP device visible - the mapping hot path has no CPU implementation")
    return torch.device("cuda", torch.cuda.current_device())


_scan_cache = weakref.WeakKeyDictionary()      # scan object -> (key, directions, distances) on the device


def device_scan(scan: LidarScan, device):
    """HBM-resident copy of a scan's SoA buffers (uploaded once per keyframe).  The cache lives HERE, keyed weakly by the scan
    object - nothing is written onto the caller's LidarScan (in an integration that is the reference's own class)."""
    key = (scan.ray_directions.data_ptr(), scan.distances.data_ptr(), str(device))
    try:
        cache = _scan_cache.get(scan)
    except TypeError:                  # an object that cannot be weakly referenced: no caching
        cache = None
    if cache is None or cache[0] != key:
        dirs = scan.ray_directions.detach().to(device=device, dtype=torch.float32).contiguous()
        dist = scan.distances.detach().to(device=device, dtype=torch.float32).contiguous()
        cache = (key, dirs, dist)
        try:
            _scan_cache[scan] = cache
        except TypeError:
            pass
    return cache[1], cache[2]


def get_far_val(pts_o: torch.Tensor, pts_d: torch.Tensor, no_nan: bool = False):
    """Distance to the exit of the cube [-1,1]^3 (ray_utils.py:31-60); small helper kept in torch ops."""
    if no_nan:
        pts_d = pts_d + 1e-15
    t_lo = ((-1.0 - pts_o) / pts_d).clamp(min=0)
    t_hi = ((1.0 - pts_o) / pts_d).clamp(min=0)
    return torch.maximum(t_lo, t_hi).min(dim=1, keepdim=True).values


class _BuildLidarRays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lidar_pose, dirs_dev, dist_dev, index_dev, ray_range, scale, shift):
        T12 = lidar_pose.detach()[:3, :4].to(device=dirs_dev.device, dtype=torch.float32).contiguous().reshape(12)
        rays, depths, keep = ops.build_lidar_rays(dirs_dev, dist_dev, index_dev, T12, ray_range, scale, shift)
        ctx.save_for_backward(rays, index_dev, dirs_dev, T12)
        ctx.scale = float(scale)
        ctx.pose_device = lidar_pose.device
        ctx.mark_non_differentiable(depths, keep)
        return rays, depths, keep

    @staticmethod
    def backward(ctx, d_rays, _d_depths, _d_keep):
        rays, index_dev, dirs_dev, T12 = ctx.saved_tensors
        n = rays.shape[0]
        seg = torch.tensor([0, n], device=rays.device, dtype=torch.int32)
        dT = ops.lidar_rays_backward(d_rays.contiguous(), rays, index_dev, seg, [dirs_dev], T12.reshape(1, 12), ctx.scale)
        grad = torch.zeros(4, 4, device=rays.device, dtype=torch.float32)
        grad[:3, :4] = dT.reshape(3, 4)
        return grad.to(ctx.pose_device), None, None, None, None, None, None


class LidarRayDirections:
    def __init__(self, lidar_scan: LidarScan, chunk_size=512):
        self.lidar_scan = lidar_scan
        self._chunk_size = chunk_size
        self.num_chunks = -(-self.lidar_scan.ray_directions.shape[1] // self._chunk_size)

    def __len__(self):
        return self.lidar_scan.ray_directions.shape[1]

    def fetch_chunk_rays(self, chunk_idx: int, pose, world_cube: WorldCube, ray_range, device=None):
        start = chunk_idx * self._chunk_size
        end = min(len(self), (chunk_idx + 1) * self._chunk_size)
        return self.build_lidar_rays(torch.arange(start, end), ray_range, world_cube, pose.get_transformation_matrix())[0]

    def build_lidar_rays(self, lidar_indices: torch.Tensor, ray_range: torch.Tensor, world_cube: WorldCube,
                         lidar_pose: torch.Tensor, ignore_world_cube: bool = False):
        """-> (rays [k,13], depths [k]) on the HIP device; rays carry the gradient w.r.t. lidar_pose."""
        dev = mapping_device()
        dirs_dev, dist_dev = device_scan(self.lidar_scan, dev)
        index_dev = lidar_indices.to(device=dev, dtype=torch.int64)
        shift = world_cube.shift.detach().cpu().reshape(-1).tolist()
        rr = [float(ray_range[0]), float(ray_range[1])]
        rays, depths, keep = _BuildLidarRays.apply(lidar_pose, dirs_dev, dist_dev, index_dev, rr,
                                                   float(world_cube.scale_factor), shift)
        if ignore_world_cube:
            return rays, depths
        # the reference asserts that origins are inside the cube (ray_utils.py:301-303)
        if rays.shape[0] and bool((rays[0, :3].abs() > 1).any()):
            raise AssertionError("ray origins are outside the world cube")
        valid = keep.bool()
        return rays[valid], depths[valid]
