"""6-vector pose -> 4x4 transform -- CPU oracle (test infrastructure).

Follows the reference's tensor_to_transform (src/common/pose_utils.py:288-302)
which calls pytorch3d.transforms.axis_angle_to_matrix (pytorch3d 0.7.2, pinned
at docker/container_dockerhub.Dockerfile:67; not present in /root/reference).
pytorch3d's published algorithm is restated: axis-angle -> unit quaternion
(with the small-angle series 0.5 - a^2/48 below 1e-6 rad) -> rotation matrix.
Parity for this function is UNPINNED (no reference fixture exists);
tests check orthonormality and agreement with scipy's Rotation.
"""
import torch


def rotation_from_axis_angle(aa: torch.Tensor) -> torch.Tensor:
    """aa [...,3] -> R [...,3,3], differentiable."""
    theta = torch.linalg.vector_norm(aa, dim=-1, keepdim=True)
    half = 0.5 * theta
    tiny = theta.abs() < 1e-6
    safe = torch.where(tiny, torch.ones_like(theta), theta)
    k = torch.where(tiny, 0.5 - theta * theta / 48.0, torch.sin(half) / safe)
    q = torch.cat([torch.cos(half), aa * k], dim=-1)
    w, x, y, z = q.unbind(-1)
    s2 = 2.0 / (q * q).sum(-1)
    rows = torch.stack([
        1 - s2 * (y * y + z * z), s2 * (x * y - z * w), s2 * (x * z + y * w),
        s2 * (x * y + z * w), 1 - s2 * (x * x + z * z), s2 * (y * z - x * w),
        s2 * (x * z - y * w), s2 * (y * z + x * w), 1 - s2 * (x * x + y * y)], dim=-1)
    return rows.reshape(aa.shape[:-1] + (3, 3))


def transform_from_pose6(p: torch.Tensor) -> torch.Tensor:
    """p [6] = [t(3), axis-angle(3)] -> T [4,4], differentiable."""
    R = rotation_from_axis_angle(p[3:6])
    top = torch.cat([R, p[0:3].reshape(3, 1)], dim=1)
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=p.dtype, device=p.device)
    return torch.cat([top, bottom], dim=0)
