"""WorldCube and 6-vector <-> 4x4 conversions.

Mirrors the part of the reference's src/common/pose_utils.py that the mapping hot path touches:
WorldCube (:24-57) and tensor_to_transform (:288-302).  The reference delegates the axis-angle ->
matrix map to pytorch3d 0.7.2 (absent here); pytorch3d's published algorithm (axis-angle -> unit
quaternion with the small-angle series, quaternion -> matrix) is written out in stock torch ops so
that the pose-Jacobian tail stays in torch autograd on whatever device the pose lives on.
"""
from dataclasses import dataclass

import torch


@dataclass
class WorldCube:
    """Shift and scale that map the scene into the cube [-1,1]^3 (pose_utils.py:24-57)."""
    scale_factor: torch.Tensor
    shift: torch.Tensor

    def to(self, device, clone=False) -> "WorldCube":
        shift = self.shift if isinstance(self.shift, torch.Tensor) else torch.tensor(self.shift, dtype=torch.float32)
        scale = self.scale_factor if isinstance(self.scale_factor, torch.Tensor) else torch.tensor(float(self.scale_factor))
        if clone:
            return WorldCube(scale.to(device, copy=True), shift.to(device, copy=True))
        self.shift = shift.to(device)
        self.scale_factor = scale.to(device)
        return self

    def as_dict(self) -> dict:
        return {"scale_factor": float(self.scale_factor), "shift": [float(s) for s in self.shift.cpu()]}


def axis_angle_to_matrix(aa: torch.Tensor) -> torch.Tensor:
    """[...,3] -> [...,3,3]"""
    theta = torch.linalg.vector_norm(aa, dim=-1, keepdim=True)
    half = 0.5 * theta
    small = theta.abs() < 1e-6
    denom = torch.where(small, torch.ones_like(theta), theta)
    k = torch.where(small, 0.5 - theta * theta / 48.0, torch.sin(half) / denom)
    q = torch.cat([torch.cos(half), aa * k], dim=-1)
    r, i, j, kk = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    m = torch.stack([
        1 - two_s * (j * j + kk * kk), two_s * (i * j - kk * r), two_s * (i * kk + j * r),
        two_s * (i * j + kk * r), 1 - two_s * (i * i + kk * kk), two_s * (j * kk - i * r),
        two_s * (i * kk - j * r), two_s * (j * kk + i * r), 1 - two_s * (i * i + j * j)], dim=-1)
    return m.reshape(aa.shape[:-1] + (3, 3))


def matrix_to_axis_angle(R: torch.Tensor) -> torch.Tensor:
    """[3,3] -> [3] via the unit quaternion (largest-component branch), in float64.
    Used only when a Pose is created from a matrix; not on the hot path."""
    m = R.detach().double().cpu()
    t = m.trace()
    cand = torch.stack([1 + t, 1 + 2 * m[0, 0] - t, 1 + 2 * m[1, 1] - t, 1 + 2 * m[2, 2] - t])
    i = int(torch.argmax(cand))
    if i == 0:
        q = torch.stack([cand[0], m[2, 1] - m[1, 2], m[0, 2] - m[2, 0], m[1, 0] - m[0, 1]])
    elif i == 1:
        q = torch.stack([m[2, 1] - m[1, 2], cand[1], m[0, 1] + m[1, 0], m[0, 2] + m[2, 0]])
    elif i == 2:
        q = torch.stack([m[0, 2] - m[2, 0], m[0, 1] + m[1, 0], cand[2], m[1, 2] + m[2, 1]])
    else:
        q = torch.stack([m[1, 0] - m[0, 1], m[0, 2] + m[2, 0], m[1, 2] + m[2, 1], cand[3]])
    q = q / q.norm()
    if q[0] < 0:
        q = -q
    v = q[1:]
    n = v.norm()
    if n < 1e-12:
        return (2 * v).to(R.dtype).to(R.device)
    angle = 2 * torch.atan2(n, q[0])
    return (v / n * angle).to(R.dtype).to(R.device)


def tensor_to_transform(t: torch.Tensor) -> torch.Tensor:
    """[6] or [N,6] = [translation, axis-angle] -> [4,4] or [N,4,4] (pose_utils.py:288-302)."""
    single = t.dim() == 1
    if single:
        t = t[None]
    R = axis_angle_to_matrix(t[:, 3:])
    top = torch.cat([R, t[:, :3, None]], dim=2)
    bottom = torch.zeros(t.shape[0], 1, 4, dtype=t.dtype, device=t.device)
    bottom[:, 0, 3] = 1
    T = torch.cat([top, bottom], dim=1)
    return T[0] if single else T


def transform_to_tensor(T: torch.Tensor, device=None) -> torch.Tensor:
    """[4,4] -> [6] (pose_utils.py:255-282)."""
    out = torch.cat([T[:3, 3].detach(), matrix_to_axis_angle(T[:3, :3]).to(T.device)]).float()
    return out.to(device) if device is not None else out
