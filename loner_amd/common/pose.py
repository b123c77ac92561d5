"""Pose: a possibly optimisable rigid transform, stored as a 6-vector [t, axis-angle].

Mirrors the members of the reference's src/common/pose.py that the mapping path uses
(constructor :32-60, set_fixed :68, to :72, detach :80, clone :120, get_transformation_matrix
:140-144, get_pose_tensor :147-150, get_translation/rotation/axis_angle).
"""
from typing import Union

import torch

from .pose_utils import matrix_to_axis_angle, tensor_to_transform, transform_to_tensor


class Pose:
    def __init__(self, transformation_matrix: torch.Tensor = None, pose_tensor: torch.Tensor = None,
                 fixed: bool = None, requires_tensor: bool = False):
        if transformation_matrix is None and pose_tensor is None:
            transformation_matrix = torch.eye(4)
        if fixed is None:
            fixed = not (pose_tensor if transformation_matrix is None else transformation_matrix).requires_grad
        if pose_tensor is not None:
            self._pose_tensor = pose_tensor
            self._pose_tensor.requires_grad_(not fixed)
            transformation_matrix = tensor_to_transform(self._pose_tensor).float()
        elif requires_tensor:
            self._pose_tensor = transform_to_tensor(transformation_matrix).float()
            self._pose_tensor.requires_grad_(not fixed)
            transformation_matrix = tensor_to_transform(self._pose_tensor).float()
        else:
            self._pose_tensor = None
            transformation_matrix = transformation_matrix.float()
        self._transformation_matrix = transformation_matrix.detach() if fixed else transformation_matrix

    def __repr__(self) -> str:
        return str(self.get_transformation_matrix())

    def set_fixed(self, fixed: bool = True) -> None:
        self.get_pose_tensor().requires_grad_(not fixed)

    def to(self, device: Union[str, int]) -> "Pose":
        if self._pose_tensor is not None:
            rg = self._pose_tensor.requires_grad
            self._pose_tensor = self._pose_tensor.detach().to(device).requires_grad_(rg)
        self._transformation_matrix = self._transformation_matrix.detach().to(device)
        return self

    def detach(self) -> "Pose":
        return Pose(self.get_transformation_matrix().detach())

    def clone(self, fixed=None, requires_tensor=False) -> "Pose":
        if fixed is None:
            fixed = not self.get_transformation_matrix().requires_grad
        return Pose(self.get_transformation_matrix().detach().clone(), fixed=fixed, requires_tensor=requires_tensor)

    def __mul__(self, other: "Pose") -> "Pose":
        return Pose(self.get_transformation_matrix() @ other.get_transformation_matrix())

    def inv(self) -> "Pose":
        return Pose(self.get_transformation_matrix().inverse())

    def get_transformation_matrix(self) -> torch.Tensor:
        if self._pose_tensor is None or not self._pose_tensor.requires_grad:
            if self._pose_tensor is not None:
                return tensor_to_transform(self._pose_tensor.detach())
            return self._transformation_matrix
        return tensor_to_transform(self._pose_tensor)

    def get_pose_tensor(self) -> torch.Tensor:
        if self._pose_tensor is None:
            self._pose_tensor = transform_to_tensor(self.get_transformation_matrix())
        return self._pose_tensor

    def get_translation(self) -> torch.Tensor:
        if self._pose_tensor is not None:
            return self._pose_tensor[:3]
        return self.get_transformation_matrix()[:3, 3]

    def get_rotation(self) -> torch.Tensor:
        return self.get_transformation_matrix()[:3, :3]

    def get_axis_angle(self) -> torch.Tensor:
        if self._pose_tensor is not None:
            return self._pose_tensor[3:]
        return matrix_to_axis_angle(self.get_rotation())
