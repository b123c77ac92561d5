"""Pose holder used on the host side of the mapping path.

The optimiser itself never differentiates through this class: during a phase the poses of the window live in ONE
device tensor [K,6] (translation, axis-angle) that `lnr_pose_forward/backward` and the fused Adam kernel work on
(mapping/optimizer.py), and the result is written back into the keyframes' `Pose` objects once per phase.  What the
callers on either side of the hot path need from a pose is therefore small, and this class provides exactly that
under the member names the reference's `Mapper`, `KeyFrame` and analysis scripts call (src/common/pose.py:
constructor keywords, set_fixed, to, detach, clone, `*`, inv, get_transformation_matrix, get_pose_tensor,
get_translation, get_rotation, get_axis_angle).  In an integration the reference's own class is used
(INTEGRATION.md); this one exists so that the package is self-contained on the GPU box.

Representation: either a 4x4 matrix (`_mat`) or a 6-vector (`_vec6`), converted lazily.  "Free" poses keep
autograd alive through `get_transformation_matrix()` (the API-parity path, `KeyFrame.build_lidar_rays`, relies on
that); fixed ones hand out constants.
"""
from typing import Optional, Union

import torch

from . import pose_utils as PU


def _as_f32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.float32 else t.float()


class Pose:
    def __init__(self, transformation_matrix: Optional[torch.Tensor] = None, pose_tensor: Optional[torch.Tensor] = None,
                 fixed: Optional[bool] = None, requires_tensor: bool = False):
        source = pose_tensor if pose_tensor is not None else transformation_matrix
        if source is None:
            source = torch.eye(4)
            transformation_matrix = source
        free = bool(source.requires_grad) if fixed is None else not fixed
        self._vec6 = None
        if pose_tensor is not None:
            self._vec6 = pose_tensor                       # the caller's tensor IS the optimisation variable
        elif requires_tensor:
            self._vec6 = _as_f32(PU.transform_to_tensor(transformation_matrix))
        if self._vec6 is not None:
            self._vec6.requires_grad_(free)
            mat = _as_f32(PU.tensor_to_transform(self._vec6))
        else:
            mat = _as_f32(transformation_matrix)
        self._mat = mat if free else mat.detach()

    # ---- flags / placement ---------------------------------------------------------------------
    def set_fixed(self, fixed: bool = True) -> None:
        self.get_pose_tensor().requires_grad_(not fixed)

    def to(self, device: Union[str, int]) -> "Pose":
        if self._vec6 is not None:
            free = self._vec6.requires_grad
            self._vec6 = self._vec6.detach().to(device).requires_grad_(free)
        self._mat = self._mat.detach().to(device)
        return self

    # ---- views ---------------------------------------------------------------------------------
    def get_transformation_matrix(self) -> torch.Tensor:
        if self._vec6 is None:
            return self._mat
        if self._vec6.requires_grad:
            return PU.tensor_to_transform(self._vec6)      # differentiable w.r.t. the 6-vector
        return PU.tensor_to_transform(self._vec6.detach())  # the vector may have been stepped since construction

    def get_pose_tensor(self) -> torch.Tensor:
        if self._vec6 is None:
            self._vec6 = PU.transform_to_tensor(self._mat)
        return self._vec6

    def get_translation(self) -> torch.Tensor:
        return self._vec6[:3] if self._vec6 is not None else self._mat[:3, 3]

    def get_rotation(self) -> torch.Tensor:
        return self.get_transformation_matrix()[:3, :3]

    def get_axis_angle(self) -> torch.Tensor:
        return self._vec6[3:] if self._vec6 is not None else PU.matrix_to_axis_angle(self._mat[:3, :3])

    # ---- algebra / copies ----------------------------------------------------------------------
    def detach(self) -> "Pose":
        return Pose(self.get_transformation_matrix().detach())

    def clone(self, fixed=None, requires_tensor=False) -> "Pose":
        mat = self.get_transformation_matrix()
        return Pose(mat.detach().clone(), fixed=(not mat.requires_grad) if fixed is None else fixed,
                    requires_tensor=requires_tensor)

    def __mul__(self, other: "Pose") -> "Pose":
        return Pose(self.get_transformation_matrix() @ other.get_transformation_matrix())

    def inv(self) -> "Pose":
        return Pose(torch.linalg.inv(self.get_transformation_matrix()))

    def __repr__(self) -> str:
        return f"Pose({self.get_transformation_matrix()})"
