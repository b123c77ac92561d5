"""KeyFrame: a Frame plus optimisation metadata (src/mapping/keyframe.py:24-135, lidar part)."""
import torch

from ..common.frame import Frame
from ..common.pose import Pose
from ..common.pose_utils import WorldCube
from ..common.ray_utils import LidarRayDirections
from ..common.sensors import LidarScan


class KeyFrame:
    def __init__(self, frame: Frame, device=None) -> None:
        self._frame = frame.to(device) if device is not None else frame
        self._device = device
        self._tracked_lidar_pose: Pose = frame.get_lidar_pose().clone()
        self.is_anchored = False
        self.lidar_loss_distribution = None

    def to(self, device) -> "KeyFrame":
        self._frame.to(device)
        self._device = device
        return self

    def get_lidar_pose(self) -> Pose:
        return self._frame.get_lidar_pose()

    def get_lidar_scan(self) -> LidarScan:
        return self._frame.lidar_points

    def get_time(self):
        return self._frame.get_time()

    def build_lidar_rays(self, lidar_indices: torch.Tensor, ray_range: torch.Tensor, world_cube: WorldCube,
                         use_gt_poses: bool = False, ignore_world_cube: bool = False,
                         sky_indices: torch.Tensor = None):
        """keyframe.py:71-101: lidar rays (pose differentiable) followed by sky rays (pose detached)."""
        scan = self.get_lidar_scan()
        pose = self._frame._gt_lidar_pose if use_gt_poses else self._frame.get_lidar_pose()
        T = pose.get_transformation_matrix()
        rays, depths = LidarRayDirections(scan).build_lidar_rays(lidar_indices, ray_range, world_cube, T, ignore_world_cube)
        if sky_indices is not None:
            sky_scan = scan.get_sky_scan(float(ray_range[1]) + 1)
            s_rays, s_depths = LidarRayDirections(sky_scan).build_lidar_rays(sky_indices, ray_range, world_cube,
                                                                            T.detach(), ignore_world_cube)
            rays = torch.cat((rays, s_rays))
            depths = torch.cat((depths, s_depths))
        return rays, depths

    def get_pose_state(self) -> dict:
        return {
            "timestamp": torch.as_tensor(self.get_time()).detach().cpu().clone(),
            "lidar_to_camera": (self._frame._lidar_to_camera.get_pose_tensor().detach().cpu().clone()
                                if self._frame._lidar_to_camera is not None else None),
            "lidar_pose": self._frame.get_lidar_pose().get_pose_tensor().detach().cpu().clone(),
            "gt_lidar_pose": (self._frame._gt_lidar_pose.get_pose_tensor().detach().cpu().clone()
                              if self._frame._gt_lidar_pose is not None else None),
            "tracked_pose": self._tracked_lidar_pose.get_pose_tensor().detach().cpu().clone(),
        }
