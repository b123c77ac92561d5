"""Frame: lidar points + poses for one instant (subset of src/common/frame.py:22-156)."""
from typing import Union

from .pose import Pose
from .sensors import LidarScan


class Frame:
    def __init__(self, image=None, lidar_points: LidarScan = None, T_lidar_to_camera: Pose = None) -> None:
        self.image = image
        self.lidar_points = LidarScan() if lidar_points is None else lidar_points
        self._lidar_to_camera = T_lidar_to_camera
        self._lidar_pose: Pose = None
        self._gt_lidar_pose: Pose = None
        self._id = -1

    def to(self, device: Union[int, str]) -> "Frame":
        self.lidar_points.to(device)
        for pose in (self._lidar_to_camera, self._lidar_pose, self._gt_lidar_pose):
            if pose is not None:
                pose.to(device)
        return self

    def get_time(self):
        return self.lidar_points.get_start_time()

    def get_lidar_pose(self) -> Pose:
        return self._lidar_pose

    def get_camera_pose(self) -> Pose:
        return self._lidar_pose * self._lidar_to_camera
