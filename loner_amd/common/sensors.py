"""LidarScan: the keyframe point buffer (SoA), mirroring src/common/sensors.py:57-167."""
from typing import Union

import torch


class LidarScan:
    """ray_directions [3,n] unit vectors (sensor frame), distances [n] metres, timestamps [n];
    optional sky_rays [3,m] (directions that hit nothing)."""

    def __init__(self, ray_directions: torch.Tensor = None, distances: torch.Tensor = None,
                 timestamps: torch.Tensor = None, sky_rays: torch.Tensor = None) -> None:
        self.ray_directions = torch.Tensor() if ray_directions is None else ray_directions
        self.distances = torch.Tensor() if distances is None else distances
        self.timestamps = torch.Tensor() if timestamps is None else timestamps
        self.sky_rays = sky_rays

    def __len__(self) -> int:
        return self.timestamps.shape[0]

    def get_start_time(self) -> torch.Tensor:
        return self.timestamps[0]

    def get_end_time(self) -> torch.Tensor:
        return self.timestamps[-1]

    def clone(self) -> "LidarScan":
        return LidarScan(self.ray_directions.clone(), self.distances.clone(), self.timestamps.clone(),
                         self.sky_rays.clone() if self.sky_rays is not None else None)

    def to(self, device: Union[int, str]) -> "LidarScan":
        self.ray_directions = self.ray_directions.to(device)
        self.distances = self.distances.to(device)
        self.timestamps = self.timestamps.to(device)
        return self

    def get_sky_scan(self, distance: float) -> "LidarScan":
        sky = self.sky_rays
        return LidarScan(sky, torch.full_like(sky[0], float(distance)), torch.full_like(sky[0], float(self.timestamps[-1])))
