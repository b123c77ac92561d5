"""Synthetic LiDAR data for tests and bench.py.

One scan = 64 beams x 1024 azimuths (65 536 rays), elevation -22.5..+22.5 deg
(cfg/fusion_portable/canteen.yaml:28 in the reference), beam-major, ranges from
an analytic scene (box room 40 x 30 x 8 m with one sphere of radius 3 m) so
that exact depths exist.  The world cube is the one the reference derives for
the canteen sequence (src/common/pose_utils.py:159-261 with
cfg/fusion_portable/canteen.yaml:9-21): scale 85.7614, shift [7.5, 5, 0].
"""
import math

import numpy as np
import torch

BOX_MIN = (-20.0, -15.0, -2.0)
BOX_MAX = (20.0, 15.0, 6.0)
SPHERE_C = (8.0, 3.0, 0.0)
SPHERE_R = 3.0
RAY_RANGE = (1.0, 50.0)
WINDOW_HALF_Y = 4.0
WINDOW_Z = (0.0, 3.0)
WINDOW_FAR_X = 75.0


def world_cube(bbox=((-25.0, 10.0), (-25.0, 15.0), (-10.0, 10.0)), max_range=50.0, padding=0.3):
    """(scale_factor, shift[3]) of the cube enclosing bbox grown by max_range."""
    lo = np.array([b[0] - max_range for b in bbox], np.float32)
    hi = np.array([b[1] + max_range for b in bbox], np.float32)
    origin = lo + (hi - lo) / np.float32(2)
    scale = np.linalg.norm(hi - lo) / (2 * np.sqrt(np.float32(3))) * (1 + padding)
    return float(np.float32(scale)), (-origin).astype(np.float32)


def lidar_pattern(beams=64, azimuths=1024, fov_deg=(-22.5, 22.5)):
    """-> (directions [3, beams*azimuths] float32 unit vectors, timestamps [n])."""
    el = torch.deg2rad(torch.linspace(fov_deg[0], fov_deg[1], beams, dtype=torch.float64))
    az = 2 * math.pi * torch.arange(azimuths, dtype=torch.float64) / azimuths
    ce, se = torch.cos(el)[:, None], torch.sin(el)[:, None]
    d = torch.stack([ce * torch.cos(az)[None, :], ce * torch.sin(az)[None, :], se.expand(-1, azimuths)], 0)
    d = d.reshape(3, -1).float()
    return d.contiguous(), torch.linspace(0, 0.1, d.shape[1])


def trajectory_pose6(count=8, step=0.3, yaw_deg=2.0):
    """Straight line, `step` m per keyframe, `yaw_deg` per keyframe; [count,6]."""
    out = torch.zeros(count, 6)
    for i in range(count):
        out[i, 0] = step * i
        out[i, 5] = math.radians(yaw_deg) * i
    return out


def scene_ranges(directions: torch.Tensor, transform: torch.Tensor) -> torch.Tensor:
    """Exact range of every ray (sensor-frame `directions` [3,n], pose `transform` [4,4])."""
    T = transform.detach().double()
    o = T[:3, 3]
    d = (T[:3, :3] @ directions.double()).T
    lo = torch.tensor(BOX_MIN, dtype=torch.float64)
    hi = torch.tensor(BOX_MAX, dtype=torch.float64)
    safe = torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)
    t_box = torch.maximum((lo - o) / safe, (hi - o) / safe).min(dim=1).values
    oc = o - torch.tensor(SPHERE_C, dtype=torch.float64)
    b = (d * oc).sum(1)
    c = (oc * oc).sum() - SPHERE_R ** 2
    disc = b * b - c
    t_s = torch.where(disc > 0, -b - torch.sqrt(disc.clamp(min=0)), torch.full_like(b, float("inf")))
    t_s = torch.where(t_s > 0, t_s, torch.full_like(b, float("inf")))
    t = torch.minimum(t_box, t_s)
    # an open window in the +x wall: rays through it continue to the plane x = WINDOW_FAR_X, i.e.
    # beyond the 50 m ray range, which makes them "transparent" rays for the loss.
    hit = o + d * t_box[:, None]
    through = (t_box <= t_s) & (hit[:, 0] > BOX_MAX[0] - 1e-6) & (hit[:, 1].abs() < WINDOW_HALF_Y) \
        & (hit[:, 2] > WINDOW_Z[0]) & (hit[:, 2] < WINDOW_Z[1])
    t_far = (WINDOW_FAR_X - o[0]) / safe[:, 0]
    t = torch.where(through, t_far, t)
    return t.float()
