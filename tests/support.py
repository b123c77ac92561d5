"""Helpers shared by the GPU tests and the fixture generators (tests/golden/make_golden*.py).  No reference code, no GPU."""
import torch

SMALL_ENC = dict(otype="HashGrid", n_levels=4, n_features_per_level=2, log2_hashmap_size=12, base_resolution=8)
SMALL_NET = dict(activation="ReLU", n_neurons=32, n_hidden_layers=1, otype="FullyFusedMLP", output_activation="None")


def small_settings(n_rays, n_samples, voxel=32, n_test=None):
    """default_optimizer_settings() with the small density network of the G8/G9/G11/G12 fixtures"""
    from loner_amd.common.settings import default_optimizer_settings
    s = default_optimizer_settings()
    mc = s["model_config"]
    mc["model"]["nerf_config"]["pos_encoding_sigma"] = dict(SMALL_ENC)
    mc["model"]["nerf_config"]["sigma_network"] = dict(SMALL_NET)
    mc["model"]["nerf_config"]["pos_encoding_intensity"]["log2_hashmap_size"] = 10
    mc["model"]["render"]["N_samples_train"] = n_samples
    if n_test is not None:
        mc["model"]["render"]["N_samples_test"] = n_test
    mc["model"]["occ_model"]["voxel_size"] = voxel
    s["num_samples"]["lidar"] = n_rays
    s["num_samples"]["sky"] = 0
    return s


# G13 (tests/golden/make_golden3.py): the reduced window of the matched-L1 fixture
G13 = dict(n_rays=256, n_samples=128, n_test=256, n_l1_rays=512, pose_noise_seed=41, init_seed=1234)


class SeededReplay:
    """The random draws of a recorded reference run, regenerated from its seed: every draw of the reference came from torch's
    global CPU generator in a fixed order, so a private generator with the same seed issuing the same calls yields the same values
    (make_golden3.py verified that for the recorded run).  Interface of the `draws` hook (Optimizer.set_draws / sampler.set_draws);
    `kinds` / `args` are the recorded call signatures, checked call by call; `checksum` accumulates the values handed out."""

    def __init__(self, seed, kinds, args):
        self.gen = torch.Generator().manual_seed(int(seed))
        self.kinds, self.args, self.i, self.checksum = [int(k) for k in kinds], [tuple(int(v) for v in a) for a in args], 0, 0.0

    def _next(self, kind, a):
        assert self.i < len(self.kinds), "more draws than the reference made"
        assert (self.kinds[self.i], self.args[self.i]) == (kind, a), (self.i, self.kinds[self.i], self.args[self.i], kind, a)
        self.i += 1

    def _out(self, t):
        self.checksum += float(t.double().sum())
        return t

    def ray_index(self, n_points, count):
        self._next(0, (int(n_points), int(count), 0))
        return self._out(torch.randint(0, int(n_points), (int(count),), generator=self.gen))

    sky_index = ray_index

    def jitter(self, n, h):
        self._next(1, (0, int(n), int(h)))
        return self._out(torch.rand(int(n), int(h), generator=self.gen))

    pdf = jitter

    def noise(self, n, s):
        self._next(2, (0, int(n), int(s)))
        return self._out(torch.randn(int(n), int(s), generator=self.gen))


def sky_directions(n=40, seed=3):
    """unit vectors pointing up and outwards, sensor frame [3,n] (what the sky segmentation hands to LidarScan.sky_rays)"""
    gen = torch.Generator().manual_seed(seed)
    v = torch.randn(3, n, generator=gen)
    v[2] = v[2].abs() + 0.5
    return torch.nn.functional.normalize(v, dim=0)


def l1_scan_subset(n=512, total=65536):
    """ray indices of the synthetic scan used by the compute_l1_depth fixture (every 128th ray, offset 7)"""
    return torch.arange(7, total, total // n)[:n]


def write_repo_checkpoint(path, seed=5):
    """A checkpoint as the reference's Mapper.build_ckpt would write it (mapper.py:161-175), with the state_dicts of THIS
    repo's Model / OccupancyGridModel: "trained-like" parameters (tables x3000, random occupancy logits) from fixed seeds.
    Building a state_dict needs no GPU.  -> dict(sigma_params, occ_grid)"""
    from loner_amd.models.model_tcnn import Model, OccupancyGridModel
    s = small_settings(48, 64, n_test=256)
    mc = s.model_config.model
    torch.manual_seed(seed)
    model = Model(mc)
    occ = OccupancyGridModel(mc.occ_model)
    gen = torch.Generator().manual_seed(seed + 1)
    sig = model.nerf_model._model_sigma
    with torch.no_grad():
        sig.params[sig.spec.n_mlp_params:] *= 3000.0
        occ.occupancy_grid.copy_(torch.randn(occ.occupancy_grid.shape, generator=gen) * 2.0)
    adam = torch.optim.Adam([{"params": [sig.params], "lr": 1e-2}])
    sgd = torch.optim.SGD([occ.occupancy_grid], lr=1e-4)
    ckpt = {"global_step": 0, "network_state_dict": model.state_dict(), "optimizer_state_dict": adam.state_dict(), "poses": [],
            "occ_model_state_dict": occ.state_dict(), "occ_optimizer_state_dict": sgd.state_dict()}
    torch.save(ckpt, path)
    return dict(sigma_params=sig.params.detach().clone(), occ_grid=occ.occupancy_grid.detach().clone())


# ---------------------------------------------------------------------------------------------------------------------------
# Reference-shaped callers (boundary test).  In an integration the Optimizer is handed the REFERENCE's own
# KeyFrame / Frame / Pose / LidarScan / Settings objects.  The stand-ins below expose exactly the members
# those classes expose - written from the member list of SURVEY 8b and the class summaries in INTEGRATION.md, with the behaviours the
# hot path depends on (CPU tensors; list -> tuple on attribute access of the settings; a Pose whose matrix is cached at construction
# and recomputed from the 6-vector only while that vector requires a gradient) - and REFUSE every other attribute: setting a member
# the reference class does not have raises, reading one raises AttributeError as usual.  If loner_amd touches anything beyond the
# reference's surface, the boundary test fails.
# ---------------------------------------------------------------------------------------------------------------------------
class _Surface:
    _MEMBERS = ()

    def __setattr__(self, name, value):
        assert name in self._MEMBERS, f"{type(self).__name__} has no member '{name}' in the reference: loner_amd must not set it"
        object.__setattr__(self, name, value)


def _aa_to_matrix(p6):
    """[t, axis-angle] -> 4x4 (the quaternion route of pytorch3d 0.7.2), differentiable"""
    aa = p6[3:6]
    theta = torch.linalg.norm(aa)
    half = theta * 0.5
    k = torch.where(theta.abs() < 1e-6, 0.5 - theta * theta / 48.0, torch.sin(half) / torch.where(theta.abs() < 1e-6, torch.ones_like(theta), theta))
    q = torch.cat([torch.cos(half)[None], aa * k])
    r, i, j, kk = q[0], q[1], q[2], q[3]
    two_s = 2.0 / (q * q).sum()
    R = torch.stack([1 - two_s * (j * j + kk * kk), two_s * (i * j - kk * r), two_s * (i * kk + j * r),
                     two_s * (i * j + kk * r), 1 - two_s * (i * i + kk * kk), two_s * (j * kk - i * r),
                     two_s * (i * kk - j * r), two_s * (j * kk + i * r), 1 - two_s * (i * i + j * j)]).reshape(3, 3)
    T = torch.eye(4, dtype=p6.dtype)
    T = T.clone()
    T[:3, :3] = R
    T[:3, 3] = p6[0:3]
    return T


class RefShapedPose(_Surface):
    """src/common/pose.py:23-166 by member name.  The quirk that matters at the boundary: get_transformation_matrix() returns the
    matrix CACHED AT CONSTRUCTION unless the 6-vector currently requires a gradient (pose.py:140-144)."""
    _MEMBERS = ("_pose_tensor", "_transformation_matrix")

    def __init__(self, transformation_matrix=None, pose_tensor=None, fixed=None, requires_tensor=False):
        if transformation_matrix is None and pose_tensor is None:
            transformation_matrix = torch.eye(4)
        if fixed is None:
            fixed = not (pose_tensor if transformation_matrix is None else transformation_matrix).requires_grad
        if pose_tensor is not None:
            self._pose_tensor = pose_tensor
            self._pose_tensor.requires_grad_(not fixed)
            transformation_matrix = _aa_to_matrix(self._pose_tensor).float()
        else:
            assert not requires_tensor, "not needed by the boundary test"
            self._pose_tensor = None
            transformation_matrix = transformation_matrix.float()
        self._transformation_matrix = transformation_matrix.detach()
        self._transformation_matrix.requires_grad_(not fixed)

    def set_fixed(self, fixed=True):
        self._pose_tensor.requires_grad_(not fixed)

    def to(self, device):
        if self._pose_tensor is not None:
            self._pose_tensor = self._pose_tensor.to(device)
        self._transformation_matrix = self._transformation_matrix.to(device)
        return self

    def detach(self):
        return RefShapedPose(self.get_transformation_matrix().detach())

    def clone(self, fixed=None, requires_tensor=False):
        m = self.get_transformation_matrix()
        return RefShapedPose(m.detach().clone(), fixed=(not m.requires_grad) if fixed is None else fixed)

    def get_transformation_matrix(self):
        if self._pose_tensor is None or not self._pose_tensor.requires_grad:
            return self._transformation_matrix
        return _aa_to_matrix(self._pose_tensor)

    def get_pose_tensor(self):
        assert self._pose_tensor is not None, "the boundary test builds its poses from 6-vectors"
        return self._pose_tensor

    def get_translation(self):
        return self._pose_tensor[:3] if self._pose_tensor is not None else self._transformation_matrix[:3, 3]

    def get_rotation(self):
        return self.get_transformation_matrix()[:3, :3]

    def get_axis_angle(self):
        return self._pose_tensor[3:]


class RefShapedLidarScan(_Surface):
    """src/common/sensors.py:57-167 by member name: SoA CPU tensors, `len()` = number of timestamps."""
    _MEMBERS = ("ray_directions", "distances", "timestamps", "sky_rays")

    def __init__(self, ray_directions, distances, timestamps, sky_rays=None):
        self.ray_directions, self.distances, self.timestamps, self.sky_rays = ray_directions, distances, timestamps, sky_rays

    def __len__(self):
        return self.timestamps.shape[0]

    def get_start_time(self):
        return self.timestamps[0]

    def get_end_time(self):
        return self.timestamps[-1]

    def to(self, device):
        self.ray_directions, self.distances, self.timestamps = (t.to(device) for t in (self.ray_directions, self.distances, self.timestamps))
        return self


class RefShapedFrame(_Surface):
    """src/common/frame.py:22-156 by member name."""
    _MEMBERS = ("image", "lidar_points", "_lidar_to_camera", "_lidar_pose", "_gt_lidar_pose", "_id")

    def __init__(self, image=None, lidar_points=None, T_lidar_to_camera=None):
        self.image, self.lidar_points, self._lidar_to_camera = image, lidar_points, T_lidar_to_camera
        self._lidar_pose, self._gt_lidar_pose, self._id = None, None, -1

    def to(self, device):
        self.lidar_points.to(device)
        for p in (self._lidar_to_camera, self._lidar_pose, self._gt_lidar_pose):
            if p is not None:
                p.to(device)
        return self

    def get_time(self):
        return self.lidar_points.get_start_time()

    def get_lidar_pose(self):
        return self._lidar_pose


class RefShapedKeyFrame(_Surface):
    """src/mapping/keyframe.py:24-69,126-135 by member name (build_lidar_rays / build_camera_rays are the reference's CPU ray
    builders: the optimiser under test must not need them)."""
    _MEMBERS = ("_frame", "_device", "_tracked_lidar_pose", "is_anchored", "lidar_loss_distribution")

    def __init__(self, frame, device=None):
        self._frame = frame.to(device) if device is not None else frame
        self._device = device
        self._tracked_lidar_pose = frame.get_lidar_pose().clone()
        self.is_anchored = False
        self.lidar_loss_distribution = None

    def get_lidar_pose(self):
        return self._frame.get_lidar_pose()

    def get_lidar_scan(self):
        return self._frame.lidar_points

    def get_time(self):
        return self._frame.get_time()

    def build_lidar_rays(self, *a, **k):
        raise AssertionError("the HIP optimiser builds its rays on the device from the scan buffers")

    def get_pose_state(self):
        return {"timestamp": self.get_time().detach().cpu().clone(),
                "lidar_pose": self._frame.get_lidar_pose().get_pose_tensor().detach().cpu().clone(),
                "gt_lidar_pose": self._frame._gt_lidar_pose.get_pose_tensor().detach().cpu().clone(),
                "tracked_pose": self._tracked_lidar_pose.get_transformation_matrix().detach().cpu().clone()}


class RefShapedSettings(dict):
    """attrdict.AttrDict as the reference's Settings uses it (src/common/settings.py:51-75): attribute access, nested dicts wrapped on
    access, LISTS BECOME TUPLES on attribute access (and stay lists on item access); unknown keys raise."""

    def __getattr__(self, key):
        if key not in self:
            raise AttributeError(key)
        return self._build(self[key])

    def __setattr__(self, key, value):
        self[key] = value

    @classmethod
    def _build(cls, obj):
        if isinstance(obj, dict):
            return cls(obj)
        if isinstance(obj, (list, tuple)):
            return tuple(cls._build(v) for v in obj)
        return obj


class RefShapedWorldCube(_Surface):
    """src/common/pose_utils.py:24-57 by member name."""
    _MEMBERS = ("scale_factor", "shift")

    def __init__(self, scale_factor, shift):
        self.scale_factor, self.shift = scale_factor, shift

    def to(self, device, clone=False):
        if clone:
            return RefShapedWorldCube(self.scale_factor.to(device), self.shift.to(device))
        self.scale_factor, self.shift = self.scale_factor.to(device), self.shift.to(device)
        return self
