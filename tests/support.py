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
