"""Boundary classes (Model, samplers, KeyFrame, Optimizer) on the MI355X vs oracle / golden fixtures."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mapping_step as MS
from oracle import network as NW
from oracle import poses as OP
from oracle import render as ORD

DEV = "cuda"


def rel(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


from tests.support import SMALL_ENC, SMALL_NET, small_settings       # noqa: E402,F401


def make_keyframes(pose6_list, device=None):
    from loner_amd.common.frame import Frame
    from loner_amd.common.pose import Pose
    from loner_amd.common.sensors import LidarScan
    from loner_amd.mapping.keyframe import KeyFrame
    from loner_amd.utils import synthetic as SY
    dirs, ts = SY.lidar_pattern()
    base = SY.trajectory_pose6(8)
    kfs = []
    for i, p6 in enumerate(pose6_list):
        dist = SY.scene_ranges(dirs, OP.transform_from_pose6(base[i]))
        fr = Frame(None, LidarScan(dirs.clone(), dist, ts + float(i), sky_rays=torch.Tensor()), Pose())
        fr._lidar_pose = Pose(pose_tensor=p6.clone(), fixed=False)
        fr._gt_lidar_pose = Pose(pose_tensor=base[i].clone(), fixed=True)
        kfs.append(KeyFrame(fr, device))
    return kfs


def world_cube():
    from loner_amd.common.pose_utils import WorldCube
    from loner_amd.utils import synthetic as SY
    scale, shift = SY.world_cube()
    return WorldCube(torch.tensor(scale), torch.from_numpy(shift))


def test_keyframe_build_lidar_rays_api_and_pose_gradient(golden):
    g = golden("g1_rays")
    from loner_amd.common.pose_utils import WorldCube, tensor_to_transform
    from loner_amd.common.ray_utils import LidarRayDirections
    from loner_amd.common.sensors import LidarScan
    wc = WorldCube(torch.tensor(float(g["scale"])), torch.from_numpy(g["shift"]))
    for i in range(3):
        scan = LidarScan(torch.from_numpy(g[f"dirs_g{i}"]), torch.from_numpy(g[f"dist_g{i}"]), torch.zeros(g[f"dist_g{i}"].shape[0]))
        p6 = torch.from_numpy(g[f"pose{i}"]).clone().requires_grad_(True)       # pose on the CPU, like the reference
        rays, depths = LidarRayDirections(scan).build_lidar_rays(torch.arange(len(scan)), torch.from_numpy(g["ray_range"]), wc,
                                                                tensor_to_transform(p6))
        assert rays.is_cuda and rays.shape == g[f"rays{i}"].shape
        assert rel(rays, g[f"rays{i}"]) < 1e-6
        (rays * torch.from_numpy(g[f"cot{i}"]).to(DEV)).sum().backward()
        assert rel(p6.grad, g[f"dpose{i}"]) < 5e-4


def test_model_forward_api_and_autograd_match_oracle(golden):
    from loner_amd.common.settings import Settings, default_model_config
    from loner_amd.models.model_tcnn import Model, OccupancyGridModel
    from loner_amd.models.ray_sampling import OccGridRaySampler
    g = golden("g4_samplers")
    mc = default_model_config()
    mc["model"]["nerf_config"]["pos_encoding_sigma"] = dict(SMALL_ENC)
    mc["model"]["nerf_config"]["sigma_network"] = dict(SMALL_NET)
    mc["model"]["nerf_config"]["pos_encoding_intensity"]["log2_hashmap_size"] = 10
    mc["model"]["render"].update(N_samples_train=128, raw_noise_std=0.0, perturb=1.0, chunk=48)   # 64 rays -> 2 chunks
    cfg = Settings(mc).model
    model = Model(cfg).to(DEV)
    spec_o = NW.NetworkSpec.from_config(SMALL_ENC, SMALL_NET)
    params = NW.init_params(spec_o, 5)
    params[spec_o.n_mlp_params:] *= 3000
    with torch.no_grad():
        model.nerf_model._model_sigma.params.copy_(params)
    assert set(model.state_dict()) >= {"nerf_model._model_sigma.params", "nerf_model._pos_encoding.params",
                                       "nerf_model._model_intensity.params"}
    occ = OccupancyGridModel(Settings(dict(voxel_size=24))).to(DEV)
    with torch.no_grad():
        occ.occupancy_grid.copy_(torch.from_numpy(g["trained_grid"])[None, None])
    sampler = OccGridRaySampler()
    sampler.update_occ_grid(occ().detach())
    rays = torch.from_numpy(g["rays"]).to(DEV).requires_grad_(True)
    torch.manual_seed(0)
    res = model(rays, sampler, torch.tensor(85.76), camera=False, return_variance=True)
    assert set(res) >= {"rgb_fine", "depth_fine", "weights_fine", "opacity_fine", "variance", "samples_fine", "points_fine"}
    z = res["samples_fine"]
    assert z.shape == (64, 128) and bool((z[:, 1:] >= z[:, :-1]).all())
    # oracle on the same sample depths
    p_o = params.clone().requires_grad_(True)
    rays_o = torch.from_numpy(g["rays"]).clone().requires_grad_(True)
    zc = z.detach().cpu()
    sig = NW.density(spec_o, p_o, ORD.sample_points(rays_o, zc).reshape(-1, 3)).reshape(64, 128)
    out = ORD.composite(sig, zc, rays_o[:, 3:6], rays_o[:, 12:13])
    assert rel(res["depth_fine"], out["depth"]) < 1e-4                      # north_star tolerance
    assert rel(res["weights_fine"], out["weights"]) < 1e-4
    assert rel(res["opacity_fine"], out["opacity"]) < 1e-4 and rel(res["variance"], out["variance"]) < 1e-3
    assert rel(res["points_fine"], ORD.sample_points(rays_o, zc)) < 1e-6
    gen = torch.Generator().manual_seed(1)
    cd, cw, co = torch.randn(64, generator=gen), torch.randn(64, 128, generator=gen), torch.randn(64, generator=gen)
    (res["depth_fine"] * cd.to(DEV)).sum().add((res["weights_fine"] * cw.to(DEV)).sum()).add((res["opacity_fine"] * co.to(DEV)).sum()).backward()
    ((out["depth"] * cd).sum() + (out["weights"] * cw).sum() + (out["opacity"] * co).sum()).backward()
    assert rel(model.nerf_model._model_sigma.params.grad, p_o.grad) < 2e-4
    assert rel(rays.grad, rays_o.grad) < 2e-4
    # inference entry used by the analysis scripts (testing=True -> N_samples_test, no jitter)
    with torch.no_grad():
        res_t = model(rays.detach(), sampler, torch.tensor(85.76), testing=True, camera=False, return_variance=True)
    assert res_t["samples_fine"].shape == (64, 2048) and torch.isfinite(res_t["depth_fine"]).all()
    pts = torch.rand(3, 5, 3, device=DEV) * 1.8 - 0.9
    sig_pts = model.inference_points(pts, None, sigma_only=True)
    assert rel(sig_pts[:, 0], NW.density(spec_o, params, pts.cpu().reshape(-1, 3))) < 1e-5


class _Replay:
    def __init__(self, g):
        keys = sorted(k for k in g if k.startswith("draw"))
        self.draws = [torch.from_numpy(g[k]) for k in keys]
        self.kinds = [k.split("_")[1] for k in keys]
        self.i = 0

    def _next(self, kind):
        assert self.kinds[self.i] == kind, (self.i, self.kinds[self.i], kind)
        v = self.draws[self.i]
        self.i += 1
        return v

    def ray_index(self, n, c): return self._next("randint")
    def sky_index(self, n, c): return self._next("randint")
    def jitter(self, n, h): return self._next("rand")
    def pdf(self, n, h): return self._next("rand")
    def noise(self, n, s): return self._next("randn")


def test_optimizer_loop_reproduces_reference_trajectory(golden):
    """_do_iterate_optimizer for 12 iterations / 2 keyframes with the reference's recorded random draws:
    final pose, occupancy grid (2 grid steps) and parameters against the reference's (G9)."""
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    g = golden("g9_loop")
    opt = Optimizer(small_settings(48, 64), None, world_cube(), 0, False, True, False)
    with torch.no_grad():
        opt._model.nerf_model._model_sigma.params.copy_(torch.from_numpy(g["params0"]))
    kfs = make_keyframes([torch.from_numpy(g["pose_init0"]), torch.from_numpy(g["pose_init1"])])
    kfs[0].is_anchored = True
    rp = _Replay(g)
    opt.set_draws(rp)
    opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(12, False, False, False, True))
    assert rp.i == int(g["n_draws"])                                   # same number and order of draws (A.9)
    assert opt._global_step == int(g["global_step"])
    assert np.abs(kfs[0].get_lidar_pose().get_pose_tensor().detach().numpy() - g["pose_final0"]).max() == 0
    pose1 = kfs[1].get_lidar_pose().get_pose_tensor().detach().cpu().numpy()
    print("pose error vs reference", np.abs(pose1 - g["pose_final1"]).max(), "pose travel", np.abs(g["pose_final1"] - g["pose_init1"]).max())
    assert np.abs(pose1 - g["pose_final1"]).max() < 5e-4
    assert rel(opt._occupancy_grid_model.occupancy_grid[0, 0], g["grid1"]) < 2e-3
    assert rel(opt._model.nerf_model._model_sigma.params, g["params1"]) < 5e-2
    sd = opt._optimizer.state_dict()
    assert sd["state"][0]["step"] == 12 and set(sd["state"][0]) >= {"exp_avg", "exp_avg_sq"}


@pytest.mark.parametrize("k", [1, 2, 3])
def test_optimizer_first_iterations_match_reference_state_by_state(golden, k):
    """G14: the state after k = 1, 2, 3 iterations of the reference's loop on G9's configuration and draws - density parameters,
    both Adam moments of the density group and of the pose group, the free pose, the grid - asserted TIGHTLY (the 12-iteration
    end state above is Adam-amplified and loose: a wrong bias correction or moment update would move every element by ~lr here).
    The parameters are a quantile statement: a handful of table entries whose gradient is of the size of Adam's eps (1e-8) take a
    different-sized step on any rounding difference (the CPU oracle shows the same: tests/test_oracle_golden.py)."""
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    g, h = golden("g9_loop"), golden("g14_loop_first_steps")
    opt = Optimizer(small_settings(48, 64), None, world_cube(), 0, False, True, False)
    with torch.no_grad():
        opt._model.nerf_model._model_sigma.params.copy_(torch.from_numpy(g["params0"]))
    kfs = make_keyframes([torch.from_numpy(g["pose_init0"]), torch.from_numpy(g["pose_init1"])])
    kfs[0].is_anchored = True
    opt.set_draws(_Replay(g))
    opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(k, False, False, False, True))
    sd = opt._optimizer.state_dict()
    st_sigma, st_pose = sd["state"][sd["param_groups"][0]["params"][0]], sd["state"][sd["param_groups"][1]["params"][0]]
    assert int(st_sigma["step"]) == int(h[f"step_{k}"]) == k
    e = dict(m=rel(st_sigma["exp_avg"], h[f"exp_avg_{k}"]), v=rel(st_sigma["exp_avg_sq"], h[f"exp_avg_sq_{k}"]),
             pm=rel(st_pose["exp_avg"][1], h[f"pose_exp_avg_{k}"]), pv=rel(st_pose["exp_avg_sq"][1], h[f"pose_exp_avg_sq_{k}"]))
    dp = np.abs(opt._model.nerf_model._model_sigma.params.detach().cpu().numpy() - h[f"params_{k}"])
    pose1 = kfs[1].get_lidar_pose().get_pose_tensor().detach().cpu().numpy()
    e.update(p99=float(np.quantile(dp, 0.99)), frac_1e4=float((dp > 1e-4).mean()), pmax=float(dp.max()),
             pose=float(np.abs(pose1 - h[f"pose1_{k}"]).max()), grid=rel(opt._occupancy_grid_model.occupancy_grid[0, 0], h[f"grid_{k}"]))
    print(f"G14 k={k}:", {a: f"{b:.2e}" for a, b in e.items()})
    assert e["m"] < 1e-4 and e["v"] < 2e-4, e                           # Adam moments of the density parameters
    assert e["pm"] < 5e-4 and e["pv"] < 1e-3, e                         # ... of the free pose (its gradient is asserted to 5e-4 elsewhere)
    assert e["p99"] < 1e-5 and e["frac_1e4"] < 2e-3 and e["pmax"] <= 2.001 * 1e-2 * k, e
    assert e["pose"] < 2e-6 and e["grid"] < 1e-5, e


def test_optimizer_schedule_default_config_properties():
    """Default network (16-level hash grid, 64-wide MLP), 2 keyframes, in-kernel RNG: the schedule runs, the
    anchored keyframe does not move, the free one does, the loss goes down, nothing is NaN."""
    from loner_amd.common.settings import default_optimizer_settings
    from loner_amd.mapping.optimizer import Optimizer
    from loner_amd.utils import synthetic as SY
    s = default_optimizer_settings()
    s["num_samples"]["sky"] = 0
    s["keyframe_schedule"][0]["iteration_schedule"][0]["num_iterations"] = 60
    s["keyframe_schedule"][1]["iteration_schedule"][1]["num_iterations"] = 20
    torch.manual_seed(0)
    opt = Optimizer(s, None, world_cube(), 0, False, True, False)
    base = SY.trajectory_pose6(8)
    noisy = base[1].clone(); noisy[:3] += 0.02
    kfs = make_keyframes([base[0], noisy])
    opt.iterate_optimizer(kfs[:1])
    assert opt._keyframe_count == 1 and opt._global_step == 60 and kfs[0].is_anchored
    l0 = opt.last_stats["loss_terms"][:, 0]
    assert torch.isfinite(l0).all() and float(l0[-10:].mean()) < float(l0[:10].mean())
    assert opt.last_stats["n_valid_rays"] == 60 * 512
    p_before = kfs[1].get_lidar_pose().get_pose_tensor().detach().clone()
    opt.iterate_optimizer(kfs)
    assert opt._global_step == 80
    assert torch.equal(kfs[0].get_lidar_pose().get_pose_tensor().detach(), base[0])
    assert not torch.equal(kfs[1].get_lidar_pose().get_pose_tensor().detach(), p_before)
    assert float(opt._occupancy_grid_model.occupancy_grid.abs().max()) > 0
    with open(f"{s['log_directory']}/timing.csv") as f:
        assert len(f.read().strip().splitlines()) >= 2


def test_compute_loss_is_differentiable_like_the_reference(golden):
    """Optimizer.compute_loss returns a scalar whose .backward() reaches the density parameters and CPU poses."""
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    opt = Optimizer(small_settings(64, 64), None, world_cube(), 0, False, True, False)
    from loner_amd.utils import synthetic as SY
    base = SY.trajectory_pose6(8)
    kfs = make_keyframes([base[0], base[1]])
    for kf in kfs:
        kf.get_lidar_pose().set_fixed(False)
    opt._optimization_settings = OptimizationSettings(1, False, False, False, True)
    rays, depths = [], []
    for kf in kfs:
        r, d = kf.build_lidar_rays(torch.randint(65536, (64,)), opt._ray_range, opt._world_cube)
        rays.append(r); depths.append(d)
    loss = opt.compute_loss(None, (torch.vstack(rays), torch.cat(depths)), 0)
    assert loss.dim() == 0 and torch.isfinite(loss)
    loss.backward()
    assert opt._model.nerf_model._model_sigma.params.grad.abs().sum() > 0
    for kf in kfs:
        gp = kf.get_lidar_pose().get_pose_tensor().grad
        assert gp is not None and torch.isfinite(gp).all() and gp.abs().sum() > 0
    opt._step_occupancy_grid()
    assert float(opt._occupancy_grid_model.occupancy_grid.abs().max()) > 0


def test_api_parity_mode_reproduces_the_reference_and_the_fused_loss(golden):
    """Optimizer.compute_loss_api - Model.forward -> result dictionary -> torch ops -> autograd, the route of the reference's own
    compute_loss (optimizer.py:437-595; SURVEY 8d "API-parity mode") - on the G8 fixture: the loss, the dynamic margin and the
    gradients with respect to the density parameters and the ray records equal the REFERENCE's, and they equal what the fused-loss
    route (compute_loss: one kernel pass, no dictionary) returns for the same draws."""
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    g = golden("g8_compute_loss")
    opt = Optimizer(small_settings(96, 128), None, world_cube(), 0, False, True, False)
    sig = opt._model.nerf_model._model_sigma.params
    with torch.no_grad():
        sig.copy_(torch.from_numpy(g["params"]))
        opt._occupancy_grid_model.occupancy_grid.copy_(torch.from_numpy(g["grid"])[None, None])
    opt._occupancy_grid = opt._occupancy_grid_model()
    opt._ray_sampler.update_occ_grid(opt._occupancy_grid.detach())
    opt._optimization_settings = OptimizationSettings(1, False, False, False, True)
    opt._model.freeze_sigma_head(False)

    class Draws:
        def jitter(self, n, h): return torch.from_numpy(g["u1"])
        def pdf(self, n, h): return torch.from_numpy(g["u2"])
        def noise(self, n, s_): return torch.from_numpy(g["noise"])
    depths = torch.from_numpy(g["depths"]).to(DEV)
    out = {}
    for mode in ("api", "fused"):
        rays = torch.from_numpy(g["rays"]).to(DEV).requires_grad_(True)
        sig.grad = None
        if mode == "api":
            opt._ray_sampler.set_draws(Draws())
            loss = opt.compute_loss_api((rays, depths), 0)
            opt._ray_sampler.set_draws(None)
            res = opt._results_lidar
            assert res["weights_fine"].shape == (192, 128) and res["points_fine"].shape == (192, 128, 3)
            # sample depths: bit-identical to the reference's on the rays where its float32 exp rounded correctly (the grid of this
            # fixture is random logits: SURVEY B.5), within an ulp-sized shift elsewhere
            zk = res["samples_fine"].detach().cpu().numpy()
            same_rows = float((zk == g["z"]).all(axis=1).mean())
            print(f"API mode: rays with bit-identical sample depths {same_rows:.3f}, max |dz| {np.abs(zk - g['z']).max():.2e}")
            assert same_rows > 0.5 and np.abs(zk - g["z"]).max() < 2e-3 and np.median(np.abs(zk - g["z"])) == 0
            assert rel(res["weights_fine"], g["weights"]) < 1e-4 and rel(res["depth_fine"], g["depth"]) < 1e-4
            grid_before = opt._occupancy_grid_model.occupancy_grid.detach().clone()
            opt._step_occupancy_grid()       # the occupancy step reads the API mode's result dictionary too (optimizer.py:598-609)
            assert not torch.equal(grid_before, opt._occupancy_grid_model.occupancy_grid.detach())
            with torch.no_grad():
                opt._occupancy_grid_model.occupancy_grid.copy_(grid_before)
            opt._ray_sampler.update_occ_grid(opt._occupancy_grid_model().detach())
        else:
            opt.set_draws(Draws())
            loss = opt.compute_loss(None, (rays, depths), 0)
            opt.set_draws(None)
        loss.backward()
        out[mode] = (float(loss), sig.grad.clone(), rays.grad.clone(), float(opt._depth_eps))
        assert abs(float(loss) - float(g["loss"])) / float(g["loss"]) < 1e-4, mode
        assert rel(sig.grad, g["dparams"]) < 2e-4 and rel(rays.grad, g["drays"]) < 2e-4, mode
        assert abs(float(opt._depth_eps) - float(g["depth_eps"])) < 1e-4 * float(g["depth_eps"]), mode
    assert abs(out["api"][0] - out["fused"][0]) < 2e-5 * abs(out["fused"][0])
    assert rel(out["api"][1], out["fused"][1]) < 1e-4 and rel(out["api"][2], out["fused"][2]) < 1e-4


def test_training_reduces_l1_depth_matched_quality_gate():
    """The quality half of the metric ("training rays/s at matched L1 depth"): map-only optimisation of one synthetic
    keyframe must pull the rendered depth (Model.forward(testing=True), S=2048) towards the analytic ranges."""
    from loner_amd.analysis.l1_depth import compute_l1_depth
    from loner_amd.common.ray_utils import LidarRayDirections
    from loner_amd.common.settings import default_optimizer_settings
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    from loner_amd.utils import synthetic as SY
    s = default_optimizer_settings()
    s["num_samples"]["sky"] = 0
    torch.manual_seed(0)
    wc = world_cube()
    opt = Optimizer(s, None, wc, 0, False, True, False)
    kf = make_keyframes([SY.trajectory_pose6(1)[0]])[0]
    rr = torch.tensor([1.0, 50.0])
    lrd = LidarRayDirections(kf.get_lidar_scan(), chunk_size=2048)
    args = (kf.get_lidar_pose(), lrd, opt._model, opt._ray_sampler, wc, rr, DEV)
    before = compute_l1_depth(*args, max_rays=4096)
    opt._do_iterate_optimizer([kf], [None], optimizer_settings=OptimizationSettings(300, True, False, False, True))
    after = compute_l1_depth(*args, max_rays=4096)
    print(f"L1 depth before {before:.3f} m, after 300 iterations {after:.3f} m")
    assert after < 0.5 * before and after < 2.0


def test_checkpoint_round_trip_as_the_mapper_writes_it(tmp_path):
    """On-disk contract either side of the path: the reference's Mapper.build_ckpt (mapper.py:161-175)
    reads _model / _optimizer / _occupancy_grid_model / _occupancy_grid_optimizer / _global_step off the Optimizer and
    torch.save()s them; its analysis scripts (compute_l1_depth.py:140-155, renderer_lidar.py:170-182) rebuild Model and
    OccupancyGridModel from the config and load_state_dict() the two model entries.  Same steps here, then the restored
    map must render exactly what the trained one renders."""
    from loner_amd.analysis.l1_depth import compute_l1_depth
    from loner_amd.common.ray_utils import LidarRayDirections
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    from loner_amd.models.model_tcnn import Model, OccupancyGridModel
    from loner_amd.models.ray_sampling import OccGridRaySampler
    from loner_amd.utils import synthetic as SY
    s = small_settings(128, 64)
    torch.manual_seed(0)
    wc = world_cube()
    opt = Optimizer(s, None, wc, 0, False, True, False)
    kfs = make_keyframes(list(SY.trajectory_pose6(2)))
    opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(25, False, False, False, True))
    ckpt = {"global_step": opt._global_step,
            "network_state_dict": opt._model.state_dict(),
            "optimizer_state_dict": opt._optimizer.state_dict(),
            "poses": [kf.get_pose_state() for kf in kfs],
            "occ_model_state_dict": opt._occupancy_grid_model.state_dict(),
            "occ_optimizer_state_dict": opt._occupancy_grid_optimizer.state_dict()}
    path = tmp_path / "final.tar"
    torch.save(ckpt, str(path))
    back = torch.load(str(path), map_location="cpu", weights_only=False)
    assert back["global_step"] == 25 and len(back["poses"]) == 2
    assert set(back["poses"][0]) == {"timestamp", "lidar_to_camera", "lidar_pose", "gt_lidar_pose", "tracked_pose"}
    assert set(back["network_state_dict"]) >= {"nerf_model._model_sigma.params", "nerf_model._pos_encoding.params",
                                               "nerf_model._dir_encoding.params", "nerf_model._model_intensity.params"}
    assert set(back["occ_model_state_dict"]) == {"occupancy_grid"}
    st = back["optimizer_state_dict"]
    assert set(st) == {"state", "param_groups"} and st["state"][0]["exp_avg"].shape == opt._model.nerf_model._model_sigma.params.shape
    # the consumer side, as compute_l1_depth.py does it
    mc = opt._model_config.model
    model = Model(mc).to(DEV)
    occ = OccupancyGridModel(mc.occ_model).to(DEV)
    model.load_state_dict(back["network_state_dict"])
    occ.load_state_dict(back["occ_model_state_dict"])
    sampler = OccGridRaySampler()
    sampler.update_occ_grid(occ().detach())
    rr = torch.tensor([1.0, 50.0])
    lrd = LidarRayDirections(kfs[0].get_lidar_scan(), chunk_size=1024)
    torch.manual_seed(123)            # the importance samples are random even at test time (ray_sampling.py:86, det=False)
    a = compute_l1_depth(kfs[0].get_lidar_pose(), lrd, opt._model, opt._ray_sampler, wc, rr, DEV, max_rays=1024)
    torch.manual_seed(123)
    b = compute_l1_depth(kfs[0].get_lidar_pose(), lrd, model, sampler, wc, rr, DEV, max_rays=1024)
    assert a == b and np.isfinite(a)


def test_training_run_is_bit_reproducible():
    """Same seeds -> same bits: every accumulation that crosses waves (table gradient, weight gradient, ray gradient, occupancy
    pseudo-gradient, loss terms) is a fixed-order or a 64-bit fixed-point sum, so two runs of the joint map + pose optimisation
    (incl. two occupancy steps) end with identical parameters, Adam state, poses and occupancy grid.  The reference cannot
    offer this (tinycudann accumulates with fp16 atomics)."""
    from loner_amd.common.settings import default_optimizer_settings
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    from loner_amd.utils import synthetic as SY

    def run():
        s = default_optimizer_settings()
        s["num_samples"]["sky"] = 0
        s["num_samples"]["lidar"] = 256
        s["model_config"]["model"]["render"]["N_samples_train"] = 128
        torch.manual_seed(0)
        opt = Optimizer(s, None, world_cube(), 0, False, True, False)
        base = SY.trajectory_pose6(3)
        kfs = make_keyframes([base[0], base[1] + torch.tensor([0.02, -0.01, 0.0, 0.0, 0.0, 0.0]), base[2]])
        kfs[0].is_anchored = True
        torch.manual_seed(7)
        opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(25, False, False, False, True))
        p = opt._model.nerf_model._model_sigma.params
        st = opt._optimizer.state[p]
        return (p.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), opt._occupancy_grid_model.occupancy_grid.detach().clone(),
                torch.stack([kf.get_lidar_pose().get_pose_tensor().detach().cpu() for kf in kfs]), opt.last_stats["loss_terms"].clone())

    a, b = run(), run()
    names = ["params", "exp_avg", "exp_avg_sq", "occupancy grid", "poses", "loss terms"]
    for n, x, y in zip(names, a, b):
        assert torch.equal(x, y), f"{n} differ between two identically seeded runs ({int((x != y).sum())} entries)"
    assert float((a[3] != 0).sum()) > 0 and float((a[4][1] - a[4][0]).abs().max()) > 0


def test_sky_rays_tracking_phase_and_uniform_sampler():
    """The schedule variants around the default path: sky rays (keyframe.py:91-100), the pose-refinement phase
    (latest_kf_only + frozen density net, optimizer.py:239-259) and the UNIFORM sampler / FIXED ray selection."""
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    from loner_amd.utils import synthetic as SY
    base = SY.trajectory_pose6(8)
    s = small_settings(64, 64)
    s["num_samples"]["sky"] = 16
    torch.manual_seed(0)
    opt = Optimizer(s, None, world_cube(), 0, False, True, True)          # sky segmentation enabled
    noisy = base[2].clone(); noisy[0] += 0.05
    kfs = make_keyframes([base[0], base[1], noisy])
    up = torch.nn.functional.normalize(torch.tensor([[0.0, 0.1, -0.1, 0.3], [0.0, 0.2, 0.1, -0.2], [1.0, 1.0, 1.0, 1.0]]), dim=0)
    for kf in kfs:
        kf.get_lidar_scan().sky_rays = up.clone()
    kfs[0].is_anchored = True
    opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(8, False, False, False, True))
    assert opt.last_stats["n_valid_rays"] == 8 * 3 * (64 + 16)            # lidar + sky rays of every keyframe, none dropped
    assert torch.isfinite(opt.last_stats["loss_terms"]).all()
    # tracking phase: only the most recent keyframe's pose moves, the density parameters stay put
    p_before = opt._model.nerf_model._model_sigma.params.detach().clone()
    poses_before = [kf.get_lidar_pose().get_pose_tensor().detach().clone() for kf in kfs]
    opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(6, False, True, True, True))
    assert torch.equal(opt._model.nerf_model._model_sigma.params.detach(), p_before)
    assert torch.equal(kfs[0].get_lidar_pose().get_pose_tensor().detach(), poses_before[0])
    assert torch.equal(kfs[1].get_lidar_pose().get_pose_tensor().detach(), poses_before[1])
    assert not torch.equal(kfs[2].get_lidar_pose().get_pose_tensor().detach(), poses_before[2])
    assert opt.last_stats["n_valid_rays"] == 6 * (64 + 16)
    # UNIFORM sampler + FIXED ray selection + L2_LOS loss
    s2 = small_settings(64, 64)
    s2["samples_selection"]["strategy"] = "UNIFORM"
    s2["rays_selection"]["strategy"] = "FIXED"
    s2["model_config"]["loss"]["loss_selection"] = "L2_LOS"
    opt2 = Optimizer(s2, None, world_cube(), 0, False, True, False)
    kf = make_keyframes([base[0]])
    opt2._do_iterate_optimizer(kf, [None], optimizer_settings=OptimizationSettings(5, True, False, False, True))
    assert torch.isfinite(opt2.last_stats["loss_terms"]).all() and opt2.last_stats["n_valid_rays"] == 5 * 64
    with pytest.raises(RuntimeError):
        s3 = small_settings(8, 64); s3["rays_selection"]["strategy"] = "NOPE"
        Optimizer(s3, None, world_cube(), 0, False, True, False)._do_iterate_optimizer(
            make_keyframes([base[0]]), [None], optimizer_settings=OptimizationSettings(1, True, False, False, True))


def test_sky_rays_and_tracking_phase_reproduce_the_reference(golden):
    """G11: the reference's Optimizer on three keyframes with sky rays (keyframe.py:91-100) - a joint map + pose phase, then the
    pose-refinement phase of the default schedule (latest_kf_only, frozen density net: optimizer.py:239-259) - replayed on the
    HIP path with the reference's recorded draws: loss of every iteration, poses, parameters and occupancy grid."""
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    from tests import support
    g = golden("g11_sky_tracking")
    s = small_settings(48, 64)
    s["num_samples"]["sky"] = 16
    opt = Optimizer(s, None, world_cube(), 0, False, True, True)
    with torch.no_grad():
        opt._model.nerf_model._model_sigma.params.copy_(torch.from_numpy(g["params0"]))
    kfs = make_keyframes([torch.from_numpy(g[f"pose_init{i}"]) for i in range(3)])
    for kf in kfs:
        kf.get_lidar_scan().sky_rays = torch.from_numpy(g["sky"]).clone()
    assert [float(kf.get_time()) for kf in kfs] == [0.0, 1.0, 2.0]
    kfs[0].is_anchored = True
    rp = _Replay(g)
    opt.set_draws(rp)
    opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(6, False, False, False, True))
    assert rp.i == int(g["n_draws_a"])
    loss_a = opt.last_stats["loss_terms"][:, 0].numpy()
    assert opt.last_stats["n_valid_rays"] == 6 * 3 * (48 + 16)
    assert rel(opt._model.nerf_model._model_sigma.params, g["params_a"]) < 5e-2            # (Adam: the tolerances of the G9 loop test)
    assert rel(opt._occupancy_grid_model.occupancy_grid[0, 0], g["grid_a"]) < 2e-3
    for i in range(3):
        assert np.abs(kfs[i].get_lidar_pose().get_pose_tensor().detach().cpu().numpy() - g[f"pose_a{i}"]).max() < 5e-4
    params_a = opt._model.nerf_model._model_sigma.params.detach().clone()
    opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(6, False, True, True, True))
    assert rp.i == int(g["n_draws"]) and opt._global_step == int(g["global_step"])
    assert torch.equal(opt._model.nerf_model._model_sigma.params.detach(), params_a)      # frozen density net
    loss = np.concatenate([loss_a, opt.last_stats["loss_terms"][:, 0].numpy()])
    print("loss trace vs reference: rel", np.abs(loss - g["losses"]).max() / np.abs(g["losses"]).max())
    assert np.abs(loss[:6] - g["losses"][:6]).max() < 1e-4 * np.abs(g["losses"]).max()
    assert np.abs(loss - g["losses"]).max() < 2e-3 * np.abs(g["losses"]).max()            # phase b starts from phase a's (Adam-stepped) map
    for i in range(3):
        p = kfs[i].get_lidar_pose().get_pose_tensor().detach().cpu().numpy()
        assert np.abs(p - g[f"pose_b{i}"]).max() < 5e-4
    travel = np.abs(g["pose_b2"] - g["pose_a2"]).max()
    err = np.abs(kfs[2].get_lidar_pose().get_pose_tensor().detach().cpu().numpy() - g["pose_b2"]).max()
    print(f"tracking phase: latest keyframe moved {travel:.2e}, error vs reference {err:.2e}")
    assert travel > 1e-3 and err < 0.2 * travel
    assert rel(opt._occupancy_grid_model.occupancy_grid[0, 0], g["grid_b"]) < 2e-3
    # the Adam of a tracking phase holds the pose only; its state_dict is what Mapper.build_ckpt saves (mapper.py:161-175)
    sd = opt._optimizer.state_dict()
    assert len(sd["param_groups"]) == 1 and sd["state"][0]["step"] == 6


def test_checkpoint_written_here_renders_like_the_reference_loading_it(golden, tmp_path):
    """G12: a checkpoint written by this repo's classes was loaded by the reference's Model / OccupancyGridModel, rendered by
    its Model.forward(testing=True) and scored by its compute_l1_depth (analysis/compute_l1_depth.py:42-64,140-155).  The same
    file loaded by OUR consumer side must render the same depths and score the same L1 on the same draws."""
    from loner_amd.analysis.l1_depth import compute_l1_depth
    from loner_amd.common.frame import Frame
    from loner_amd.common.pose import Pose
    from loner_amd.common.ray_utils import LidarRayDirections
    from loner_amd.common.sensors import LidarScan
    from loner_amd.models.model_tcnn import Model, OccupancyGridModel
    from loner_amd.models.ray_sampling import OccGridRaySampler
    from loner_amd.utils import synthetic as SY
    from tests import support
    g = golden("g12_checkpoint_l1_depth")
    path = str(tmp_path / "final.tar")
    meta = support.write_repo_checkpoint(path)
    assert float(meta["sigma_params"].double().sum()) == float(g["sigma_params_checksum"])
    mc = support.small_settings(48, 64, n_test=int(g["n_samples_test"])).model_config.model
    model, occ = Model(mc).to(DEV), OccupancyGridModel(mc.occ_model).to(DEV)
    back = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(back["network_state_dict"])                       # strict
    occ.load_state_dict(back["occ_model_state_dict"])
    sampler = OccGridRaySampler()
    sampler.update_occ_grid(occ().detach())
    dirs, ts = SY.lidar_pattern()
    sub = torch.from_numpy(g["scan_subset"])
    pose6 = torch.from_numpy(g["pose6"])
    scan = LidarScan(dirs[:, sub].clone(), SY.scene_ranges(dirs, OP.transform_from_pose6(pose6))[sub], ts[sub])

    class Draws:
        def pdf(self, n, h): return torch.from_numpy(g["u_pdf"])
        def noise(self, n, s_): return torch.from_numpy(g["noise"])
    sampler.set_draws(Draws())
    lrd = LidarRayDirections(scan, chunk_size=4096)
    wc = world_cube()
    rays = lrd.fetch_chunk_rays(0, Pose(pose_tensor=pose6.clone(), fixed=True), wc, torch.tensor([1.0, 50.0]))
    out = model(rays, sampler, wc.scale_factor, testing=True, return_variance=True, camera=False)
    e_depth = rel(out["depth_fine"], g["depth"])
    l1 = compute_l1_depth(Pose(pose_tensor=pose6.clone(), fixed=True), lrd, model, sampler, wc, torch.tensor([1.0, 50.0]), DEV)
    print(f"rendered depth vs reference rel {e_depth:.2e}; L1 depth {l1:.5f} m vs reference {float(g['l1']):.5f} m")
    assert out["depth_fine"].shape == (512,) and e_depth < 1e-4
    assert abs(l1 - float(g["l1"])) < 1e-4 * float(g["l1"])


def test_front_to_back_render_depth_equals_the_full_route(tmp_path):
    """Model.render_depth(front_to_back=True): the network is evaluated block by block along the ray and only on rays whose transmittance
    is still >= 2^-24 (lnr_render_ftb_*).  On a trained-like checkpoint (dense surfaces: most rays die within the first blocks) and on
    IDENTICAL random draws - importance draws and density noise replayed through the sampler's `draws` hook - the depth equals the
    default route's to 1e-6 relative and the L1 metric to 1e-5; several ray chunks, a ragged last one, and the in-kernel generator."""
    from loner_amd.analysis.l1_depth import compute_l1_depth
    from loner_amd.common.pose import Pose
    from loner_amd.common.ray_utils import LidarRayDirections
    from loner_amd.common.sensors import LidarScan
    from loner_amd.models.model_tcnn import Model, OccupancyGridModel
    from loner_amd.models.ray_sampling import OccGridRaySampler
    from loner_amd.utils import synthetic as SY
    from tests import support
    path = str(tmp_path / "final.tar")
    support.write_repo_checkpoint(path)
    n_test = 1024
    mc = support.small_settings(48, 64, n_test=n_test).model_config.model
    model, occ = Model(mc).to(DEV), OccupancyGridModel(mc.occ_model).to(DEV)
    back = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(back["network_state_dict"])
    occ.load_state_dict(back["occ_model_state_dict"])
    # surfaces: the output row of the density MLP (the first row of the padded [16][H] output matrix, the tail of the MLP block) x 300 -
    # where the network is positive the transmittance now dies within a few samples, as behind the walls of a trained map
    net = model.nerf_model._model_sigma
    n_mlp, H = int(net.spec.n_mlp_params), int(net.spec.n_neurons)
    with torch.no_grad():
        net.params[n_mlp - 16 * H:n_mlp - 15 * H] *= 300.0
    sampler = OccGridRaySampler()
    sampler.update_occ_grid(occ().detach())
    dirs, ts = SY.lidar_pattern()
    pose6 = SY.trajectory_pose6(4)[2]
    sub = torch.arange(5, dirs.shape[1], 21)[:3000]
    scan = LidarScan(dirs[:, sub].clone(), SY.scene_ranges(dirs, OP.transform_from_pose6(pose6))[sub], ts[sub])
    lrd = LidarRayDirections(scan, chunk_size=4096)
    wc = world_cube()
    rays = lrd.fetch_chunk_rays(0, Pose(pose_tensor=pose6.clone(), fixed=True), wc, torch.tensor([1.0, 50.0]))
    n = rays.shape[0]
    assert n > 2500

    class Draws:                                     # the same numbers for both routes: keyed by (kind, shape), not by call order
        def pdf(self, m, h): return torch.rand(m, h, generator=torch.Generator().manual_seed(11))
        def noise(self, m, s_): return torch.randn(m, s_, generator=torch.Generator().manual_seed(12))
    sampler.set_draws(Draws())
    model._FTB_RAYS = 1024                           # three chunks, the last one ragged
    full = model.render_depth(rays, sampler, wc.scale_factor, testing=True)
    ftb = model.render_depth(rays, sampler, wc.scale_factor, testing=True, front_to_back=True)
    assert ftb.shape == full.shape == (n,) and torch.isfinite(ftb).all()
    e = float(((ftb - full).abs() / full.abs().clamp_min(1e-6)).max())
    # how much the route skipped: the transmittance behind the first block of the default route's own weights
    out = model(rays, sampler, wc.scale_factor, testing=True, camera=False)
    w = out["weights_fine"]
    # how much the route had to skip: every weight behind sample 512 is at most the transmittance there, so rays whose weights behind it
    # sum to <= 2^-24 were dropped after the second of the four blocks at the latest
    dead_half = float((w[:, 512:].sum(1) <= 2.0 ** -24).float().mean())
    print(f"front-to-back vs full route: max relative depth difference {e:.2e}; rays dead behind sample 512 of {n_test}: {100 * dead_half:.1f} %")
    assert e <= 1e-6
    assert dead_half > 0.2                           # (the checkpoint is trained-like: the route has something to skip)
    pose = Pose(pose_tensor=pose6.clone(), fixed=True)
    l1_full = compute_l1_depth(pose, lrd, model, sampler, wc, torch.tensor([1.0, 50.0]), DEV)
    model.cfg.render["front_to_back"] = True         # the configuration switch compute_l1_depth's caller sees
    l1_ftb = compute_l1_depth(pose, lrd, model, sampler, wc, torch.tensor([1.0, 50.0]), DEV)
    assert abs(l1_ftb - l1_full) <= 1e-5 * l1_full
    # the in-kernel generator (no replayed draws): different launch sizes key different random numbers, so the two routes agree
    # statistically only - the mean depth over the scan, dominated by the surfaces, within 1e-3
    sampler.set_draws(None)
    model.cfg.render["front_to_back"] = False
    torch.manual_seed(5); a = model.render_depth(rays, sampler, wc.scale_factor, testing=True)
    torch.manual_seed(5); b = model.render_depth(rays, sampler, wc.scale_factor, testing=True, front_to_back=True)
    assert abs(float(a.mean()) - float(b.mean())) < 1e-3 * float(a.mean())


def test_optimizer_survives_spawn_pickling_and_mask_ray_selection():
    """The reference constructs the Optimizer in the parent and hands it to the mapping process through spawn
    (src/loner.py:59,188,205): a constructed Optimizer must pickle (no live HIP handles) and work after unpickling.  Also the
    MASK ray-selection strategy (optimizer.py:290-293): every drawn ray index lies inside the scan's mask."""
    import pickle
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    from loner_amd.utils import synthetic as SY
    s = small_settings(64, 64)
    s["rays_selection"]["strategy"] = "MASK"
    torch.manual_seed(0)
    opt = Optimizer(s, None, world_cube(), 0, False, True, False)
    blob = pickle.dumps(opt)
    opt2 = pickle.loads(blob)
    assert torch.equal(opt2._model.nerf_model._model_sigma.params.detach().cpu(), opt._model.nerf_model._model_sigma.params.detach().cpu())
    kf = make_keyframes([SY.trajectory_pose6(1)[0]])
    n = len(kf[0].get_lidar_scan())
    mask = torch.zeros(n, dtype=torch.bool)
    mask[1000:1400] = True
    kf[0].get_lidar_scan().mask = mask
    seen = []
    orig = opt2._draw_window_indices

    def spy(active, tab):
        idx = orig(active, tab)
        seen.append(idx.cpu())
        return idx
    opt2._draw_window_indices = spy
    opt2._do_iterate_optimizer(kf, [None], optimizer_settings=OptimizationSettings(5, True, False, False, True))
    assert torch.isfinite(opt2.last_stats["loss_terms"]).all() and opt2.last_stats["n_valid_rays"] == 5 * 64
    drawn = torch.cat(seen)
    assert drawn.numel() == 5 * 64 and int(drawn.min()) >= 1000 and int(drawn.max()) < 1400 and len(torch.unique(drawn)) > 100


def test_deferred_density_step_is_taken_and_changes_nothing():
    """The density Adam step is deferred to just before the next density forward (so that, sharded, the gradient all-reduce
    overlaps the pose tail and the next batch's ray build).  It must actually be deferred in a joint phase, and the result
    must be bit-identical to stepping right away."""
    from loner_amd.mapping import optimizer as OM
    from loner_amd.utils import synthetic as SY

    def run(defer):
        torch.manual_seed(0)
        opt = OM.Optimizer(small_settings(96, 64), None, world_cube(), 0, False, True, False)
        opt._defer_density_step = defer
        base = SY.trajectory_pose6(2)
        kfs = make_keyframes([base[0], base[1] + torch.tensor([0.02, 0.0, -0.01, 0.0, 0.0, 0.0])])
        kfs[0].is_anchored = True
        pending_seen = []
        flush = opt._flush_density_step

        def spy():
            pending_seen.append(opt._pending_density is not None)
            return flush()
        opt._flush_density_step = spy
        torch.manual_seed(3)
        opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OM.OptimizationSettings(12, False, False, False, True))
        p = opt._model.nerf_model._model_sigma.params
        return p.detach().clone(), opt._optimizer.state[p]["exp_avg"].clone(), kfs[1].get_lidar_pose().get_pose_tensor().detach().clone(), pending_seen

    a = run(True)
    b = run(False)
    assert sum(a[3]) >= 11                  # a pending density step was found (and flushed) before (almost) every forward
    assert sum(b[3]) == 0                   # eager mode never leaves one pending
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)


def test_cfg1_loop_matches_oracle_with_default_network():
    """BASELINE configs[0]: one synthetic 64x1024 frame, 512 rays x 128 samples, default network - 4 map-only
    iterations on the GPU against the CPU oracle consuming the same random draws."""
    from loner_amd.common.settings import default_optimizer_settings
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    from loner_amd.utils import synthetic as SY
    s = default_optimizer_settings()
    s["num_samples"]["sky"] = 0
    s["model_config"]["model"]["render"]["N_samples_train"] = 128
    torch.manual_seed(0)
    opt = Optimizer(s, None, world_cube(), 0, False, True, False)
    nc = s["model_config"]["model"]["nerf_config"]
    spec = NW.NetworkSpec.from_config(dict(nc["pos_encoding_sigma"]), dict(nc["sigma_network"]))
    params0 = opt._model.nerf_model._model_sigma.params.detach().cpu().clone()
    scale, shift = SY.world_cube()
    oracle = MS.OracleMapper(spec, params0, scale, shift, MS.MapperConfig(n_rays=512, n_samples=128), grid_size=100)
    base = SY.trajectory_pose6(1)
    kf = make_keyframes([base[0]])
    dirs, _ = SY.lidar_pattern()
    okf = [MS.OracleKeyframe(dirs, SY.scene_ranges(dirs, OP.transform_from_pose6(base[0])), base[0].clone(), anchored=True)]

    class Recorded(MS.TorchDraws):
        def __init__(self): self.log = []; self.replay = None; self.i = 0
        def _draw(self, fn, *a):
            if self.replay is None:
                v = getattr(MS.TorchDraws, fn)(self, *a); self.log.append(v.clone()); return v
            v = self.replay[self.i]; self.i += 1; return v
        def ray_index(self, n, c): return self._draw("ray_index", n, c)
        def jitter(self, n, h): return self._draw("jitter", n, h)
        def pdf(self, n, h): return self._draw("pdf", n, h)
        def noise(self, n, s_): return self._draw("noise", n, s_)
    rec = Recorded()
    torch.manual_seed(5)
    oracle.iterate(okf, 4, draws=rec)
    rep = Recorded(); rep.replay = rec.log
    opt.set_draws(rep)
    opt._do_iterate_optimizer(kf, [None], optimizer_settings=OptimizationSettings(4, True, False, False, True))
    assert rep.i == len(rec.log)
    gpu_loss = opt.last_stats["loss_terms"][:, 0].numpy()
    print("loss trace gpu", gpu_loss, "oracle", oracle.trace)
    assert np.abs(gpu_loss - np.array(oracle.trace)).max() / abs(oracle.trace[0]) < 1e-4
    p_gpu = opt._model.nerf_model._model_sigma.params.detach().cpu()
    moved = float((oracle.params - params0).abs().max())
    # Adam turns every gradient whose magnitude exceeds eps=1e-8 into a step of ~lr, so entries whose gradient is a
    # near-cancelling sum can differ by a whole step between two float32 summation orders; compare distributions.
    diff = (p_gpu - oracle.params).abs()
    print("params moved", moved, " max diff", float(diff.max()), " mean diff", float(diff.mean()), " frac > 10% of a step",
          float((diff > 0.1 * moved).float().mean()))
    nm = spec.n_mlp_params
    for name, d in (("mlp", diff[:nm]), ("table", diff[nm:])):
        q = torch.quantile(d[torch.randperm(d.numel())[:1000000]], torch.tensor([0.5, 0.9, 0.99, 0.999]))
        print(name, "diff quantiles 50/90/99/99.9%:", q.tolist(), "max", float(d.max()))
    touched = (oracle.params - params0).abs() > 0
    print("touched params", int(touched.sum()), "of", touched.numel(), " mean diff over touched", float(diff[touched].mean()))
    assert moved > 1e-3 and float(torch.quantile(diff[touched][:2000000], 0.9)) < 2e-2 * moved
    assert rel(opt._occupancy_grid_model.occupancy_grid[0, 0], oracle.grid[0, 0]) < 1e-3


def _guarded_run(inject, n_it=8, at=4, precision="fp32"):
    """A joint map + pose phase of n_it iterations on the small network; `inject(opt, pose_dev)` runs inside the loop right before
    the density forward of iteration `at` (no host sync).  -> (opt, keyframes, snapshot taken at that moment, raised exception)"""
    from loner_amd import ops
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    from loner_amd.utils import synthetic as SY
    s = small_settings(96, 64)
    s["model_config"]["model"]["nerf_config"]["sigma_network"]["precision"] = precision
    torch.manual_seed(0)
    opt = Optimizer(s, None, world_cube(), 0, False, True, False)
    base = SY.trajectory_pose6(2)
    kfs = make_keyframes([base[0], base[1] + torch.tensor([0.02, -0.01, 0.0, 0.0, 0.0, 0.0])])
    kfs[0].is_anchored = True
    snap, calls = {}, [0]
    orig_fwd, orig_pose_fwd = ops.density_forward, ops.pose_forward
    pose_seen = []

    def pose_fwd(p):
        pose_seen.append(p)
        return orig_pose_fwd(p)

    def fwd(spec, params, **kw):
        if calls[0] == at:
            # the deferred density step of iteration at-1 has landed (it is flushed right before this call): this is the state
            # the failing iteration starts from
            snap["params"] = params.detach().clone()
            snap["poses"] = pose_seen[-1].detach().clone()
            snap["grid"] = opt._occupancy_grid_model.occupancy_grid.detach().clone()
            inject(opt, pose_seen[-1], params)
        calls[0] += 1
        return orig_fwd(spec, params, **kw)
    ops.density_forward, ops.pose_forward = fwd, pose_fwd
    err = None
    try:
        torch.manual_seed(3)
        opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(n_it, False, False, False, True))
    except (RuntimeError, AssertionError) as e:
        err = e
    finally:
        ops.density_forward, ops.pose_forward = orig_fwd, orig_pose_fwd
    return opt, kfs, snap, err


def test_failure_in_iteration_k_stops_every_update_at_iteration_k():
    """The reference checks loss / pose gradient / pose in EVERY iteration and raises before optimizer.step()
    (optimizer.py:368-374,590).  Here the kernels of the failing iteration mark a device word and every later step kernel obeys
    it: parameters, poses and the occupancy grid keep the values they had when iteration k began, and the reference's error is
    raised at the end of the phase - without a host sync per iteration."""
    from loner_amd import hip

    # (a) an infinite MLP weight: sigma is clipped (finite loss, like the reference's nan_to_num), but the gradient that flows back
    # through that weight is not finite -> "invalid gradient in pose"
    def inf_weight(opt, pose_dev, params):
        params.view(-1)[5] = float("inf")
    opt, kfs, snap, err = _guarded_run(inf_weight)
    assert isinstance(err, RuntimeError) and str(err) == "Fatal: Encountered invalid gradient in pose."
    assert opt.last_failure == {"code": hip.POISON_POSE_GRAD, "iteration": 4}
    p = opt._model.nerf_model._model_sigma.params.detach()
    keep = torch.ones_like(p, dtype=torch.bool); keep[5] = False
    assert torch.equal(p[keep], snap["params"][keep]), "density parameters moved after the failing iteration"
    assert torch.equal(opt._occupancy_grid_model.occupancy_grid.detach(), snap["grid"])
    got = torch.stack([kf.get_lidar_pose().get_pose_tensor().detach().cpu() for kf in kfs])
    assert torch.equal(got, snap["poses"].cpu()), "poses moved after the failing iteration"
    assert torch.isfinite(got).all()

    # (b) a pose that turns NaN in the middle of iteration 4 (after its rays were built): the loss of that iteration is still
    # finite, its pose check is not -> one of the two pose errors, at iteration 4, and nothing moves from there on
    def nan_translation(opt, pose_dev, params):
        pose_dev.data[1, 0] = float("nan")
    opt, kfs, snap, err = _guarded_run(nan_translation)
    assert isinstance(err, RuntimeError) and str(err) in ("Fatal: Encountered invalid gradient in pose.", "Fatal: Encountered invalid pose tensor.")
    assert opt.last_failure["iteration"] == 4 and opt.last_failure["code"] in (hip.POISON_POSE, hip.POISON_POSE_GRAD)
    p = opt._model.nerf_model._model_sigma.params.detach()
    assert torch.equal(p, snap["params"]) and torch.isfinite(p).all(), "a step was applied after the failing iteration"
    assert torch.equal(opt._occupancy_grid_model.occupancy_grid.detach(), snap["grid"])

    # (c) a healthy run raises nothing and leaves the word at zero
    opt, kfs, snap, err = _guarded_run(lambda *a: None)
    assert err is None and opt.last_failure is None


def test_fp16_mode_clips_sigma_at_the_half_extremes_and_warns_once(capsys):
    """nerf_tcnn.py:51-52,70-78: non-finite densities are replaced by the extremes of the NETWORK's dtype (fp16 in the reference:
    +-65504; whatever exceeds that is +-inf in an fp16 output) and NaN by 0, with one warning."""
    from loner_amd import hip, ops
    from loner_amd.models.nerf_tcnn import DecoupledNeRF
    from loner_amd.common.settings import default_nerf_config
    for prec, lim in (("fp16", 65504.0), ("fp32", float(torch.finfo(torch.float32).max))):
        nc = default_nerf_config()
        nc["sigma_network"]["precision"] = prec
        torch.manual_seed(1)
        net = DecoupledNeRF(nc).to(DEV)
        sig = net._model_sigma
        n_mlp = int(sig.spec.n_mlp_params)
        pts = (torch.rand(4096, 3, device=DEV) * 1.6 - 0.8)
        with torch.no_grad():
            sig.params[n_mlp:] *= 1e4                       # features of order one
            out_row = sig.params[sig.spec.n_neurons * sig.spec.in_dim: sig.spec.n_neurons * sig.spec.in_dim + sig.spec.n_neurons]
            out_row.fill_(3e4 if prec == "fp16" else 1e38)  # sum of 64 relu(z) * 3e4 exceeds 65504 for many points, never inf in fp32 ...
            if prec == "fp32":
                out_row[0] = float("inf")                   # ... so the fp32 case gets a genuine inf
        before = ops.density_clipped_count(pts.device)
        capsys.readouterr()
        s = net(pts, sigma_only=True)
        text = capsys.readouterr().out
        assert torch.isfinite(s).all() and float(s.abs().max()) == lim
        assert ops.density_clipped_count(pts.device) > before
        assert text.count("Clipping infinite outputs") == 1 and net._warn_infinite is False
        net(pts, sigma_only=True)
        assert "Clipping" not in capsys.readouterr().out    # once only
        assert (net._max_float, net._min_float) == (lim, -lim)


def test_l1_depth_curve_matches_the_reference_on_its_own_draws(golden):
    """G13 ("matched L1 depth"): the reference's own Optimizer trained the DEFAULT density network on a reduced window (2 keyframes x
    256 rays x 128 samples, joint map + pose optimisation, phases of 50 / 50 / 100 / 200 iterations) and its compute_l1_depth scored
    512 held-out rays after every phase: 30.0 m -> 14.6 -> 10.9 -> 4.8 -> 3.3 m.  The HIP path replays the same run on the same random
    draws (regenerated from the recorded seed, call by call) and must follow that curve: the quality half of the benchmark metric is
    the reference's, not a self-assessment.  400 Adam iterations amplify rounding differences, hence a band rather than equality."""
    from loner_amd.analysis.l1_depth import compute_l1_depth
    from loner_amd.common.pose import Pose
    from loner_amd.common.pose_utils import WorldCube
    from loner_amd.common.ray_utils import LidarRayDirections
    from loner_amd.common.sensors import LidarScan
    from loner_amd.common.settings import default_optimizer_settings
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    from loner_amd.utils import synthetic as SY
    from tests import support
    g = golden("g13_l1_curve")
    cfg = support.G13
    s = default_optimizer_settings()
    s["num_samples"]["lidar"], s["num_samples"]["sky"] = cfg["n_rays"], 0
    s["model_config"]["model"]["render"]["N_samples_train"] = cfg["n_samples"]
    s["model_config"]["model"]["render"]["N_samples_test"] = cfg["n_test"]
    wc = WorldCube(torch.tensor(float(g["scale"])), torch.from_numpy(g["shift"]))
    opt = Optimizer(s, None, wc, 0, False, True, False)
    sig = opt._model.nerf_model._model_sigma
    spec_o = NW.NetworkSpec.from_config(sig.encoding_config, sig.network_config)
    with torch.no_grad():
        sig.params.copy_(NW.init_params(spec_o, seed=cfg["init_seed"]).to(DEV))          # the initial parameters of the recorded run
    assert float(sig.params.detach().double().sum().cpu()) == float(g["params0_sum"])
    base = SY.trajectory_pose6(8)
    kfs = make_keyframes([base[0], torch.from_numpy(g["pose1_init"])])
    kfs[0].is_anchored = True
    replay = support.SeededReplay(int(g["seed"]), g["draw_kind"], g["draw_args"])
    opt.set_draws(replay)
    opt._ray_sampler.set_draws(replay)
    dirs, ts = SY.lidar_pattern()
    sub = support.l1_scan_subset(cfg["n_l1_rays"])
    scan = LidarScan(dirs[:, sub].clone(), SY.scene_ranges(dirs, OP.transform_from_pose6(base[0]))[sub], ts[sub])
    lrd = LidarRayDirections(scan, chunk_size=512)
    rr = torch.tensor([1.0, 50.0])
    pose0 = Pose(pose_tensor=base[0].clone(), fixed=True)
    score = lambda: compute_l1_depth(pose0, lrd, opt._model, opt._ray_sampler, wc, rr, DEV)
    l1 = [score()]
    assert abs(l1[0] - float(g["l1_init"])) < 1e-3 * float(g["l1_init"])                   # same map, same draws: same number
    losses = []
    for ph, n_it in enumerate(int(v) for v in g["phases"]):
        opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(n_it, False, False, False, True))
        losses += opt.last_stats["loss_terms"][:, 0].tolist()
        l1.append(score())
        print(f"phase {ph}: L1 {l1[-1]:.4f} m (reference {float(g['l1'][ph]):.4f}), last loss {losses[-1]:.4f} (reference {float(g['losses'][len(losses) - 1]):.4f}), "
              f"pose error vs reference {float((kfs[1].get_lidar_pose().get_pose_tensor().detach().cpu() - torch.from_numpy(g[f'pose1_after{ph}'])).abs().max()):.2e}")
    assert replay.i == len(replay.kinds), "the run consumed fewer draws than the reference"
    assert abs(replay.checksum - float(g["draw_checksum"])) <= 1e-9 * abs(float(g["draw_checksum"])), "different random draws than the reference's"
    ref = g["l1"]
    # The first 50 iterations track the reference closely (loss trace to 1e-3, L1 14.53 - 14.56 vs 14.552 m); from there on the two fp32
    # trajectories decorrelate - Adam turns rounding differences into sign-sized steps - and the L1 of 512 rays after 200 / 400
    # iterations is a sample of a noisy quantity: measured on the MI355X 11.07 / 5.25 / 2.97 m (round 3) and 10.55 / 5.21 / 2.66 m
    # (round 4, the forward's interpolation accumulating with fmaf like tiny-cuda-nn) against the reference's 10.86 / 4.79 / 3.29 m:
    # two builds that differ in the last bit of a feature are 0.3 m apart after 400 iterations, as far as either is from the reference.
    # 5 % while the runs are comparable step by step, 25 % where they are only statistically (bench.py's matched_quality compares
    # 8 runs per leg for that reason).
    first = np.array(losses[:50]); ref_first = g["losses"][:50]
    assert np.abs(first - ref_first).max() < 2e-2 * np.abs(ref_first).max()
    done = np.cumsum(g["phases"])
    for ph in range(len(ref)):
        band = 0.05 if done[ph] <= 100 else 0.25
        assert abs(l1[ph + 1] - float(ref[ph])) < band * float(ref[ph]) + 0.05, (ph, l1[ph + 1], float(ref[ph]))
    assert l1[-1] < 0.5 * l1[0] and float(ref[-1]) < 0.5 * float(g["l1_init"])             # both off the plateau


def test_pipelined_loop_equals_the_single_stream_loop():
    """The training loop runs the pose tail, the occupancy step and the next iteration's front end (ray build, compaction, loss
    normalisers, sampler) on a second stream beside the table-gradient reduce (lnr_density_backward's input_grad_event).  Same
    kernels, same arguments, same order of the random draws: parameters, Adam state, poses, occupancy grid and the loss trace are
    bit-identical to the single-stream loop - with the in-kernel generator (joint phase incl. two occupancy steps, then a tracking
    phase with frozen parameters) and with the reference's recorded draws (G9)."""
    from loner_amd.mapping import optimizer as OM
    from loner_amd.utils import synthetic as SY

    def run(pipeline):
        s = small_settings(128, 64)
        torch.manual_seed(0)
        opt = OM.Optimizer(s, None, world_cube(), 0, False, True, False)
        opt._pipeline = pipeline
        base = SY.trajectory_pose6(3)
        kfs = make_keyframes([base[0], base[1] + torch.tensor([0.02, 0.0, -0.01, 0.0, 0.002, 0.0]), base[2] + torch.tensor([0.0, 0.01, 0.0, 0.001, 0.0, 0.0])])
        kfs[0].is_anchored = True
        torch.manual_seed(5)
        opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OM.OptimizationSettings(23, False, False, False, True))
        traces = [opt.last_stats["loss_terms"].clone()]
        opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OM.OptimizationSettings(7, False, True, True, True))     # tracking
        traces.append(opt.last_stats["loss_terms"].clone())
        p = opt._model.nerf_model._model_sigma.params
        used_side = opt._side_stream is not None
        return (p.detach().clone(), opt._occupancy_grid_model.occupancy_grid.detach().clone(),
                torch.stack([kf.get_lidar_pose().get_pose_tensor().detach().cpu() for kf in kfs]), traces[0], traces[1]), used_side

    (a, side_a), (b, side_b) = run(True), run(False)
    assert side_a and not side_b
    for n, x, y in zip(["params", "occupancy grid", "poses", "loss trace (joint)", "loss trace (tracking)"], a, b):
        assert torch.equal(x, y), f"{n} differ between the pipelined and the single-stream loop"
    assert float((a[2][1:] - a[2][0]).abs().max()) > 0


def _ref_shaped_window(pose6_list, scale_settings=None):
    """keyframes, world cube and settings as the reference's Mapper hands them to the Optimizer: its own classes (tests/support.py
    stand-ins that expose exactly the reference's members), CPU tensors, AttrDict settings"""
    from loner_amd.utils import synthetic as SY
    from tests.support import RefShapedFrame, RefShapedKeyFrame, RefShapedLidarScan, RefShapedPose, RefShapedSettings, RefShapedWorldCube
    dirs, ts = SY.lidar_pattern()
    base = SY.trajectory_pose6(8)
    kfs = []
    for i, p6 in enumerate(pose6_list):
        dist = SY.scene_ranges(dirs, OP.transform_from_pose6(base[i]))
        fr = RefShapedFrame(None, RefShapedLidarScan(dirs.clone(), dist, ts + float(i), sky_rays=torch.Tensor()), RefShapedPose())
        fr._lidar_pose = RefShapedPose(pose_tensor=p6.clone(), fixed=True)
        fr._gt_lidar_pose = RefShapedPose(pose_tensor=base[i].clone(), fixed=True)
        kfs.append(RefShapedKeyFrame(fr))
    scale, shift = SY.world_cube()
    wc = RefShapedWorldCube(torch.tensor(scale), torch.from_numpy(shift))
    s = small_settings(96, 64)
    s["data_prep_on_cpu"] = True
    s["keyframe_schedule"] = [{"num_keyframes": 1, "iteration_schedule": [{"num_iterations": 6, "freeze_poses": True, "freeze_sigma_mlp": False, "freeze_rgb_mlp": True}]},
                              {"num_keyframes": -1, "iteration_schedule": [
                                  {"num_iterations": 3, "freeze_poses": False, "latest_kf_only": True, "freeze_sigma_mlp": True, "freeze_rgb_mlp": True},
                                  {"num_iterations": 5, "freeze_poses": False, "freeze_sigma_mlp": False, "freeze_rgb_mlp": True}]}]
    s["skip_pose_refinement"] = False
    import json
    return kfs, wc, RefShapedSettings(json.loads(json.dumps(s)))        # plain nested dicts / LISTS, as yaml.load produces them


def test_optimizer_is_driven_by_reference_shaped_callers(tmp_path):
    """The drop-in boundary from the caller's side (mapper.py:62-66,104-130,161-175): the reference's Mapper constructs the
    Optimizer from ITS Settings (AttrDict: lists become tuples on attribute access) and ITS WorldCube, hands iterate_optimizer windows
    of ITS KeyFrame objects (CPU tensors, data_prep_on_cpu; Pose with get_pose_tensor / set_fixed / get_transformation_matrix) and
    reads state back through _keyframe_count, _global_step and four state_dict() calls.  The stand-ins refuse every member the
    reference classes do not have, so this run also proves that loner_amd touches nothing beyond that surface."""
    from loner_amd.mapping.optimizer import Optimizer
    from loner_amd.utils import synthetic as SY
    gen = torch.Generator().manual_seed(3)
    base = SY.trajectory_pose6(2)
    init = [base[0].clone(), base[1] + torch.cat([torch.randn(3, generator=gen) * 0.02, torch.randn(3, generator=gen) * 0.003])]
    kfs, wc, settings = _ref_shaped_window(init)
    settings["log_directory"] = str(tmp_path)
    assert isinstance(settings.keyframe_schedule, tuple) and isinstance(settings.model_config.model.ray_range, tuple)
    torch.manual_seed(0)
    opt = Optimizer(settings, None, wc, 0, False, True, False)              # the Mapper's call: mapper.py:62-66
    # keyframe 0 alone (mapper.py:104: first keyframe), then the two-keyframe window
    opt.iterate_optimizer([kfs[0]])
    assert kfs[0].is_anchored and opt._keyframe_count == 1 and opt._global_step == 6
    p0 = kfs[0].get_lidar_pose().get_pose_tensor().clone()
    p1_before = kfs[1].get_lidar_pose().get_pose_tensor().detach().clone()
    opt.iterate_optimizer(kfs)
    assert opt._keyframe_count == 2 and opt._global_step == 6 + 3 + 5
    # the anchored pose did not move, the other one was optimised IN PLACE on its CPU tensor (the Mapper keeps references to it)
    assert torch.equal(kfs[0].get_lidar_pose().get_pose_tensor(), p0)
    p1 = kfs[1].get_lidar_pose().get_pose_tensor()
    assert p1.device.type == "cpu" and p1.requires_grad and not torch.equal(p1.detach(), p1_before)
    assert torch.isfinite(p1).all() and float((p1.detach() - p1_before).abs().max()) < 0.1
    # what Mapper.build_ckpt reads (mapper.py:161-175)
    ck = {"global_step": opt._global_step, "network_state_dict": opt._model.state_dict(), "optimizer_state_dict": opt._optimizer.state_dict(),
          "poses": [kf.get_pose_state() for kf in kfs], "occ_model_state_dict": opt._occupancy_grid_model.state_dict(),
          "occ_optimizer_state_dict": opt._occupancy_grid_optimizer.state_dict()}
    torch.save(ck, str(tmp_path / "ckpt.tar"))
    assert set(ck["optimizer_state_dict"]) == {"state", "param_groups"} and len(ck["optimizer_state_dict"]["param_groups"]) == 2
    assert [l.split(",")[0] for l in (tmp_path / "timing.csv").read_text().splitlines()] == ["6", "8"]     # optimizer.py:179-186
    opt._global_step = 100                                                   # (rw: mapper.py:117)
    # a pose that was optimised and is frozen afterwards enters with the matrix ITS Pose hands out - for the reference's class the one
    # cached at construction (pose.py:140-144) - and use_gt_poses builds the rays from the ground-truth poses (keyframe.py:83-86)
    kfs[1].get_lidar_pose().set_fixed(True)
    stale = kfs[1].get_lidar_pose().get_transformation_matrix()
    fresh = OP.transform_from_pose6(kfs[1].get_lidar_pose().get_pose_tensor().detach())
    assert float((stale - fresh).abs().max()) > 1e-5                          # the cached matrix no longer matches the stepped vector
    from loner_amd.mapping.optimizer import OptimizationSettings
    seen = {}
    import loner_amd.ops as OPS
    orig = OPS.build_window_rays

    def spy(tab, transforms, *a, **k):
        seen["T"] = transforms.detach().cpu().clone()
        return orig(tab, transforms, *a, **k)
    OPS.build_window_rays = spy
    try:
        opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(1, True, False, False, True))
        assert rel(seen["T"][1].reshape(3, 4), stale[:3, :4]) < 1e-7
        opt_gt = Optimizer(settings, None, wc, 0, True, True, False)
        opt_gt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(1, False, False, False, True))
        for k in range(2):
            assert rel(seen["T"][k].reshape(3, 4), kfs[k]._frame._gt_lidar_pose.get_transformation_matrix()[:3, :4]) < 1e-7
    finally:
        OPS.build_window_rays = orig


def test_training_with_the_in_kernel_generator_reaches_the_oracles_l1_distribution():
    """The quality half of the metric, on the kernels' OWN random numbers: several runs of the HIP path and of the oracle (its torch
    ops on this GPU, torch's generator) on the G13 configuration, each run with different draws, ONE optimisation phase (one Adam)
    each - the means of L1 depth after the same number of iterations must agree within one pooled standard deviation (with a floor of
    3 %) and both legs must have left the plateau.  (BENCH_r03's 12.4 m vs 9.5 - 11.1 m was not a biased generator: the baseline legs
    restarted Adam in every iteration.)  bench.py reports the same comparison with 8 runs per leg at 100 iterations."""
    import bench
    iters, seeds = 60, (0, 1, 2)
    q = bench._Shape(bench.QUALITY_SHAPE.keyframes, bench.QUALITY_SHAPE.rays, bench.QUALITY_SHAPE.samples)
    hip = [bench.hip_quality_run(s, iters=iters) for s in seeds]
    ora = [bench.oracle_leg(q, "cuda", budget_s=0.0, max_iters=iters, min_iters=iters, seed=s) for s in seeds]
    h = np.array([r["l1_depth_m_after"] for r in hip]); o = np.array([r["l1_depth_m_after"] for r in ora])
    h0 = np.array([r["l1_depth_m_before"] for r in hip]); o0 = np.array([r["l1_depth_m_before"] for r in ora])
    print(f"L1 after {iters} iterations: HIP {h.round(3).tolist()} (from {h0.mean():.2f}), torch-ROCm oracle {o.round(3).tolist()} (from {o0.mean():.2f})")
    assert abs(h0.mean() - o0.mean()) < 0.02 * o0.mean()                         # same untrained map, same probe
    assert h.max() < 0.6 * h0.mean() and o.max() < 0.6 * o0.mean()                # both trained
    pooled = np.sqrt((h.var(ddof=1) + o.var(ddof=1)) / 2)
    assert abs(h.mean() - o.mean()) < max(pooled, 0.03 * o.mean()), (h.mean(), o.mean(), pooled)


@pytest.mark.parametrize("pipeline", [True, False])
def test_overwrite_gradient_mode_changes_nothing(pipeline):
    """LNR_BWD_OVERWRITE_GRAD: the training loop lets the table-gradient reduce and the weight-gradient fold STORE the gradient (no read
    of the old one) and Adam not zero it - 60 MB of HBM traffic less per iteration.  Bit-identical to accumulate + zero, in the pipelined
    and in the single-stream loop; and the gradient buffer is zero when the phase ends, as after the reference's last zero_grad."""
    from loner_amd.mapping import optimizer as OM
    from loner_amd.utils import synthetic as SY

    def run(overwrite):
        torch.manual_seed(0)
        opt = OM.Optimizer(small_settings(96, 64), None, world_cube(), 0, False, True, False)
        opt._overwrite_grads, opt._pipeline = overwrite, pipeline
        base = SY.trajectory_pose6(2)
        kfs = make_keyframes([base[0], base[1] + torch.tensor([0.02, 0.0, -0.01, 0.0, 0.0, 0.0])])
        kfs[0].is_anchored = True
        torch.manual_seed(3)
        opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OM.OptimizationSettings(12, False, False, False, True))
        p = opt._model.nerf_model._model_sigma.params
        st = opt._optimizer.state[p]
        assert float(p.grad.abs().max()) == 0.0
        return p.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), kfs[1].get_lidar_pose().get_pose_tensor().detach().clone(), \
            opt._occupancy_grid.detach().clone(), opt.last_stats["loss_terms"].clone()
    for x, y in zip(run(True), run(False)):
        assert torch.equal(x, y)
