#!/usr/bin/env python3
"""Second batch of golden fixtures (round 2), again produced by IMPORTING THE REFERENCE in the build container.

    python tests/golden/make_golden2.py

G11  sky rays + tracking phase: the reference's Optimizer._do_iterate_optimizer (src/mapping/optimizer.py:194-424) on
     three keyframes with sky rays (keyframe.py:91-100, sensors.py:162-167): a joint map + pose phase, then the
     pose-refinement phase of the schedule (latest_kf_only, frozen density net: optimizer.py:239-259, defaults.yaml:88-92).
     Recorded: every random draw in call order, the loss of every iteration, poses / parameters / occupancy grid
     after each phase.
G12  a checkpoint WRITTEN BY THIS REPO (loner_amd's Model / OccupancyGridModel state_dicts in the dictionary layout of
     Mapper.build_ckpt, mapper.py:161-175) is loaded by the REFERENCE's Model / OccupancyGridModel
     (analysis/compute_l1_depth.py:140-155), rendered with the reference's Model.forward(testing=True) and scored by
     the reference's compute_l1_depth (analysis/compute_l1_depth.py:42-64).  Recorded: the importance-sampling draws,
     per-ray depths, the L1 value.

The stand-ins of make_golden.py apply (tinycudann -> oracle.network for the density net, pytorch3d -> oracle.poses);
everything else executing below is the reference's own code.  The fixtures are data; nothing reads /root/reference
at test time.
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import make_golden as MG                      # noqa: E402  (stubs, Recorder, save)
from loner_amd.utils import synthetic as SY   # noqa: E402
from oracle import poses as OP                # noqa: E402
from tests import support                     # noqa: E402


def small_optimizer_settings(Settings, S_all, n_lidar, n_sky, n_samples, n_test=256):
    S_opt = Settings(dict(S_all["mapper"]["optimizer"]))
    S_opt["debug"] = {k: False for k in S_all["debug"]["flags"]}
    S_opt["log_directory"] = "/tmp/loner_golden_logs"
    os.makedirs(S_opt["log_directory"], exist_ok=True)
    mc = S_opt["model_config"]
    mc["data"]["ray_range"] = [1, 50]
    mc["model"]["ray_range"] = [1, 50]
    mc["model"]["render"]["N_samples_train"] = n_samples
    mc["model"]["render"]["N_samples_test"] = n_test
    S_opt["num_samples"]["lidar"] = n_lidar
    S_opt["num_samples"]["sky"] = n_sky
    mc["model"]["nerf_config"]["pos_encoding_sigma"].update(dict(support.SMALL_ENC))
    mc["model"]["nerf_config"]["sigma_network"].update(dict(n_neurons=32, n_hidden_layers=1))
    mc["model"]["occ_model"]["voxel_size"] = 32
    return S_opt


def main():
    MG.install_stubs()
    from common.pose_utils import compute_world_cube
    from common.sensors import LidarScan
    from common.pose import Pose
    from common.frame import Frame
    from common.settings import Settings
    from common.ray_utils import LidarRayDirections
    from mapping.keyframe import KeyFrame
    from mapping.optimizer import Optimizer, OptimizationSettings
    from models.model_tcnn import Model, OccupancyGridModel
    from models.ray_sampling import OccGridRaySampler

    torch.manual_seed(2)
    np.random.seed(2)
    torch.set_num_threads(8)
    wc = compute_world_cube(None, None, None, None, (1, 50), padding=0.3,
                            traj_bounding_box={"x": [-25, 10], "y": [-25, 15], "z": [-10, 10]})
    scale, shift = wc.scale_factor, wc.shift
    scan_dirs, _ = SY.lidar_pattern()
    poses6 = SY.trajectory_pose6(8)
    ranges = [SY.scene_ranges(scan_dirs, OP.transform_from_pose6(p)) for p in poses6]
    S_all = Settings.load_from_file(os.path.join(MG.REF, "cfg/defaults.yaml"))

    # ------------------------------------------------------------------------------------------ G11
    S_opt = small_optimizer_settings(Settings, S_all, n_lidar=48, n_sky=16, n_samples=64)
    sky = support.sky_directions()

    def make_kf(i, noise_seed=None):
        p6 = poses6[i].clone()
        if noise_seed is not None:
            gen = torch.Generator().manual_seed(noise_seed)
            p6[:3] += torch.randn(3, generator=gen) * 0.02
            p6[3:] += torch.randn(3, generator=gen) * np.deg2rad(0.2)
        n = scan_dirs.shape[1]
        scan = LidarScan(scan_dirs.clone(), ranges[i].clone(), torch.linspace(float(i), float(i) + 0.1, n), sky_rays=sky.clone())
        fr = Frame(None, scan, Pose())
        fr._lidar_pose = Pose(pose_tensor=p6.clone(), fixed=False)
        fr._gt_lidar_pose = Pose(pose_tensor=poses6[i].clone(), fixed=True)
        return KeyFrame(fr, "cpu")

    opt = Optimizer(S_opt, S_all.calibration, wc, "cpu", False, True, True)        # sky segmentation enabled
    sig = opt._model.nerf_model._model_sigma
    with torch.no_grad():
        sig.params[sig.spec.n_mlp_params:] *= 3000.0
    params0 = sig.params.detach().clone()
    kfs = [make_kf(0), make_kf(1, 31), make_kf(2, 32)]
    kfs[0].is_anchored = True
    assert [float(kf.get_time()) for kf in kfs] == [0.0, 1.0, 2.0]
    pose_init = [kf.get_lidar_pose().get_pose_tensor().detach().clone() for kf in kfs]
    opt._progress_bar = mock.MagicMock()
    losses = []
    orig_loss = opt.compute_loss

    def logging_loss(*a, **k):
        v = orig_loss(*a, **k)
        losses.append(float(v.detach()))
        return v
    opt.compute_loss = logging_loss
    torch.manual_seed(91)
    out = {}
    with MG.Recorder() as rec:
        opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(6, False, False, False, True))
        n_draws_a = len(rec.log)
        out.update(params_a=sig.params.detach().clone(), grid_a=opt._occupancy_grid_model.occupancy_grid[0, 0].detach().clone(),
                   **{f"pose_a{i}": kf.get_lidar_pose().get_pose_tensor().detach().clone() for i, kf in enumerate(kfs)})
        opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(6, False, True, True, True))
    assert torch.equal(sig.params.detach(), out["params_a"]), "tracking phase must not touch the density parameters"
    draws = {f"draw{j:03d}_{fn}": t for j, (fn, t) in enumerate(rec.log)}
    MG.save("g11_sky_tracking", params0=params0, sky=sky, scale=scale, shift=shift,
            **{f"pose_init{i}": p for i, p in enumerate(pose_init)},
            **{f"pose_b{i}": kf.get_lidar_pose().get_pose_tensor() for i, kf in enumerate(kfs)},
            grid_b=opt._occupancy_grid_model.occupancy_grid[0, 0], losses=np.array(losses), n_draws_a=np.int64(n_draws_a),
            n_draws=np.int64(len(rec.log)), global_step=np.int64(opt._global_step), **out, **draws)

    # ------------------------------------------------------------------------------------------ G12
    S_opt = small_optimizer_settings(Settings, S_all, n_lidar=48, n_sky=0, n_samples=64, n_test=256)
    ckpt_path = "/tmp/loner_golden_logs/repo_final.tar"
    meta = support.write_repo_checkpoint(ckpt_path)                    # written by loner_amd classes (no GPU needed to build a state_dict)
    model_cfg = S_opt.model_config.model
    model = Model(model_cfg)
    occ = OccupancyGridModel(model_cfg.occ_model)
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    ref_keys, repo_keys = set(model.state_dict()), set(ckpt["network_state_dict"])
    assert ref_keys <= repo_keys, f"reference Model expects keys the repo checkpoint lacks: {ref_keys - repo_keys}"
    # the colour-branch tensors of the stand-in tinycudann modules have no meaningful size; the density branch and the occupancy
    # grid - what the lidar path uses - are loaded strictly
    own = model.state_dict()
    loadable = {k: v for k, v in ckpt["network_state_dict"].items() if k in own and own[k].shape == v.shape}
    assert "nerf_model._model_sigma.params" in loadable
    model.load_state_dict(loadable, strict=False)
    occ.load_state_dict(ckpt["occ_model_state_dict"])                  # strict
    assert torch.equal(model.nerf_model._model_sigma.params.detach(), meta["sigma_params"])
    sampler = OccGridRaySampler()
    sampler.update_occ_grid(occ().detach())
    sub = support.l1_scan_subset()
    scan = LidarScan(scan_dirs[:, sub].clone(), ranges[1][sub].clone(), torch.linspace(0, 0.1, len(sub)))
    pose = Pose(pose_tensor=poses6[1].clone(), fixed=True)

    # the reference's scoring function, imported from its analysis script (its other imports are bag/GUI tooling)
    ru = types.ModuleType("render_utils"); ru.np = np
    sys.modules["render_utils"] = ru
    for name in ["rosbag", "rospy", "ros_numpy", "examples", "examples.run_loner", "pandas", "tqdm"]:
        sys.modules.setdefault(name, mock.MagicMock())
    sys.path.insert(0, os.path.join(MG.REF, "analysis"))
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("ref_compute_l1_depth", os.path.join(MG.REF, "analysis", "compute_l1_depth.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    lrd = LidarRayDirections(scan, chunk_size=mod.CHUNK_SIZE)
    depths_log = []
    orig_forward = model.forward

    def logging_forward(*a, **k):
        r = orig_forward(*a, **k)
        depths_log.append(r["depth_fine"].detach().clone())
        return r
    model.forward = logging_forward
    torch.manual_seed(17)
    with MG.Recorder() as rec:
        l1 = mod.compute_l1_depth(pose, lrd, (model, sampler, wc, torch.Tensor([1, 50]), "cpu"), False)
    # testing=True: no jitter, but the importance draw AND the density noise (Model.forward hands raw_noise_std to render_rays
    # in test mode too, model_tcnn.py:92) are random
    assert [fn for fn, _ in rec.log] == ["rand", "randn"]
    MG.save("g12_checkpoint_l1_depth", l1=np.float64(float(l1)), depth=torch.cat(depths_log), u_pdf=rec.log[0][1], noise=rec.log[1][1],
            scan_subset=sub, pose6=poses6[1], scale=scale, shift=shift, sigma_params_checksum=np.float64(float(meta["sigma_params"].double().sum())),
            n_samples_test=np.int64(256))
    print("reference compute_l1_depth on the repo-written checkpoint:", float(l1))


if __name__ == "__main__":
    main()
