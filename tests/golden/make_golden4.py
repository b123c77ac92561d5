#!/usr/bin/env python3
"""Fourth batch of golden fixtures (round 5), produced by IMPORTING THE REFERENCE in the build container.

    python tests/golden/make_golden4.py

G14  the first iterations of the G9 loop, state by state.  G9 pins the reference's 12-iteration trajectory only at its end,
     where Adam has amplified last-bit differences into a 5e-2 band on the parameters - a band that would also pass a wrong
     bias correction.  G14 re-runs the SAME optimisation (same initial parameters, same keyframes, torch.manual_seed(77): the
     draws are the first 5 k of G9's) for k = 1, 2, 3 iterations of the reference's Optimizer._do_iterate_optimizer
     (src/mapping/optimizer.py:194-424) and records, after each: the density parameters, Adam's exp_avg / exp_avg_sq / step
     for the density group and the pose group (torch.optim.Adam as the reference constructs it, optimizer.py:257-269), and
     the free pose.  The test asserts those tightly before the loose 12-iteration check.

The stand-ins of make_golden.py apply (tinycudann -> oracle.network for the density net, pytorch3d -> oracle.poses);
everything else executing below is the reference's own code.  The fixture is data; nothing reads /root/reference at test time.
"""
import os
import sys
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


def main():
    MG.install_stubs()
    from common.pose_utils import compute_world_cube
    from common.sensors import LidarScan
    from common.pose import Pose
    from common.frame import Frame
    from common.settings import Settings
    from mapping.keyframe import KeyFrame
    from mapping.optimizer import Optimizer, OptimizationSettings

    torch.set_num_threads(8)
    wc = compute_world_cube(None, None, None, None, (1, 50), padding=0.3,
                            traj_bounding_box={"x": [-25, 10], "y": [-25, 15], "z": [-10, 10]})
    scan_dirs, _ = SY.lidar_pattern()
    poses6 = SY.trajectory_pose6(8)
    ranges = [SY.scene_ranges(scan_dirs, OP.transform_from_pose6(p)) for p in poses6]

    # the optimiser settings of G9 (make_golden.py: the G7-G9 block)
    S_all = Settings.load_from_file(os.path.join(MG.REF, "cfg/defaults.yaml"))
    S_opt = Settings(dict(S_all["mapper"]["optimizer"]))
    S_opt["debug"] = {k: False for k in S_all["debug"]["flags"]}
    S_opt["log_directory"] = "/tmp/loner_golden_logs"
    os.makedirs(S_opt["log_directory"], exist_ok=True)
    mc = S_opt["model_config"]
    mc["data"]["ray_range"] = [1, 50]
    mc["model"]["ray_range"] = [1, 50]
    mc["model"]["render"]["N_samples_train"] = 64
    S_opt["num_samples"]["lidar"] = 48
    S_opt["num_samples"]["sky"] = 0
    mc["model"]["nerf_config"]["pos_encoding_sigma"].update(
        dict(n_levels=4, log2_hashmap_size=12, base_resolution=8, n_features_per_level=2))
    mc["model"]["nerf_config"]["sigma_network"].update(dict(n_neurons=32, n_hidden_layers=1))
    mc["model"]["occ_model"]["voxel_size"] = 32

    def make_kf(i, noise_seed=None):
        p6 = poses6[i].clone()
        if noise_seed is not None:
            gen = torch.Generator().manual_seed(noise_seed)
            p6[:3] += torch.randn(3, generator=gen) * 0.02
            p6[3:] += torch.randn(3, generator=gen) * np.deg2rad(0.2)
        scan = LidarScan(scan_dirs.clone(), ranges[i].clone(), torch.linspace(0, 0.1, scan_dirs.shape[1]),
                         sky_rays=torch.Tensor())
        fr = Frame(None, scan, Pose())
        fr._lidar_pose = Pose(pose_tensor=p6.clone(), fixed=False)
        fr._gt_lidar_pose = Pose(pose_tensor=poses6[i].clone(), fixed=True)
        fr._lidar_start_time = torch.tensor(float(i)); fr._lidar_end_time = torch.tensor(float(i) + 0.1)
        return KeyFrame(fr, "cpu")

    g9 = dict(np.load(os.path.join(HERE, "g9_loop.npz")))
    out = {}
    for k in (1, 2, 3):
        opt = Optimizer(S_opt, S_all.calibration, wc, "cpu", False, True, False)
        sig = opt._model.nerf_model._model_sigma
        with torch.no_grad():
            sig.params[sig.spec.n_mlp_params:] *= 3000.0
        assert np.array_equal(sig.params.detach().numpy(), g9["params0"]), "not the initial parameters of G9"
        kfs = [make_kf(0, None), make_kf(1, 21)]
        kfs[0].is_anchored = True
        assert np.array_equal(kfs[1].get_lidar_pose().get_pose_tensor().detach().numpy(), g9["pose_init1"])
        opt._progress_bar = mock.MagicMock()
        torch.manual_seed(77)
        with MG.Recorder() as rec:
            opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(k, False, False, False, True))
        # the draws are G9's first 5 k, bit for bit (same generator, same call order)
        keys = sorted(n for n in g9 if n.startswith("draw"))
        assert len(rec.log) == 5 * k
        for j, (fn, t) in enumerate(rec.log):
            assert keys[j].endswith(fn) and np.array_equal(t.numpy(), g9[keys[j]]), (k, j)
        sd = opt._optimizer.state_dict()
        groups = sd["param_groups"]
        assert len(groups) == 2 and len(groups[0]["params"]) == 1 and len(groups[1]["params"]) == 1
        st_sigma, st_pose = sd["state"][groups[0]["params"][0]], sd["state"][groups[1]["params"][0]]
        out[f"params_{k}"] = sig.params.detach().clone()
        out[f"exp_avg_{k}"] = st_sigma["exp_avg"].clone()
        out[f"exp_avg_sq_{k}"] = st_sigma["exp_avg_sq"].clone()
        out[f"step_{k}"] = np.int64(int(st_sigma["step"]))
        out[f"pose1_{k}"] = kfs[1].get_lidar_pose().get_pose_tensor().detach().clone()
        out[f"pose_exp_avg_{k}"] = st_pose["exp_avg"].clone()
        out[f"pose_exp_avg_sq_{k}"] = st_pose["exp_avg_sq"].clone()
        out[f"grid_{k}"] = opt._occupancy_grid_model.occupancy_grid[0, 0].detach().clone()
        print(k, "param travel", float((sig.params.detach() - torch.from_numpy(g9["params0"])).abs().max()),
              "pose travel", float((out[f"pose1_{k}"] - torch.from_numpy(g9["pose_init1"])).abs().max()))
    MG.save("g14_loop_first_steps", lr_sigma=np.float64(groups[0]["lr"]), lr_pose=np.float64(groups[1]["lr"]), **out)


if __name__ == "__main__":
    main()
