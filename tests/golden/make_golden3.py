#!/usr/bin/env python3
"""Third batch of golden fixtures (round 3), produced by IMPORTING THE REFERENCE in the build container.

    python tests/golden/make_golden3.py

G13  "matched L1 depth": the reference's own Optimizer (src/mapping/optimizer.py:194-424) trains the DEFAULT density network
     (cfg/nerf_config/default_nerf_hash.yaml: 16 levels x 2 features, T = 2^18, 64 neurons) on a reduced window - 2 keyframes x
     256 rays x 128 samples, joint map + pose optimisation - in four phases of 50 / 50 / 100 / 200 iterations, and after each
     phase the reference's compute_l1_depth (analysis/compute_l1_depth.py:42-64) scores 512 held-out rays of the first keyframe
     (Model.forward(testing=True), 256 samples).  Recorded: the L1 curve at 50 / 100 / 200 / 400 iterations, the loss of every
     iteration, poses and a parameter checksum after each phase.

     The random draws (400 x {2 randint, 2 rand, 1 randn} + 5 x {rand, randn}) would be 200 MB; they are NOT stored.  Every draw
     comes from torch's global CPU generator, seeded once: the fixture holds the seed, the call signature (kind, arguments) of
     every draw in order and a checksum of their values, and this script verifies that a fresh generator with that seed reproduces
     the recorded values call by call - which is how the GPU test regenerates them (tests/support.SeededReplay).

Stand-ins as in make_golden.py (tinycudann -> oracle.network, pytorch3d -> oracle.poses); the optimisation loop, the sampler, the
renderer, the loss and the L1 metric are the reference's own code executing.
"""
import os
import sys
import time
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import make_golden as MG                      # noqa: E402
from loner_amd.utils import synthetic as SY   # noqa: E402
from oracle import poses as OP                # noqa: E402
from tests import support                     # noqa: E402

SEED = 131
PHASES = (50, 50, 100, 200)


class SigRecorder(MG.Recorder):
    """also keeps the arguments of every draw, so that the sequence can be regenerated from the seed alone"""

    def _wrap(self, fn):
        orig = self._orig[fn]

        def inner(*a, **k):
            r = orig(*a, **k)
            assert not k or set(k) <= {"device"}, (fn, k)
            self.log.append((fn, a, r.clone()))
            return r
        return inner


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

    torch.set_num_threads(8)
    wc = compute_world_cube(None, None, None, None, (1, 50), padding=0.3,
                            traj_bounding_box={"x": [-25, 10], "y": [-25, 15], "z": [-10, 10]})
    scan_dirs, _ = SY.lidar_pattern()
    poses6 = SY.trajectory_pose6(8)
    ranges = [SY.scene_ranges(scan_dirs, OP.transform_from_pose6(p)) for p in poses6]
    S_all = Settings.load_from_file(os.path.join(MG.REF, "cfg/defaults.yaml"))
    S_opt = Settings(dict(S_all["mapper"]["optimizer"]))
    S_opt["debug"] = {k: False for k in S_all["debug"]["flags"]}
    S_opt["log_directory"] = "/tmp/loner_golden_logs"
    os.makedirs(S_opt["log_directory"], exist_ok=True)
    mc = S_opt["model_config"]
    mc["data"]["ray_range"] = [1, 50]
    mc["model"]["ray_range"] = [1, 50]
    mc["model"]["render"]["N_samples_train"] = support.G13["n_samples"]
    mc["model"]["render"]["N_samples_test"] = support.G13["n_test"]
    S_opt["num_samples"]["lidar"] = support.G13["n_rays"]
    S_opt["num_samples"]["sky"] = 0
    # the DEFAULT density network and occupancy grid (V = 100)

    def make_kf(i, noise_seed=None):
        p6 = poses6[i].clone()
        if noise_seed is not None:
            gen = torch.Generator().manual_seed(noise_seed)
            p6[:3] += torch.randn(3, generator=gen) * 0.02
            p6[3:] += torch.randn(3, generator=gen) * np.deg2rad(0.2)
        n = scan_dirs.shape[1]
        scan = LidarScan(scan_dirs.clone(), ranges[i].clone(), torch.linspace(float(i), float(i) + 0.1, n), sky_rays=torch.Tensor())
        fr = Frame(None, scan, Pose())
        fr._lidar_pose = Pose(pose_tensor=p6.clone(), fixed=False)
        fr._gt_lidar_pose = Pose(pose_tensor=poses6[i].clone(), fixed=True)
        return KeyFrame(fr, "cpu")

    opt = Optimizer(S_opt, S_all.calibration, wc, "cpu", False, True, False)
    sig = opt._model.nerf_model._model_sigma                     # oracle.network, parameters from NW.init_params(spec, seed=1234)
    assert sig.spec.n_params == 7416832
    params0_sum = float(sig.params.detach().double().sum())
    kfs = [make_kf(0), make_kf(1, support.G13["pose_noise_seed"])]
    kfs[0].is_anchored = True
    opt._progress_bar = mock.MagicMock()
    losses = []
    orig_loss = opt.compute_loss

    def logging_loss(*a, **k):
        v = orig_loss(*a, **k)
        losses.append(float(v.detach()))
        return v
    opt.compute_loss = logging_loss

    # the reference's scoring function
    ru = types.ModuleType("render_utils"); ru.np = np
    sys.modules["render_utils"] = ru
    for name in ["rosbag", "rospy", "ros_numpy", "examples", "examples.run_loner", "pandas", "tqdm"]:
        sys.modules.setdefault(name, mock.MagicMock())
    sys.path.insert(0, os.path.join(MG.REF, "analysis"))
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("ref_compute_l1_depth", os.path.join(MG.REF, "analysis", "compute_l1_depth.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    sub = support.l1_scan_subset(support.G13["n_l1_rays"])
    l1_scan = LidarScan(scan_dirs[:, sub].clone(), ranges[0][sub].clone(), torch.linspace(0, 0.1, len(sub)))
    l1_pose = Pose(pose_tensor=poses6[0].clone(), fixed=True)
    lrd = LidarRayDirections(l1_scan, chunk_size=mod.CHUNK_SIZE)

    def score():
        return float(mod.compute_l1_depth(l1_pose, lrd, (opt._model, opt._ray_sampler, wc, torch.Tensor([1, 50]), "cpu"), False))

    torch.manual_seed(SEED)
    out, l1s = {}, []
    t0 = time.time()
    with SigRecorder() as rec:
        l1_init = score()
        for ph, n_it in enumerate(PHASES):
            opt._do_iterate_optimizer(kfs, [None], optimizer_settings=OptimizationSettings(n_it, False, False, False, True))
            opt._ray_sampler.update_occ_grid(opt._occupancy_grid.detach())
            l1s.append(score())
            out[f"pose1_after{ph}"] = kfs[1].get_lidar_pose().get_pose_tensor().detach().clone()
            out[f"params_sum_after{ph}"] = np.float64(float(sig.params.detach().double().sum()))
            out[f"params_abs_after{ph}"] = np.float64(float(sig.params.detach().double().abs().sum()))
            print(f"phase {ph}: {n_it} iterations, L1 {l1s[-1]:.4f} m, loss {losses[-1]:.4f}, {time.time() - t0:.0f} s", flush=True)
    # the draws are reproducible from the seed alone: regenerate every one of them with a private generator
    gen = torch.Generator().manual_seed(SEED)
    checksum, sig_kind, sig_args = 0.0, [], []
    for fn, a, r in rec.log:
        if fn == "randint":
            args = (0,) + tuple(a) if len(a) == 2 else tuple(a)          # randint(high, size) / randint(low, high, size)
            again = torch.randint(args[0], args[1], tuple(args[2]), generator=gen)
            sig_args.append([args[1], int(np.prod(args[2])), 0])
        else:
            shape = tuple(a[0]) if len(a) == 1 and isinstance(a[0], (list, tuple, torch.Size)) else tuple(a)
            again = getattr(torch, fn)(*shape, generator=gen)
            sig_args.append([0, int(shape[0]), int(np.prod(shape[1:]))])
        assert again.shape == r.shape and torch.equal(again, r), f"draw {len(sig_kind)} ({fn}{a}) is not reproducible from the seed"
        sig_kind.append({"randint": 0, "rand": 1, "randn": 2}[fn])
        checksum += float(r.double().sum())
    MG.save("g13_l1_curve", seed=np.int64(SEED), phases=np.array(PHASES), l1_init=np.float64(l1_init), l1=np.array(l1s), losses=np.array(losses),
            draw_kind=np.array(sig_kind, np.int8), draw_args=np.array(sig_args, np.int64), draw_checksum=np.float64(checksum),
            params0_sum=np.float64(params0_sum), pose1_init=kfs[1].get_lidar_pose().get_pose_tensor().detach() * 0 + make_kf(1, support.G13["pose_noise_seed"]).get_lidar_pose().get_pose_tensor().detach(),
            scale=wc.scale_factor, shift=wc.shift, **out)
    print("L1 curve (reference):", l1_init, l1s)


if __name__ == "__main__":
    main()
