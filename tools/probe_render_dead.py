"""Development probe (round 6): how much of an inference launch a front-to-back route could skip.  For the bench scene after 110 / 310 /
1000 training iterations: the fraction of (ray, 256-sample block) pairs of a 64 x 1024 scan x 2048 samples that start with a
transmittance below 2^-24 (their weights are below fp32 resolution of the rendered depth)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from loner_amd import ops
from loner_amd.common.ray_utils import LidarRayDirections
from loner_amd.mapping.optimizer import OptimizationSettings

opt = bench.make_bench_optimizer(512, 512, "f32")
window = bench.build_window(8)
phase = lambda n: OptimizationSettings(n, False, False, False, True)
kf = window[0]
scan = kf.get_lidar_scan()
lrd = LidarRayDirections(scan, chunk_size=len(scan))
done = 0
for target in (110, 310, 1000, 3000):
    opt._do_iterate_optimizer(window, [None], optimizer_settings=phase(target - done)); done = target
    T = kf.get_lidar_pose().get_transformation_matrix().detach()
    with torch.no_grad():
        rays, depths = lrd.build_lidar_rays(torch.arange(len(scan)), opt._ray_range, opt._world_cube, T)
        rays = rays[::16].contiguous()                                    # 4096 rays
        model, sampler = opt._model, opt._ray_sampler
        S = int(model.cfg.render.N_samples_test)
        z = sampler.get_samples(rays, S, 0.0)
        net = model.nerf_model._model_sigma
        sigma = ops.density_forward(net.spec, net.params.detach(), rays=rays, z=z, forward_only=True)
        d = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], 1) * rays[:, 3:6].norm(dim=1, keepdim=True)
        alpha = 1.0 - torch.exp(-torch.relu(sigma) * d)
        Tr = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha[:, :-1] + 1e-10], 1), 1)      # transmittance in front of each sample
        dead = Tr < 2.0 ** -24
        frac_samples = float(dead.float().mean())
        blocks = dead[:, ::256]                                              # a block is skipped when its FIRST sample is already dead
        frac_blocks = float(blocks.float().mean())
        w = alpha * Tr
        depth = (w * z).sum(1)
        depth_cut = (w * z * (~dead[:, ::256].repeat_interleave(256, 1))).sum(1)
        rel = float(((depth - depth_cut).abs() / depth.abs().clamp_min(1e-6)).max())
    print(f"after {target} iterations: samples behind T < 2^-24: {100 * frac_samples:.1f} %; 256-sample blocks that could be skipped: {100 * frac_blocks:.1f} %; "
          f"max relative change of the rendered depth {rel:.2e}", flush=True)
