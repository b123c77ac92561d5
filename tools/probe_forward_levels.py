"""encode_forward_kernel level by level (development build with -DLNR_ABLATE: LNR_X_FWD_LEVELS is read at every call) on the two
batches the product runs: the training window of the benchmark after some iterations (8 x 512 rays x 512 samples, what the mapping
iteration hands to lnr_density_forward) and one launch of the inference path (8192 rays of a 64 x 1024 scan x 2048 samples).  Times
are the library's own HIP events around the kernel.  (Round 4 used it with a second switch, LNR_X_PAIR_LEVELS, to compare lane modes
level by level: profiles/r04_forward_levels.txt.)
    LNR_LIB_PATH=loner_amd/_lib/libloner_hip_ablate.so python tools/probe_forward_levels.py [--warmup 60] [--dtype f32]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from loner_amd import ops                                      # noqa: E402
from loner_amd.common.ray_utils import LidarRayDirections      # noqa: E402
from loner_amd.mapping.optimizer import OptimizationSettings   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--warmup", type=int, default=60)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
phase = lambda n: OptimizationSettings(n, False, False, False, True)
opt = bench.make_bench_optimizer(512, 512, a.dtype)
window = bench.build_window(8)
opt._do_iterate_optimizer(window, [None], optimizer_settings=phase(a.warmup))
net = opt._model.nerf_model._model_sigma
spec, params = net.spec, net.params.detach()

batches = {}
orig = ops.density_forward


def grab(spec_, params_, **kw):
    if "training window" not in batches:
        nd = kw.get("n_rays_dev")
        batches["training window"] = dict(rays=kw["rays"].clone(), z=kw["z"].clone(), n_rays_dev=None if nd is None else nd.clone())
    return orig(spec_, params_, **kw)


ops.density_forward = grab
try:
    opt._do_iterate_optimizer(window, [None], optimizer_settings=phase(1))
finally:
    ops.density_forward = orig
kf = window[0]
scan = kf.get_lidar_scan()
lrd = LidarRayDirections(scan, chunk_size=len(scan))
T = kf.get_lidar_pose().get_transformation_matrix().detach()
with torch.no_grad():
    rays, _ = lrd.build_lidar_rays(torch.arange(len(scan)), opt._ray_range, opt._world_cube, T)
    r = rays[:8192].contiguous()
    z = opt._ray_sampler.get_samples(r, int(opt._model.cfg.render.N_samples_test), 0)
batches["inference launch"] = dict(rays=r, z=z, n_rays_dev=None)


def encode_ms(batch, levels):
    os.environ["LNR_X_FWD_LEVELS"] = hex(levels)
    ops.density_forward(spec, params, forward_only=True, **batch)
    torch.cuda.synchronize()
    ops.profile_read(); ops.profile_enable(True)
    for _ in range(a.reps):
        ops.density_forward(spec, params, forward_only=True, **batch)
    torch.cuda.synchronize()
    p = ops.profile_read(); ops.profile_enable(False)
    k = [v for n, v in p.items() if n.startswith("encode_forward")]
    return k[0]["avg_ms"] if k else float("nan")


n_levels = int(spec.n_levels)
every = (1 << n_levels) - 1
for name, b in batches.items():
    print(f"== {name}: {tuple(b['z'].shape)} samples, {a.dtype}; encode_forward ms (mean of {a.reps})")
    print("level   entries    res   ms")
    for lv in range(n_levels):
        print(f"{lv:5d} {int(spec.level_size[lv]):9d} {int(spec.level_res[lv]):6d} {encode_ms(b, 1 << lv):10.4f}")
    print(f"all levels: {encode_ms(b, every):.4f}")
