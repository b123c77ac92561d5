"""Development probe: why bench.py's extended_run (200 iterations after the timed region) is slower than the same iterations in a fresh run -
bench.py's own sequence (kernel timer on every 10th iteration of the timed region, then off) against the plain loop."""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
from loner_amd import ops
from loner_amd.mapping.optimizer import OptimizationSettings

phase = lambda n: OptimizationSettings(n, False, False, False, True)
for with_timer in (False, True):
    opt = bench.make_bench_optimizer(512, 512, "f32")
    window = bench.build_window(8)
    timer = None
    if with_timer:
        timer = bench.KernelTimer(ops, ["density_backward", "density_forward", "los_loss_fused", "sample_rays_occ", "adam_step",
                                        "occ_grid_step", "compact_rays", "lidar_rays_backward", "points_grad_to_rays"])
    opt._do_iterate_optimizer(window, [None], optimizer_settings=phase(10))
    torch.cuda.synchronize()
    if timer:
        timer.calls = {n: 0 for n in timer.names}
        timer.every = 10; timer.enabled = True; timer.reserve(12)
        ops.profile_enable(True); ops.profile_enable(False)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt._do_iterate_optimizer(window, [None], optimizer_settings=phase(100))
    torch.cuda.synchronize()
    print(f"kernel timer {with_timer}: timed region 10..110: {1e3 * (time.perf_counter() - t0) / 100:.3f} ms per iteration", flush=True)
    if timer:
        timer.enabled = False
        print("   calls per op in the timed region:", timer.calls, flush=True)
        kprof = ops.profile_read(); ops.profile_enable(False)
    t0 = time.perf_counter()
    opt._do_iterate_optimizer(window, [None], optimizer_settings=phase(200))
    torch.cuda.synchronize()
    print(f"kernel timer {with_timer}: extended 110..310: {1e3 * (time.perf_counter() - t0) / 200:.3f} ms per iteration", flush=True)
    if timer:
        for n in timer.names:
            setattr(ops, n, timer._orig[n])
    del opt
