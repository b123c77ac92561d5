"""What one rank of a G-GPU window runs per iteration, measured on ONE GPU without any communication: one keyframe of the window,
the Adam step of the hash tables restricted to a 1/G slice (the `reduce_scatter` exchange form, mapping/sharding.py: a rank steps
its slice, the slices are all-gathered).  The exchange itself (29.7 MB over xGMI) is what this leaves out.
    python tools/probe_shard_iteration.py [--ranks 8] [--steps 300]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from loner_amd.mapping.optimizer import OptimizationSettings   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--warmup", type=int, default=20)
a = ap.parse_args()
phase = lambda n: OptimizationSettings(n, False, False, False, True)
for ranks in (1, a.ranks):
    opt = bench.make_bench_optimizer(512, 512, "f32")
    window = bench.build_window(8)[:1]
    if ranks > 1:
        spec = opt._model.nerf_model._model_sigma.spec
        n_mlp, n_all = int(spec.n_mlp_params), int(spec.n_params)
        chunk = (n_all - n_mlp) // ranks

        def step_slice(work, group, o=opt):
            o._optimizer.step(zero_grad=not o._overwrite_grads, groups=(group,), ranges=[(0, n_mlp), (n_mlp, n_mlp + chunk)])
        opt._step_density = step_slice
    opt._do_iterate_optimizer(window, [None], optimizer_settings=phase(a.warmup))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt._do_iterate_optimizer(window, [None], optimizer_settings=phase(a.steps))
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    print(f"one keyframe x 512 rays x 512 samples, Adam over {'the whole table' if ranks == 1 else f'1/{ranks} of the table'}: {ms:.4f} ms per iteration")
