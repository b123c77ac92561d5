"""Where the HOST spends its time in one iteration of the sharded loop: RCCL at world size 1, a one-keyframe window with a GPU workload
small enough that the GPU idles (16 rays x 64 samples), cProfile over 300 iterations - next to the non-distributed loop on the same window.
    python tools/probe_host_sharded.py"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from loner_amd.mapping.optimizer import OptimizationSettings   # noqa: E402
from loner_amd.mapping.sharding import DistContext             # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29714")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
phase = lambda n: OptimizationSettings(n, False, False, False, True)
STEPS = 300
for form in (None, "all_reduce", "reduce_scatter"):
    opt = bench.make_bench_optimizer(16, 64, "f32")
    window = bench.build_window(8)[:1]
    if form is not None:
        opt.set_distributed(DistContext(exchange=form))
    opt._do_iterate_optimizer(window, [None], optimizer_settings=phase(20))
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    opt._do_iterate_optimizer(window, [None], optimizer_settings=phase(STEPS))
    torch.cuda.synchronize()
    pr.disable()
    ms = 1e3 * (time.perf_counter() - t0) / STEPS
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print(f"===== {'non-distributed loop' if form is None else 'sharded loop, world size 1, exchange ' + form}: host-bound iteration {ms:.4f} ms (under cProfile)")
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:34]))
dist.destroy_process_group()
