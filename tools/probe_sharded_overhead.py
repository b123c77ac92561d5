"""What the SHARDED form of the loop costs on its own: one process, one GPU, torch.distributed over RCCL with world size 1 - every
collective of an iteration (far[0] key, loss normalisers, the gradient exchange in both forms, the parameter all-gather) is issued
and is an identity, so the difference to the non-distributed loop on the same one-keyframe window is the price of issuing them
(host time, stream hand-overs), which a rank of an 8-GPU window pays whatever the links do.
    python tools/probe_sharded_overhead.py [--steps 300]"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from loner_amd.mapping.optimizer import OptimizationSettings   # noqa: E402
from loner_amd.mapping.sharding import DistContext             # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--only", type=int, default=-1, help="run one configuration of the list below (for a kernel trace of exactly that loop)")
a = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29713")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
phase = lambda n: OptimizationSettings(n, False, False, False, True)
import gc
for i_cfg, (form, front, native, grad) in enumerate(((None, None, None, None),
                                  ("all_reduce", "inline", False, "async"), ("all_reduce", "async", False, "async"),
                                  ("reduce_scatter", "inline", False, "async"), ("reduce_scatter", "async", False, "async"),
                                  ("all_reduce", "inline", True, "async"), ("all_reduce", "async", True, "async"),
                                  ("reduce_scatter", "inline", True, "async"), ("reduce_scatter", "async", True, "async"))):
    if a.only >= 0 and i_cfg != a.only:
        continue
    opt = bench.make_bench_optimizer(512, 512, "f32")
    window = bench.build_window(8)[:1]
    if form is not None:
        opt.set_distributed(DistContext(exchange=form, front=front, native=native))
    opt._do_iterate_optimizer(window, [None], optimizer_settings=phase(a.warmup))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt._do_iterate_optimizer(window, [None], optimizer_settings=phase(a.steps))
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    what = "non-distributed loop" if form is None else (f"sharded loop, world size 1, {'lnr_comm (own RCCL binding)' if native else 'torch.distributed (ProcessGroupNCCL)'}, "
                                                        f"exchange {form}, front collective {front}")
    print(f"one keyframe x 512 rays x 512 samples, {what}: {ms:.4f} ms per iteration", flush=True)
    del opt
    gc.collect()
dist.destroy_process_group()
