#!/usr/bin/env python3
"""bench.py -- training rays/s of the LONER mapping iteration on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one mapping iteration (optimizer.py:276-385 of the reference) over the default window:
8 keyframes x 512 rays x 512 samples, default network (16-level hash grid -> 64 -> 1), joint
optimisation of the density field and of 7 of the 8 poses (the first is anchored), occupancy-grid
step every 10th iteration, synthetic 64x1024 scans of an analytic scene (loner_amd/utils/synthetic.py).
With N > 1 the SAME 8-keyframe window is sharded over the ranks (keyframe i -> rank i mod N) with an
RCCL all-reduce of the density gradient per step ("strong" scaling).

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--keyframes", type=int, default=8)
    ap.add_argument("--rays", type=int, default=512)
    ap.add_argument("--samples", type=int, default=512)
    ap.add_argument("--dtype", choices=["f32", "f16"], default="f32",
                    help="arithmetic of the density network: f32 (default, stricter than the reference) or f16 "
                         "(the reference's storage types: fp16 features and weights on MFMA, fp32 accumulation)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def build_window(n_kf, device=None):
    from loner_amd.common.frame import Frame
    from loner_amd.common.pose import Pose
    from loner_amd.common.sensors import LidarScan
    from loner_amd.mapping.keyframe import KeyFrame
    from loner_amd.utils import synthetic as SY
    from loner_amd.common.pose_utils import tensor_to_transform
    dirs, ts = SY.lidar_pattern()
    base = SY.trajectory_pose6(n_kf)
    gen = torch.Generator().manual_seed(1)
    kfs = []
    for i in range(n_kf):
        dist = SY.scene_ranges(dirs, tensor_to_transform(base[i]))
        p6 = base[i].clone()
        if i > 0:                                   # initial pose error N(0, 2 cm / 0.2 deg)
            p6[:3] += torch.randn(3, generator=gen) * 0.02
            p6[3:] += torch.randn(3, generator=gen) * 0.2 * 3.14159265 / 180
        fr = Frame(None, LidarScan(dirs.clone(), dist, ts + 3.0 * i, sky_rays=torch.Tensor()), Pose())
        fr._lidar_pose = Pose(pose_tensor=p6, fixed=False)
        fr._gt_lidar_pose = Pose(pose_tensor=base[i].clone(), fixed=True)
        kfs.append(KeyFrame(fr, device))
    kfs[0].is_anchored = True
    return kfs


def cpu_baseline(args, budget_s=12.0):
    """The oracle (CPU restatement of the reference's mapping iteration, oracle/mapping_step.py) timed on the
    host cores on a bounded sample of the same workload: ONE keyframe x 512 rays x 512 samples per iteration,
    default network, as many iterations as fit in ~budget_s seconds (at least 1, at most 40; the every-10th-step
    occupancy update is part of the sample)."""
    from oracle import mapping_step as MS
    from oracle import network as NW
    from loner_amd.common.settings import default_nerf_config
    from loner_amd.utils import synthetic as SY
    from oracle import poses as OP
    threads = min(os.cpu_count() or 1, 16)           # torch CPU ops on [512,512] tensors stop scaling beyond this
    torch.set_num_threads(threads)
    nc = default_nerf_config()
    spec = NW.NetworkSpec.from_config(nc["pos_encoding_sigma"], nc["sigma_network"])
    scale, shift = SY.world_cube()
    cfg = MS.MapperConfig(n_rays=args.rays, n_samples=args.samples)
    m = MS.OracleMapper(spec, NW.init_params(spec, 0), scale, shift, cfg, grid_size=100)
    m.global_step = 1
    dirs, _ = SY.lidar_pattern()
    base = SY.trajectory_pose6(1)
    kfs = [MS.OracleKeyframe(dirs, SY.scene_ranges(dirs, OP.transform_from_pose6(base[0])), base[0].clone(), anchored=True)]
    torch.manual_seed(0)
    t0 = time.time()
    n_valid, iters = 0, 0
    while iters < 40 and (iters == 0 or time.time() - t0 < budget_s):
        n_valid += m.iterate(kfs, 1)
        iters += 1
    dt = time.time() - t0
    return {"value": n_valid / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{iters} mapping iterations of 1 keyframe x {args.rays} rays x {args.samples} samples (map-only, pose anchored), "
                      f"default network, torch CPU fp32, {dt:.1f} s wall",
            "ms_per_iter": 1e3 * dt / iters}


class KernelTimer:
    """HIP-event timing of selected C-ABI calls on the stream they are launched on (torch's current stream)."""

    def __init__(self, ops, names):
        self.ops, self.names = ops, names
        self.events = {n: [] for n in names}
        self.enabled = False
        self._orig = {}
        for n in names:
            self._orig[n] = getattr(ops, n)
            setattr(ops, n, self._wrap(n))

    def _wrap(self, name):
        orig = self._orig[name]

        def inner(*a, **k):
            if not self.enabled:
                return orig(*a, **k)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            self.events[name].append((e0, e1))
            return r
        return inner

    def summary(self):
        out = {}
        for n, evs in self.events.items():
            if evs:
                ms = [a.elapsed_time(b) for a, b in evs]
                out[n] = {"calls": len(ms), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms)}
        return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    # LNR_DIST_BACKEND=gloo lets two ranks share one GPU (RCCL refuses that): used to exercise the sharded code path on a 1-GPU box
    backend = os.environ.get("LNR_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()

    from loner_amd import ops
    from loner_amd.common.pose_utils import WorldCube
    from loner_amd.common.settings import default_optimizer_settings
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    from loner_amd.mapping.sharding import DistContext
    from loner_amd.utils import synthetic as SY

    settings = default_optimizer_settings(log_directory=f"/tmp/loner_amd_bench_{rank}")
    settings["num_samples"]["lidar"] = args.rays
    settings["num_samples"]["sky"] = 0
    settings["model_config"]["model"]["render"]["N_samples_train"] = args.samples
    settings["model_config"]["model"]["nerf_config"]["sigma_network"]["precision"] = "fp16" if args.dtype == "f16" else "fp32"
    scale, shift = SY.world_cube()
    torch.manual_seed(0)                                   # identical initial parameters on every rank
    opt = Optimizer(settings, None, WorldCube(torch.tensor(scale), torch.from_numpy(shift)), local, False, True, False)
    window = build_window(args.keyframes)
    if world > 1:
        ctx = DistContext()
        opt.set_distributed(ctx)                           # the optimiser shards the window itself: keyframe i -> rank i mod N
        torch.manual_seed(1000 + rank)                     # different ray draws per rank
    my_window = window
    phase = lambda n: OptimizationSettings(n, False, False, False, True)

    timer = KernelTimer(ops, ["density_backward", "density_forward", "los_loss_fused", "sample_rays_occ", "adam_step",
                              "occ_grid_step", "compact_rays", "lidar_rays_backward", "points_grad_to_rays"])
    # ---- warm-up (untimed) ----
    if args.warmup > 0:
        opt._do_iterate_optimizer(my_window, [None], optimizer_settings=phase(args.warmup))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    ops.profile_enable(True)
    t0 = time.perf_counter()
    opt._do_iterate_optimizer(my_window, [None], optimizer_settings=phase(args.steps))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False

    n_valid = torch.tensor([float(opt.last_stats["n_valid_rays"]), elapsed], device="cuda", dtype=torch.float64)
    if world > 1:
        rays_total = n_valid[0:1].clone(); dist.all_reduce(rays_total)
        t_max = n_valid[1:2].clone(); dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        total_rays, elapsed = float(rays_total), float(t_max)
    else:
        total_rays = float(n_valid[0])

    kprof = ops.profile_read()                     # kernels inside lnr_density_forward / _backward, HIP events in the library
    ops.profile_enable(False)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # quality half of the metric ("at matched L1 depth"): render held-out rays of the first keyframe with the trained
    # map (Model.forward(testing=True), 2048 samples) and compare with the analytic ranges.  Outside the timed region.
    l1_depth = None
    try:
        from loner_amd.analysis.l1_depth import compute_l1_depth
        from loner_amd.common.ray_utils import LidarRayDirections
        kf0 = my_window[0]
        l1_depth = compute_l1_depth(kf0.get_lidar_pose(), LidarRayDirections(kf0.get_lidar_scan(), chunk_size=2048), opt._model,
                                    opt._ray_sampler, opt._world_cube, opt._ray_range, opt._device, max_rays=4096)
    except Exception as e:      # never let the quality probe break the benchmark line
        l1_depth = f"failed: {e}"

    ksum = timer.summary()
    spec = opt._model.nerf_model._model_sigma.spec
    n_local = opt.last_stats["n_valid_rays"] / max(args.steps, 1)          # rays per launch on this rank
    pts = n_local * args.samples
    F = int(spec.n_features)
    n_rec = sum(1 for l in range(int(spec.n_levels)) if int(spec.level_size[l]) * F > 12288)      # levels of the record kernel
    rec_table_floats = sum(int(spec.level_size[l]) * F for l in range(int(spec.n_levels)) if int(spec.level_size[l]) * F > 12288)
    h, ind, nh = spec.n_neurons, spec.in_dim, spec.n_hidden
    mac = h * ind + (nh - 1) * h * h
    # Algorithmic work per launch:
    #   encode_backward: d_feature planes in, z and the ray records once, 6 ray-gradient floats out, the table gradient once
    #   mlp_backward:    forward recompute + input gradient + weight gradient GEMMs of the fp32 MLP
    alg = {
        "encode_backward": {"bytes": pts * (n_rec * F * 4.0 + 4.0) + n_local * (24.0 + 24.0) + rec_table_floats * 4.0},
        "encode_forward": {"bytes": pts * (4.0 + int(spec.n_levels) * F * 4.0) + n_local * 24.0 + (float(spec.n_params) - spec.n_mlp_params) * 4.0},
        "table_grad_reduce": {"bytes": rec_table_floats * 8.0},
        "mlp_backward": {"flops": pts * 2.0 * (3 * mac + h), "bytes": pts * (3 * spec.enc_dim * 4.0 + 4.0)},
        "mlp_forward": {"flops": pts * 2.0 * (mac + h), "bytes": pts * (spec.enc_dim * 4.0 + 4.0)},
    }
    traffic = {}
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf))
        except Exception:
            traffic = {}
    kernels = {}
    for name, v in kprof.items():
        ent = {"avg_ms": round(v["avg_ms"], 4), "calls": v["calls"]}
        a_ = alg.get(name)
        if a_:
            t = v["avg_ms"] * 1e-3
            if "bytes" in a_:
                ent["hbm_GBps"] = round(a_["bytes"] / t / 1e9, 1); ent["hbm_frac"] = round(a_["bytes"] / t / 8e12, 4)
            if "flops" in a_:
                ent["fp32_mfma_TFLOPs"] = round(a_["flops"] / t / 1e12, 2); ent["mfma_frac"] = round(a_["flops"] / t / 157.3e12, 4)
        kernels[name] = ent
    for name, v in ksum.items():
        if name not in ("density_forward", "density_backward"):
            kernels[name] = {"avg_ms": round(v["avg_ms"], 4), "calls": v["calls"]}
    roofline = None
    if kprof:
        dom = max(kprof, key=lambda k: kprof[k]["total_ms"])
        t = kprof[dom]["avg_ms"] * 1e-3
        a_ = alg.get(dom, {"bytes": 0.0})
        roofline = {"kernel": dom, "bound": "hbm", "achieved": a_.get("bytes", 0.0) / t / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": a_.get("bytes", 0.0) / t / 8e12, "traffic": traffic.get(dom + "_bytes_per_launch"),
                    "avg_launch_ms": kprof[dom]["avg_ms"], "algorithmic_bytes_per_launch": a_.get("bytes", 0.0),
                    "note": "dominant kernel by total time over the timed region, timed with HIP events recorded by the library on the "
                            "launch stream (lnr_profile_*).  It moves few algorithmic bytes: its time goes to L1 line lookups of the "
                            "random 8-byte table gathers, VALU work of the in-LDS radix partition and the 8-byte gradient records it "
                            "streams to HBM (DESIGN.md 4.3); `traffic` = PMC FETCH_SIZE+WRITE_SIZE bytes per launch (profiles/traffic.json)",
                    "secondary": {k: v for k, v in kernels.items() if k in ("mlp_backward", "mlp_forward", "encode_forward", "table_grad_reduce")}}
    line = {
        "metric": "training rays/sec", "value": total_rays / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"mapping iteration, {args.keyframes}-keyframe window x {args.rays} rays x {args.samples} samples, "
                               "default cfg (HashGrid 16x2 T=2^18 -> 64 -> 1, L1_JS loss, OGM sampler), joint map+pose optimisation, "
                               "synthetic 64x1024 scans (stand-in for Fusion Portable canteen, BASELINE configs[1])",
                   "keyframes": args.keyframes, "rays_per_keyframe": args.rays, "samples_per_ray": args.samples,
                   "parallelism": f"keyframe-sharded x{world}" if world > 1 else "single GPU"},
        "roofline": roofline,
        # whole-path HBM roofline of SURVEY 8d: B_ray = 72 B (ray record, gt depth, per-ray outputs) + dense Adam traffic
        # (28 B per parameter: read p,g,m,v, write p,m,v) amortised over the rays of an iteration
        "path_hbm_roofline": (lambda b: {"algorithmic_bytes_per_ray": b, "rays_per_s_at_8TBps": 8e12 / b * world,
                                         "frac": (total_rays / elapsed) / (8e12 / b * world)})(
            72.0 + 28.0 * float(spec.n_params) / max(total_rays / max(args.steps, 1) / world, 1.0)),
        "kernels_ms": {k: v["avg_ms"] for k, v in kernels.items()},
        "ops_ms": {k: round(v["avg_ms"], 4) for k, v in ksum.items()},
        "final_loss": float(opt.last_stats["loss_terms"][-1, 0]),
        "l1_depth_m": l1_depth, "iterations_trained": args.warmup + args.steps,
    }
    print(json.dumps({k: v for k, v in line.items() if k != "cpu_baseline"}), file=sys.stderr, flush=True)   # progress copy
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
