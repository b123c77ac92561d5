"""The keyframe-sharded optimisation loop (loner_amd/mapping/sharding.py + Optimizer.set_distributed) with the real HIP
kernels: two processes share the one GPU of the test box and talk over gloo (RCCL refuses two ranks on one device; the
collectives are the same torch.distributed calls).  What must hold on any backend: every rank ends with bit-identical
density parameters, Adam state and occupancy grid (replicas never drift), the loss goes down, keyframes of other ranks
are left alone."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tests.test_gpu_mapping import make_keyframes, small_settings, world_cube
    from loner_amd.mapping.optimizer import OptimizationSettings, Optimizer
    from loner_amd.mapping.sharding import DistContext
    from loner_amd.utils import synthetic as SY
    s = small_settings(128, 64)
    torch.manual_seed(0)                                    # identical initial parameters on every rank
    opt = Optimizer(s, None, world_cube(), 0, False, True, False)
    base = SY.trajectory_pose6(4)
    poses = [base[0]] + [p.clone() + torch.tensor([0.03, -0.02, 0.01, 0.0, 0.0, 0.0]) for p in base[1:]]
    window = make_keyframes(poses)
    window[0].is_anchored = True
    ctx = DistContext()
    opt.set_distributed(ctx)
    mine = ctx.owned(window)
    before = [kf.get_lidar_pose().get_pose_tensor().detach().clone() for kf in window]
    torch.manual_seed(100 + rank)                           # different ray draws per rank
    opt._do_iterate_optimizer(mine, [None], optimizer_settings=OptimizationSettings(25, False, False, False, True))
    torch.cuda.synchronize()
    params = opt._model.nerf_model._model_sigma.params.detach()
    st = opt._optimizer.state[opt._model.nerf_model._model_sigma.params]
    blobs = [params, st["exp_avg"], st["exp_avg_sq"], opt._occupancy_grid_model.occupancy_grid.detach().reshape(-1)]
    sums = torch.stack([b.double().sum() for b in blobs] + [b.double().abs().sum() for b in blobs]).cpu()
    gathered = [torch.zeros_like(sums) for _ in range(world)]
    dist.all_gather(gathered, sums)
    loss = opt.last_stats["loss_terms"][:, 0]
    moved = [float((kf.get_lidar_pose().get_pose_tensor().detach() - b).abs().max()) for kf, b in zip(window, before)]
    ret[rank] = dict(sums=[g.tolist() for g in gathered], loss0=float(loss[0]), loss1=float(loss[-1]), finite=bool(torch.isfinite(params).all()),
                     moved=moved, owned=[any(kf is m for m in mine) for kf in window], step=opt._global_step)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_replicas_stay_identical():
    world = 2
    port = 29600 + (os.getpid() % 200)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "a rank of the sharded run failed"
    r0, r1 = ret[0], ret[1]
    assert r0["sums"][0] == r0["sums"][1] == r1["sums"][0] == r1["sums"][1]      # bit-identical replicas (sum and |sum| of each blob)
    for r in (r0, r1):
        assert r["finite"] and r["step"] == 25 and r["loss1"] < r["loss0"]
        for moved, owned, k in zip(r["moved"], r["owned"], range(4)):
            assert (moved > 0) == (owned and k != 0)          # only this rank's non-anchored keyframes move
    assert r0["owned"] == [True, False, True, False] and r1["owned"] == [False, True, False, True]
