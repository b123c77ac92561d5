"""Keyframe-window sharding (loner_amd/mapping/sharding.py) on CPU: 2 processes, gloo backend.

The HIP kernels normalise the loss with GLOBAL counts (#rays, #opaque) and the density gradient is
summed over ranks.  Here the oracle plays the kernels' role: each rank evaluates its own keyframes,
the counts and gradients go through DistContext, and the result must equal the single-process
evaluation of the whole window.
"""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _window(n_kf=4, n_rays=24, n_samples=32, seed=0):
    from oracle import poses as OP
    from oracle import rays as OR
    from loner_amd.utils import synthetic as SY
    dirs, _ = SY.lidar_pattern()
    base = SY.trajectory_pose6(n_kf)
    gen = torch.Generator().manual_seed(seed)
    scale, shift = SY.world_cube()
    out = []
    for k in range(n_kf):
        T = OP.transform_from_pose6(base[k])
        dist_k = SY.scene_ranges(dirs, T)
        idx = torch.randint(dirs.shape[1], (n_rays,), generator=gen)
        if k == 0:
            idx[3] = int((dist_k > 50).nonzero()[0])          # make sure transparent rays exist
        rays, depths, _ = OR.lidar_ray_records(dirs, dist_k, idx, T, torch.tensor([1.0, 50.0]), torch.tensor(scale), torch.from_numpy(shift))
        if k == 0:
            # make the reference's far[0] quirk observable: the FIRST ray of the whole batch gets a short `far` (as a ray clipped by
            # the cube wall has), so that many depths exceed far[0] - while the first rays of the other ranks keep the usual far
            rays[0, 12] = 0.25 * rays[0, 12] + 0.75 * rays[0, 11]
        z = torch.sort(torch.rand(rays.shape[0], n_samples, generator=gen) * (rays[:, 12:13] - rays[:, 11:12]) + rays[:, 11:12], dim=1).values
        noise = torch.randn(rays.shape[0], n_samples, generator=gen)
        out.append((rays.float(), depths.float(), z.float(), noise))
    return out, scale


def _rank_loss(spec, params, items, scale, counts_global, far0=None):
    """sum over this rank's keyframes of the loss terms re-normalised by the global counts."""
    from oracle import loss as OL
    from oracle import network as NW
    from oracle import render as ORD
    rays = torch.cat([i[0] for i in items]); depths = torch.cat([i[1] for i in items])
    z = torch.cat([i[2] for i in items]); noise = torch.cat([i[3] for i in items])
    sigma = NW.density(spec, params, ORD.sample_points(rays, z).reshape(-1, 3)).reshape(z.shape)
    out = ORD.composite(sigma, z, rays[:, 3:6], rays[:, 12:13], noise)
    loss, aux = OL.lidar_loss(out, z, rays, depths, torch.tensor(scale), OL.LossConfig(), far0=far0)
    n_local, op_local = rays.shape[0], int(aux["opaque"].sum())
    td, tl, to = aux["terms"]
    n_glob, op_glob = counts_global
    return td * (op_local / op_glob) + tl * (n_local / n_glob) + to * (op_local / op_glob), (n_local, op_local)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import loss as OL
    from oracle import network as NW
    from loner_amd.mapping.sharding import DistContext, front_record, shard_window
    ctx = DistContext()
    window, scale = _window()
    spec = NW.NetworkSpec.from_config(dict(otype="HashGrid", n_levels=4, n_features_per_level=2, log2_hashmap_size=10, base_resolution=4),
                                      dict(n_neurons=16, n_hidden_layers=1))
    params = NW.init_params(spec, 3)
    params[spec.n_mlp_params:] *= 3000
    mine = ctx.owned(window)
    assert [i for i, w in enumerate(window) if any(w is m for m in mine)] == shard_window(len(window), world, rank)
    # far[0] of the whole batch and the global counts: ONE all-gather of the ranks' front records (the kernels' lnr_shard_front_pack /
    # lnr_shard_front_reduce; here their plain-torch forms)
    rays = torch.cat([i[0] for i in mine]); depths = torch.cat([i[1] for i in mine])
    idx = ctx.owned_indices(len(window))
    seg_start = torch.tensor([0] + list(np.cumsum([i[0].shape[0] for i in mine])), dtype=torch.int32)
    cap = ctx.front_capacity(len(window), max(i[0].shape[0] for i in window))
    assert cap >= rays.shape[0]
    counts, far0 = ctx.gather_front(front_record(rays, seg_start, [2 * i for i in idx], depths, rays.shape[0], cap)).wait()
    far0 = far0[0]
    assert float(far0) == float(window[0][0][0, 12])                 # rank 0's first ray = first ray of the whole batch
    if rank != 0:
        assert float(rays[0, 12]) != float(far0)                     # a rank's own first ray would give a different mask
        assert int((depths > rays[0, 12]).sum()) != int((depths > far0).sum())
    all_d = torch.cat([i[1] for i in window])
    assert counts.tolist() == [all_d.shape[0], int(((all_d > 0) & ~(all_d > far0)).sum())]
    p = params.clone().requires_grad_(True)
    loss_r, _ = _rank_loss(spec, p, mine, scale, (int(counts[0]), int(counts[1])), far0=far0)
    loss_r.backward()
    grad = p.grad.clone()
    ctx.all_reduce_grads(grad)
    total = loss_r.detach().clone().reshape(1)
    dist.all_reduce(total)
    if rank == 0:
        ret["counts"] = counts.tolist(); ret["grad"] = grad.numpy(); ret["loss"] = float(total)
    # replicas stay identical: every rank applies the same reduced gradient
    gathered = [torch.zeros_like(grad) for _ in range(world)]
    dist.all_gather(gathered, grad)
    assert all(torch.equal(gathered[0], gr) for gr in gathered)
    dist.destroy_process_group()


def test_sharded_window_equals_single_process():
    from oracle import network as NW
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    window, scale = _window()
    spec = NW.NetworkSpec.from_config(dict(otype="HashGrid", n_levels=4, n_features_per_level=2, log2_hashmap_size=10, base_resolution=4),
                                      dict(n_neurons=16, n_hidden_layers=1))
    params = NW.init_params(spec, 3)
    params[spec.n_mlp_params:] *= 3000
    n = sum(i[0].shape[0] for i in window)
    p = params.clone().requires_grad_(True)
    # single process = one "rank" owning everything, interleaved in the same keyframe order as the reference
    loss, (n_all, op_all) = _rank_loss(spec, p, window, scale, (n, 1))     # provisional opaque count
    loss_single, _ = _rank_loss(spec, p, window, scale, (n_all, op_all))
    loss_single.backward()
    assert ret["counts"] == [n_all, op_all] and 0 < op_all < n_all
    assert abs(ret["loss"] - float(loss_single)) / abs(float(loss_single)) < 1e-5
    assert np.abs(ret["grad"] - p.grad.numpy()).max() / np.abs(p.grad.numpy()).max() < 1e-4


def _exchange_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from loner_amd.mapping.sharding import DistContext
    n_mlp, n_table = 48, 4096
    gen = torch.Generator().manual_seed(10 + rank)
    local = torch.randn(n_mlp + n_table, generator=gen)
    out = {}
    for exchange in ("all_reduce", "reduce_scatter"):
        for payload in ("fp32", "bf16"):
            ctx = DistContext(exchange=exchange, payload=payload)
            flat = local.clone()
            ctx.exchange_grads(flat, async_op=True).wait()
            sl = ctx.owned_range(flat.numel())
            assert (sl is None) == (exchange == "all_reduce")
            # a fake "step": every rank writes rank-independent values derived from the reduced gradient into the part it owns
            params = torch.zeros_like(flat)
            if sl is None:
                params.copy_(flat * 0.5)
            else:
                assert sl == (rank * (n_mlp + n_table) // world, (rank + 1) * (n_mlp + n_table) // world)   # equal chunks of the WHOLE vector
                params[sl[0]:sl[1]] = flat[sl[0]:sl[1]] * 0.5
                assert float(flat[:sl[0]].abs().sum()) == 0.0 and float(flat[sl[1]:].abs().sum()) == 0.0   # other ranks' chunks: zeroed
                ctx.gather_params(params)
                # zero_rest=False (the training loop's overwrite mode): the other chunks keep the rank's own contribution, the
                # rank's chunk still receives the sum
                keep = local.clone()
                ctx.exchange_grads(keep, async_op=True, zero_rest=False).wait()
                assert torch.equal(keep[sl[0]:sl[1]], flat[sl[0]:sl[1]])
                assert torch.equal(keep[:sl[0]], local[:sl[0]]) and torch.equal(keep[sl[1]:], local[sl[1]:])
                assert ctx.owned_range(n_mlp + n_table + 2) is None              # does not split into aligned equal chunks: all-reduce form
            out[(exchange, payload)] = params
    ret[rank] = {k: v.numpy() for k, v in out.items()}
    ret[f"local{rank}"] = local.numpy()
    dist.destroy_process_group()


def test_gradient_exchange_forms_agree():
    """all_reduce vs reduce_scatter(+all_gather), fp32 vs bf16 payload: every rank ends with the same full vector; the fp32
    forms equal the plain sum, the bf16 forms equal the sum of the bf16-rounded contributions (rounded to bf16)."""
    port = 29800 + (os.getpid() % 1000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_exchange_worker, args=(2, port, ret), nprocs=2, join=True)
    total = torch.from_numpy(ret["local0"]) + torch.from_numpy(ret["local1"])
    total_bf = (torch.from_numpy(ret["local0"]).bfloat16() + torch.from_numpy(ret["local1"]).bfloat16()).float()
    for key, v0 in ret[0].items():
        assert np.array_equal(v0, ret[1][key]), key                       # replicas identical
        exchange, payload = key
        want = total.clone()
        if payload == "bf16":
            want = total_bf.clone()
        assert np.array_equal(v0, (want * 0.5).numpy()), key


def test_shard_window_round_robin():
    from loner_amd.mapping.sharding import shard_window
    assert shard_window(8, 8, 3) == [3]
    assert shard_window(8, 2, 1) == [1, 3, 5, 7]
    assert shard_window(3, 4, 3) == []
    assert sorted(sum((shard_window(8, 4, r) for r in range(4)), [])) == list(range(8))


def _far0_window():
    window, _ = _window()
    for k, item in enumerate(window):            # give the first ray of every keyframe its own `far`: four distinguishable candidates
        item[0][0, 12] = item[0][0, 11] + (0.2 + 0.1 * k) * (item[0][0, 12] - item[0][0, 11])
    return window


def _far0_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from loner_amd.mapping.sharding import NO_RAY_KEY, DistContext, first_ray_key, front_record
    ctx = DistContext()
    window = _far0_window()
    cap = ctx.front_capacity(len(window), max(w[0].shape[0] for w in window))
    out, out_counts = {}, {}
    # which keyframes lost every candidate ray to the cube test (ray_utils.py:322): none / the first / both of rank 0 / all
    for name, empty in (("none", ()), ("kf0", (0,)), ("kf0_kf2", (0, 2)), ("kf0_kf1", (0, 1)), ("all", (0, 1, 2, 3))):
        idx = ctx.owned_indices(len(window))
        kept = [window[i][0] if i not in empty else window[i][0][:0] for i in idx]
        rays = torch.cat(kept)
        seg_start = torch.tensor([0] + list(np.cumsum([k.shape[0] for k in kept])), dtype=torch.int32)
        key = first_ray_key(rays, seg_start, [2 * i for i in idx])
        if all(i in empty for i in idx):
            assert int(key) == NO_RAY_KEY
        depths = torch.cat([window[i][1] if i not in empty else window[i][1][:0] for i in idx])
        rec = front_record(rays, seg_start, [2 * i for i in idx], depths, rays.shape[0], cap)
        assert rec.shape == (4 + cap,) and int(rec[0:2].view(torch.int64)) == int(key) and int(rec[2:3].view(torch.int32)) == rays.shape[0]
        counts, far0 = ctx.gather_front(rec).wait()
        assert far0.shape == (1,) and far0.dtype == torch.float32 and counts.dtype == torch.int32
        out[name] = float(far0)
        out_counts[name] = counts.tolist()
    # a rank that owns no keyframe at all joins with rays = None (Optimizer._join_without_rays)
    rec = front_record(None, None, (), None, 0, cap) if rank == 1 else \
        front_record(window[0][0], torch.tensor([0, window[0][0].shape[0]], dtype=torch.int32), [0], window[0][1], window[0][0].shape[0], cap)
    counts, far0 = ctx.gather_front(rec).wait()
    assert float(far0) == float(window[0][0][0, 12]) and int(counts[0]) == window[0][0].shape[0]
    # the failure word at the end of a sharded phase: the earliest failing iteration wins, code and iteration stay a pair
    word = lambda c, i: torch.tensor([c, i], dtype=torch.int32)
    assert ctx.earliest_failure(word(0, 0)).tolist() == [0, 0]
    assert ctx.earliest_failure(word(2, 7) if rank == 0 else word(1, 4)).tolist() == [1, 4]
    assert ctx.earliest_failure(word(0, 0) if rank == 0 else word(3, 9)).tolist() == [3, 9]
    assert ctx.earliest_failure(word(2, 5) if rank == 0 else word(1, 5)).tolist() == [1, 5]
    ret[rank] = out
    ret[f"counts{rank}"] = out_counts
    dist.destroy_process_group()


def test_far0_is_the_first_kept_ray_of_the_window_whoever_owns_it():
    """optimizer.py:460-461 compares every depth with far[0] of the WHOLE batch.  When the cube test drops every ray of the first
    keyframe(s), that ray belongs to a later keyframe - possibly another rank's: the smallest (window order | far bits) key among the
    all-gathered front records; the global counts are derived from the same records."""
    port = 30900 + (os.getpid() % 1000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_far0_worker, args=(2, port, ret), nprocs=2, join=True)
    window = _far0_window()
    first = lambda k: float(window[k][0][0, 12])
    assert len({first(k) for k in range(4)}) == 4                       # the four candidates are distinguishable
    for r in (0, 1):
        assert ret[r]["none"] == first(0) and ret[r]["kf0"] == first(1) and ret[r]["kf0_kf2"] == first(1) and ret[r]["kf0_kf1"] == first(2)
        assert np.isnan(ret[r]["all"])                                   # nobody has a ray: the value is never used
    # the global normalisers that came with it: #rays and #opaque (depth > 0 and not depth > far[0]) over the keyframes that kept rays
    assert ret["counts0"] == ret["counts1"]
    for name, empty in (("none", ()), ("kf0", (0,)), ("kf0_kf2", (0, 2)), ("kf0_kf1", (0, 1)), ("all", (0, 1, 2, 3))):
        d = torch.cat([window[k][1] for k in range(4) if k not in empty] + [torch.zeros(0)])
        far0 = ret[0][name]
        want = [int(d.shape[0]), int(((d > 0) & ~(d > far0)).sum()) if d.shape[0] else 0]
        assert ret["counts0"][name] == want, (name, ret["counts0"][name], want)
