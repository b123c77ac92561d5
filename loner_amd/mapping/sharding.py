"""Keyframe-window sharding over the GPUs of one node.

The reference has no multi-GPU mapping; rays of different keyframes are independent through
forward, loss sums and backward, pose gradients are private to a keyframe, so the window shards
by keyframe with ONE exchange step per iteration:

  * rank r owns keyframes {i : i mod G == r} of the <= 8-keyframe window;
  * every rank holds a full replica of the density parameters, Adam state and occupancy grid;
  * before the loss is scaled, the two normalisers (#rays, #opaque rays) are all-reduced (2 ints),
    because the reference divides by GLOBAL counts (optimizer.py:488-489,569-570,577-578);
  * after backward, the density-parameter gradient is all-reduced (sum) - RCCL over xGMI via
    torch.distributed backend "nccl"; every rank then applies the identical Adam step, so replicas
    stay bit-equal (the reduced buffer is used as produced by the collective on every rank);
  * the loss compares every ground-truth depth with `far` of the first ray of the batch (the far[0] quirk,
    optimizer.py:460-461): rank 0 broadcasts that float, so the sharded loss equals the single-GPU one;
  * every N_iters_acc-th step the occupancy-grid pseudo-gradient (V^3 floats) is all-reduced the
    same way so that the samplers do not diverge.

This module is backend-agnostic (it only calls torch.distributed), which is what lets the
world_size-2 `gloo` tests exercise it on CPU.
"""
from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_window(n_keyframes: int, world_size: int, rank: int) -> List[int]:
    """Indices of the window's keyframes owned by `rank` (round-robin)."""
    return [i for i in range(n_keyframes) if i % world_size == rank]


class DistContext:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def owned(self, window: Sequence) -> list:
        return [window[i] for i in shard_window(len(window), self.world_size, self.rank)]

    def all_reduce_counts(self, counts: torch.Tensor, async_op: bool = False):
        """counts int32 [2] = {#rays, #opaque} of this rank -> global, in place.  With async_op=True the collective's
        handle is returned instead and the caller `.wait()`s right before the first use: the tiny all-reduce is pure
        latency and hides behind the sampler and the density forward."""
        work = dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return work if async_op else counts

    def all_reduce_grads(self, flat: torch.Tensor, async_op: bool = False):
        """Sum a flat gradient buffer over ranks, in place (one large collective, not per-tensor buckets:
        the whole density gradient is a single 29.7 MB vector).  async_op=True returns the handle; the training loop
        waits for it right before the density Adam step, so the pose gradient of this rank runs next to the collective."""
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return work if async_op else flat

    def broadcast_far0(self, rays, device=None) -> torch.Tensor:
        """The reference's `depth > far[0]` test (optimizer.py:460-461) uses the first ray of the whole batch.  Rank 0 owns the
        first active keyframe, hence that ray: it sends `rays[0, 12]`, everyone gets a device float [1] to hand to
        lnr_count_opaque / lnr_los_loss_fused.  `rays` may be None on ranks without rays."""
        if self.rank == 0:
            far0 = rays[0:1, 12].clone()
        else:
            far0 = torch.zeros(1, device=rays.device if rays is not None else device, dtype=torch.float32)
        dist.broadcast(far0, src=0, group=self.group)
        return far0

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        dist.broadcast(t, src=src, group=self.group)
        return t
