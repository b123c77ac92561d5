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

Two forms of the gradient exchange (DistContext(exchange=...)):

  "all_reduce"      one all-reduce(sum) of the flat gradient [MLP matrices | tables] (29.7 MB); every rank runs the whole
                    Adam step (237 MB of HBM traffic, ~25-35 us).  Issued asynchronously and awaited at the deferred
                    density step, i.e. it overlaps the pose tail, the occupancy step and the next batch's ray build.
  "reduce_scatter"  the table gradient is reduce-scattered by contiguous slice (rank r receives the sum of slice r), every
                    rank runs Adam on ITS slice of the table only (parameters and Adam moments of a slice live where the
                    slice is stepped: 1/G of the Adam traffic per rank), then the stepped parameter slices are all-gathered;
                    the 3072 MLP weights are all-reduced separately and stepped everywhere.  Same bytes on the wire as a
                    ring all-reduce (it IS its two halves), but only the first half can hide behind the pose tail - the
                    all-gather sits directly in front of the next density forward.  Worth it when the dense Adam step is
                    a visible part of a rank's iteration (large tables, many ranks); the default stays "all_reduce".
  payload="bf16"    either form can put the gradient on the wire as bf16 (half the bytes; the sum over ranks is then
                    rounded to 8 mantissa bits - replicas stay bit-identical because every rank receives the same sum).

This module is backend-agnostic (it only calls torch.distributed), which is what lets the
world_size-2 `gloo` tests exercise it on CPU.
"""
from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_window(n_keyframes: int, world_size: int, rank: int) -> List[int]:
    """Indices of the window's keyframes owned by `rank` (round-robin)."""
    return [i for i in range(n_keyframes) if i % world_size == rank]


class _Pending:
    """handle of an asynchronous gradient exchange: wait() blocks (the stream, for RCCL) and finishes the bookkeeping"""

    def __init__(self, works, finish=None):
        self._works, self._finish = [w for w in works if w is not None], finish

    def wait(self):
        for w in self._works:
            w.wait()
        if self._finish is not None:
            self._finish()
            self._finish = None


class DistContext:
    def __init__(self, group=None, exchange: str = "all_reduce", payload: str = "fp32"):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if exchange not in ("all_reduce", "reduce_scatter") or payload not in ("fp32", "bf16"):
            raise ValueError(f"unknown gradient exchange {exchange!r} / payload {payload!r}")
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.exchange, self.payload = exchange, payload

    # ---- density gradient --------------------------------------------------------------------------------------------
    def table_slice(self, n_mlp: int, n_total: int):
        """(lo, hi) of this rank's slice of the flat parameter vector in the "reduce_scatter" form, or None when the whole
        vector is all-reduced (form "all_reduce", or a table that does not split into 16-byte aligned equal slices)."""
        n_table = n_total - n_mlp
        if self.exchange != "reduce_scatter" or n_table <= 0 or n_table % (4 * self.world_size) or n_mlp % 4:
            return None
        chunk = n_table // self.world_size
        return n_mlp + self.rank * chunk, n_mlp + (self.rank + 1) * chunk

    def exchange_grads(self, flat: torch.Tensor, n_mlp: int, async_op: bool = True, force_all_reduce: bool = False):
        """Sum the flat density gradient [MLP | tables] over the ranks.  After .wait(): form "all_reduce" - `flat` holds the sum
        everywhere; form "reduce_scatter" - flat[:n_mlp] and flat[lo:hi] (table_slice) hold the sums, the rest of the table
        gradient is zeroed (it belongs to other ranks).  force_all_reduce: the whole sum everywhere whatever the configured
        form - for callers that hand the gradient to an optimiser of their own (Optimizer.compute_loss through autograd): only
        the training loop knows how to step a slice and gather the parameters afterwards."""
        bf16 = self.payload == "bf16"
        sl = None if force_all_reduce else self.table_slice(n_mlp, flat.numel())
        if sl is None:
            buf = flat.to(torch.bfloat16) if bf16 else flat
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            pending = _Pending([work], (lambda: flat.copy_(buf)) if bf16 else None)
        else:
            lo, hi = sl
            w_mlp = dist.all_reduce(flat[:n_mlp], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            table = flat[n_mlp:]
            src = table.to(torch.bfloat16) if bf16 else table
            out = torch.empty(hi - lo, device=flat.device, dtype=src.dtype)
            w_tab = dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

            def finish():
                flat[n_mlp:lo].zero_()
                flat[hi:].zero_()
                flat[lo:hi].copy_(out)
            pending = _Pending([w_mlp, w_tab], finish)
        if not async_op:
            pending.wait()
        return pending

    def gather_params(self, flat_params: torch.Tensor, n_mlp: int):
        """form "reduce_scatter": every rank has stepped its slice of the table; collect the slices (in place)."""
        sl = self.table_slice(n_mlp, flat_params.numel())
        if sl is None:
            return
        mine = flat_params[sl[0]:sl[1]].clone()
        dist.all_gather_into_tensor(flat_params[n_mlp:], mine, group=self.group)

    def owned(self, window: Sequence) -> list:
        return [window[i] for i in shard_window(len(window), self.world_size, self.rank)]

    def all_reduce_counts(self, counts: torch.Tensor, async_op: bool = False):
        """counts int32 [2] = {#rays, #opaque} of this rank -> global, in place.  With async_op=True the collective's
        handle is returned instead and the caller `.wait()`s right before the first use: the tiny all-reduce is pure
        latency and hides behind the sampler and the density forward."""
        work = dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return work if async_op else counts

    def all_reduce_grads(self, flat: torch.Tensor, async_op: bool = False):
        """Sum a flat buffer over ranks, in place (the occupancy pseudo-gradient: 64-bit fixed-point accumulators, exact)."""
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return work if async_op else flat

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        """element-wise maximum over the ranks, in place (the failure guard's {code, iteration} word at the end of a phase)"""
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def broadcast_far0(self, rays, device=None) -> torch.Tensor:
        """The reference's `depth > far[0]` test (optimizer.py:460-461) uses the first ray of the whole batch.  Rank 0 owns the
        first active keyframe, hence that ray: it sends `rays[0, 12]`, everyone gets a device float [1] to hand to
        lnr_count_opaque / lnr_los_loss_fused.  `rays` may be None on ranks without rays.
        Limitation (documented): if the cube test drops EVERY candidate ray of the first active keyframe,
        the single-GPU batch starts with a later keyframe's ray while rank 0 still sends the first row of its own compacted
        batch (its next keyframe's first ray, or a dead row if it has none left); finding the true first ray would cost a
        second small collective and several device-side index operations per iteration for a case that needs a whole
        keyframe of 512 rays to miss the world cube."""
        if self.rank == 0:
            far0 = rays[0:1, 12].clone()
        else:
            far0 = torch.zeros(1, device=rays.device if rays is not None else device, dtype=torch.float32)
        dist.broadcast(far0, src=0, group=self.group)
        return far0

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        dist.broadcast(t, src=src, group=self.group)
        return t
