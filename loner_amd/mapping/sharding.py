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
    optimizer.py:460-461): the ranks agree on that ray with one 8-byte MIN all-reduce (window order of each rank's
    first kept ray | its far), so the sharded loss equals the single-GPU one also when a keyframe lost all its rays;
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
                    a visible part of a rank's iteration (large tables, many ranks): the default from 4 ranks on.
  payload="bf16"    either form can put the gradient on the wire as bf16 (half the bytes; the sum over ranks is then
                    rounded to 8 mantissa bits - replicas stay bit-identical because every rank receives the same sum).

This module is backend-agnostic (it only calls torch.distributed), which is what lets the
world_size-2 `gloo` tests exercise it on CPU.
"""
from typing import List, Sequence

import torch
import torch.distributed as dist


NO_RAY_KEY = 0x7FFFFFFFFFFFFFFF      # first_ray_key of a rank without a kept ray (its low word read as a float is NaN: no ray uses it)


def first_ray_key(rays: torch.Tensor, seg_start: torch.Tensor, seg_order: Sequence[int]) -> torch.Tensor:
    """int64 [1]: (seg_order of the first segment with a kept ray) << 32 | float bits of that ray's far, NO_RAY_KEY without one.
    rays [n,13], seg_start int32 [n_seg+1] (compacted segment starts), seg_order ascending window positions of the segments.
    Plain torch ops (any device, no host sync) - on the MI355X the optimiser uses ops.first_ray_key, one launch."""
    order = torch.as_tensor(list(seg_order), dtype=torch.int64, device=rays.device)
    lo, hi = seg_start[:-1].long(), seg_start[1:].long()
    live = hi > lo
    cand = torch.where(live, order, torch.full_like(order, 1 << 31))
    o, s = cand.min(0)
    row = lo[s].clamp(max=max(rays.shape[0] - 1, 0))
    far = rays[row, 12] if rays.shape[0] else torch.zeros((), device=rays.device)
    bits = far.reshape(1).contiguous().view(torch.int32).long() & 0xFFFFFFFF
    key = (o.reshape(1) << 32) | bits
    return torch.where(live.any().reshape(1), key, torch.full_like(key, NO_RAY_KEY))


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
    def __init__(self, group=None, exchange: str = None, payload: str = "fp32"):
        """exchange None: "reduce_scatter" from 4 ranks on, "all_reduce" below - with four or more ranks a rank's share of the window is
        one or two keyframes (an iteration of ~0.4 ms), of which the dense Adam step over all 7.4 M parameters is ~9 %; stepping a
        1/G slice of the tables removes (G-1)/G of that, at the price of the all-gather in front of the next density forward."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if exchange is None:
            exchange = "reduce_scatter" if self.world_size >= 4 else "all_reduce"
        if exchange not in ("all_reduce", "reduce_scatter") or payload not in ("fp32", "bf16"):
            raise ValueError(f"unknown gradient exchange {exchange!r} / payload {payload!r}")
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

    def earliest_failure(self, poison: torch.Tensor) -> torch.Tensor:
        """The failure guard's {code, iteration} word (int32 [2], code 0 = none) of every rank -> the word of the EARLIEST failing
        iteration over all ranks (ties: the smallest code), as one packed MIN all-reduce so that the pair stays a pair."""
        none = torch.iinfo(torch.int64).max
        packed = torch.where(poison[0:1] != 0, poison[1:2].long() * 256 + poison[0:1].long(), torch.full((1,), none, dtype=torch.int64, device=poison.device))
        dist.all_reduce(packed, op=dist.ReduceOp.MIN, group=self.group)
        failed = packed != none
        code = torch.where(failed, packed % 256, torch.zeros_like(packed))
        it = torch.where(failed, packed // 256, torch.zeros_like(packed))
        return torch.cat([code, it]).to(torch.int32)

    def owned_indices(self, n_keyframes: int) -> List[int]:
        return shard_window(n_keyframes, self.world_size, self.rank)

    def broadcast_far0(self, rays, device=None, first_key=None) -> torch.Tensor:
        """The reference's `depth > far[0]` test (optimizer.py:460-461) uses the first ray of the whole batch: the first kept ray of the
        first keyframe, in window order, that kept any (the cube test may drop every candidate of a keyframe, ray_utils.py:322).
        Every rank contributes `first_key` (first_ray_key below / ops.first_ray_key: window order << 32 | bits of its own first ray's
        far, INT64_MAX without a ray) and one MIN all-reduce leaves the key of the batch's first ray everywhere; its low word is far[0],
        returned as a device float [1] for lnr_count_opaque / lnr_los_loss_fused.  Without first_key the rank's position stands in for
        the window order and `rays` (None: no ray) is taken as a batch of kept rays - right whenever rank order is keyframe order."""
        if first_key is None:
            if rays is None or rays.shape[0] == 0:
                first_key = torch.full((1,), NO_RAY_KEY, dtype=torch.int64, device=rays.device if rays is not None else device)
            else:
                first_key = first_ray_key(rays, torch.tensor([0, rays.shape[0]], device=rays.device, dtype=torch.int32), [self.rank])
        dist.all_reduce(first_key, op=dist.ReduceOp.MIN, group=self.group)
        return first_key.view(torch.float32)[0:1]          # little endian: the low word

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        dist.broadcast(t, src=src, group=self.group)
        return t
