"""Keyframe-window sharding over the GPUs of one node.

The reference has no multi-GPU mapping; rays of different keyframes are independent through
forward, loss sums and backward, pose gradients are private to a keyframe, so the window shards
by keyframe with the density-parameter gradient as the one exchanged quantity:

  * rank r owns keyframes {i : i mod G == r} of the <= 8-keyframe window;
  * every rank holds a full replica of the density parameters and the occupancy grid;
  * per iteration there is ONE small collective in front of the loss and ONE gradient exchange behind the backward:

    front    the loss divides by GLOBAL counts (#rays, #opaque rays: optimizer.py:488-489,569-570,577-578) and compares every
             ground-truth depth with `far` of the first ray of the WHOLE batch (the far[0] quirk, optimizer.py:460-461) - and "opaque"
             itself depends on far[0].  Every rank contributes one "front record" = {first-ray key = window order of its first kept
             ray | that ray's far, live-ray count, the ground-truth depths of its kept rays} (2 KB for 512 rays); the records are
             all-gathered and every rank derives far[0] and both counts from all depths locally (lnr_shard_front_reduce): one
             latency-bound collective, issued right after the ray compaction and awaited right before the loss kernel, so it hides
             behind the sampler and the density forward.  (Rounds 2-4 used a MIN all-reduce of the key followed by an all-reduce
             of the counts: two dependent collectives.)
    gradient "all_reduce": one all-reduce(sum) of the flat gradient [MLP matrices | tables] (29.7 MB); every rank runs the whole Adam
             step (237 MB of HBM traffic, ~25-35 us).  "reduce_scatter": the flat gradient is reduce-scattered in G equal contiguous
             chunks (rank r receives the sum of chunk r; the 3072 MLP weights simply lie in chunk 0), every rank runs Adam on ITS
             chunk only (parameters and Adam moments of a chunk live where it is stepped: 1/G of the Adam traffic per rank), then
             the stepped chunks are all-gathered - ONE reduce-scatter + ONE all-gather (rounds 2-4: a separate all-reduce for the MLP
             weights as well).  Same bytes on the wire as a ring all-reduce (it IS its two halves), but only the first half can hide
             behind the pose tail - the all-gather sits directly in front of the next density forward.  Worth it when the dense
             Adam step is a visible part of a rank's iteration (large tables, many ranks): the default from 4 ranks on.  Either form
             is issued asynchronously and awaited at the deferred density step, i.e. it overlaps the pose tail, the occupancy step
             and the next batch's ray build.
             payload="bf16": either form can put the gradient on the wire as bf16 (half the bytes; the sum over ranks is then
             rounded to 8 mantissa bits - replicas stay bit-identical because every rank receives the same sum).
  * every N_iters_acc-th step the occupancy-grid pseudo-gradient (V^3 64-bit fixed-point accumulators) is all-reduced the
    same way so that the samplers do not diverge.

Replicas stay bit-equal because every rank uses the reduced buffer as produced by the collective.

This module is backend-agnostic (it only calls torch.distributed; the two record helpers have a plain-torch form for CPU
tensors next to the HIP kernels), which is what lets the world_size-2 `gloo` tests exercise it on CPU.
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


FRONT_HEADER = 4                     # float32 words in front of a record's depths (include/loner_hip.h: LNR_FRONT_HEADER)


def front_record(rays, seg_start, seg_order: Sequence[int], depths, n_live, cap: int, device=None) -> torch.Tensor:
    """A rank's front record, float32 [FRONT_HEADER + cap]: words 0-1 the first-ray key (int64 bits), word 2 the live-ray count (int32
    bits), word 3 zero, then the ground-truth depths of its kept rays (zeros beyond the count).  rays None: a rank without keyframes.
    n_live: int or int32 tensor [1].  Plain torch ops on any device - on the MI355X the optimiser uses ops.shard_front_pack, one launch."""
    if rays is None:
        dev = torch.device(device if device is not None else "cpu")
        key = torch.full((1,), NO_RAY_KEY, dtype=torch.int64, device=dev)
        n = torch.zeros(1, dtype=torch.int32, device=dev)
        body = torch.zeros(int(cap), dtype=torch.float32, device=dev)
    else:
        dev = rays.device
        key = first_ray_key(rays, seg_start, seg_order)
        n = torch.as_tensor(n_live, dtype=torch.int32, device=dev).reshape(1).clamp(max=int(cap))
        d = depths.reshape(-1).float()[:int(cap)]
        body = torch.zeros(int(cap), dtype=torch.float32, device=dev)
        body[:d.shape[0]] = torch.where(torch.arange(d.shape[0], device=dev) < n, d, torch.zeros_like(d))
    return torch.cat([key.view(torch.float32), n.view(torch.float32), torch.zeros(1, dtype=torch.float32, device=dev), body])


def reduce_front_records(records: torch.Tensor, world: int, stride: int):
    """all-gathered front records [world * stride] -> (counts int32 [2] = {#rays, #opaque rays} of the whole batch, far0 float32 [1]):
    far[0] = the `far` under the smallest key; opaque = depth > 0 and not depth > far[0] (optimizer.py:460-463).  Plain torch ops -
    on the MI355X DistContext uses ops.shard_front_reduce, one launch."""
    recs = records.reshape(world, stride)
    keys = recs[:, 0:2].contiguous().view(torch.int64).reshape(world)
    kmin = keys.min().reshape(1)
    far0 = kmin.view(torch.float32)[0:1].clone()                      # little endian: the low word
    n = recs[:, 2].contiguous().view(torch.int32).clamp(0, stride - FRONT_HEADER)
    d = recs[:, FRONT_HEADER:]
    live = torch.arange(stride - FRONT_HEADER, device=records.device)[None, :] < n[:, None]
    opaque = live & (d > 0) & ~(d > far0)
    return torch.stack([n.sum(), opaque.sum()]).to(torch.int32), far0


def shard_window(n_keyframes: int, world_size: int, rank: int) -> List[int]:
    """Indices of the window's keyframes owned by `rank` (round-robin)."""
    return [i for i in range(n_keyframes) if i % world_size == rank]


class _Pending:
    """handle of an asynchronous collective: wait() blocks (the stream, for RCCL) and finishes the bookkeeping; its value is what
    `finish` returns"""

    def __init__(self, works, finish=None):
        self._works, self._finish, self._value = [w for w in works if w is not None], finish, None

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []
        if self._finish is not None:
            self._value = self._finish()
            self._finish = None
        return self._value


class NativeComm:
    """The HIP library's own RCCL communicator (csrc/lnr_comm.hip, include/loner_hip.h `lnr_comm_*`): a collective is ONE enqueue on a HIP
    stream - torch's current stream, in line with the kernels around it, or this object's side stream when the caller wants it beside
    other work (`side=True`: one event each way).  torch.distributed is used once, to hand rank 0's communicator id to the other ranks.
    ProcessGroupNCCL cost a one-keyframe rank ~30 us of host time per collective and two cross-queue waits per asynchronous one
    (profiles/r05_host_profile_sharded.txt)."""

    def __init__(self, group=None, device=None):
        import ctypes as C
        from .. import hip
        self._hip, self._C = hip, C
        lib = hip.load()
        if not lib.lnr_comm_available():
            raise RuntimeError("librccl.so.1 could not be loaded by the HIP library")
        self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # two communicators: one for the collectives issued in line on the compute stream, one for those on the side stream - RCCL
        # orders the operations of ONE communicator among themselves, so an in-line front all-gather would otherwise queue behind a
        # gradient exchange still in flight on the side stream.  Every rank issues both sequences in program order.
        ids = []
        for _ in range(2):
            buf = (C.c_char * hip.COMM_ID_BYTES)()
            if self.rank == 0:
                hip.check(lib.lnr_comm_unique_id(buf, hip.COMM_ID_BYTES), "lnr_comm_unique_id")
            ids.append(bytes(buf.raw))
        box = [ids if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._lib, self._comms = lib, []
        with torch.cuda.device(self.device):
            for blob in box[0]:
                comm = C.c_void_p()
                hip.check(lib.lnr_comm_init(blob, hip.COMM_ID_BYTES, self.rank, self.world_size, C.byref(comm)), "lnr_comm_init")
                self._comms.append(comm)
        self._side = None

    def close(self):
        if self._comms:
            torch.cuda.synchronize(self.device)
            for comm in self._comms:
                self._lib.lnr_comm_destroy(comm)
            self._comms = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- where a collective runs: torch's current stream, or the side stream behind an event of the current one
    def _enter(self, side):
        cur = torch.cuda.current_stream(self.device)
        if not side:
            return cur, None
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        ev = torch.cuda.Event()
        ev.record(cur)
        self._side.wait_event(ev)
        return self._side, cur

    def _leave(self, stream, cur, keep):
        """-> work with .wait(): makes the stream current at wait time wait for the collective (None when it ran in line)"""
        if cur is None:
            return None
        done = torch.cuda.Event()
        done.record(stream)
        return _NativeWork(done, self.device, keep)

    def all_reduce_(self, t, op="sum", side=False):
        st, cur = self._enter(side)
        self._hip.check(self._lib.lnr_comm_all_reduce(self._comms[1 if side else 0], self._hip._ptr(t), t.numel(), self._hip.COMM_DTYPES[t.dtype], self._hip.COMM_OPS[op],
                                                      self._C.c_void_p(st.cuda_stream)), "lnr_comm_all_reduce")
        return self._leave(st, cur, (t,))

    def reduce_scatter(self, out, src, side=False):
        st, cur = self._enter(side)
        self._hip.check(self._lib.lnr_comm_reduce_scatter(self._comms[1 if side else 0], self._hip._ptr(src), self._hip._ptr(out), out.numel(), self._hip.COMM_DTYPES[src.dtype],
                                                          self._C.c_void_p(st.cuda_stream)), "lnr_comm_reduce_scatter")
        return self._leave(st, cur, (out, src))

    def all_gather(self, out, mine, side=False):
        """out [world * mine.numel()] <- every rank's `mine` (which may be this rank's own slice of out)"""
        st, cur = self._enter(side)
        self._hip.check(self._lib.lnr_comm_all_gather(self._comms[1 if side else 0], self._hip._ptr(mine), self._hip._ptr(out), mine.numel() * mine.element_size(),
                                                      self._C.c_void_p(st.cuda_stream)), "lnr_comm_all_gather")
        return self._leave(st, cur, (out, mine))

    def broadcast_(self, t, src=0):
        st, _ = self._enter(False)
        self._hip.check(self._lib.lnr_comm_broadcast(self._comms[0], self._hip._ptr(t), t.numel() * t.element_size(), int(src),
                                                     self._C.c_void_p(st.cuda_stream)), "lnr_comm_broadcast")


class _NativeWork:
    def __init__(self, event, device, keep):
        self._event, self._device, self._keep = event, device, keep          # (the tensors stay referenced until the wait)

    def wait(self):
        torch.cuda.current_stream(self._device).wait_event(self._event)
        self._keep = None


class DistContext:
    def __init__(self, group=None, exchange: str = None, payload: str = "fp32", front: str = None, native=None):
        """exchange None: "reduce_scatter" from 4 ranks on, "all_reduce" below - with four or more ranks a rank's share of the window is
        one or two keyframes (an iteration of ~0.4 ms), of which the dense Adam step over all 7.4 M parameters is ~9 %; stepping a
        1/G chunk removes (G-1)/G of that, at the price of the all-gather in front of the next density forward."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if exchange is None:
            exchange = "reduce_scatter" if self.world_size >= 4 else "all_reduce"
        # native None: the HIP library's own RCCL binding whenever the process group is RCCL ("nccl") and every rank can bring its
        # communicators up - the ranks agree on that through the process group, and fall back TOGETHER to torch.distributed over the
        # same RCCL (a slower enqueue, the same collectives) with a warning: the binding has only ever run at RCCL world size 1
        # (DESIGN 5), and a rank that fails alone would leave the others waiting in ncclCommInitRank.  True: required (raises);
        # False: torch.distributed for everything (any backend: what the gloo tests on CPU exercise)
        self.native = None
        if native is None and dist.get_backend(group) == "nccl" and torch.cuda.is_available():
            from .. import hip
            ok, err = 0, None
            try:
                ok = 1 if hip.load().lnr_comm_available() else 0
                if not ok:
                    err = "librccl.so.1 could not be loaded by the HIP library"
            except Exception as e:                                        # (the library itself missing is an error elsewhere, loudly)
                err = str(e)
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag) == 1:
                try:
                    self.native = NativeComm(group)
                except Exception as e:
                    err = str(e)
                flag = torch.tensor([1 if self.native is not None else 0], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                if int(flag) == 0 and self.native is not None:
                    self.native.close()
                    self.native = None
            if self.native is None:
                import warnings
                warnings.warn(f"loner_amd sharding: the library's own RCCL binding is not available on every rank ({err or 'another rank failed'}); "
                              "collectives go through torch.distributed (ProcessGroupNCCL)")
            native = self.native is not None
        elif native is None:
            native = False
        # front None: "inline" with the library's own RCCL binding - one enqueue on the compute stream through a communicator of its own,
        # so it never queues behind the gradient exchange in flight on the side stream; "async" through torch.distributed, where a
        # synchronous collective shares ProcessGroupNCCL's one communicator with the asynchronous gradient exchange and would stall the
        # compute stream until that has finished (measured at world size 1 only: 0.380 / 0.396 ms per iteration in line / async)
        if front is None:
            front = "inline" if native else "async"
        if exchange not in ("all_reduce", "reduce_scatter") or payload not in ("fp32", "bf16") or front not in ("inline", "async"):
            raise ValueError(f"unknown gradient exchange {exchange!r} / payload {payload!r} / front {front!r}")
        self.exchange, self.payload, self.front = exchange, payload, front
        if native and self.native is None:
            self.native = NativeComm(group)

    # ---- the front of an iteration: far[0] and the loss normalisers -----------------------------------------------------
    def front_capacity(self, n_keyframes: int, rays_per_keyframe: int) -> int:
        """depth slots of a front record: the candidate rays of the rank that owns the most keyframes of an n_keyframes window (the same
        number on every rank: an all-gather wants equal contributions)"""
        return -(-int(n_keyframes) // self.world_size) * int(rays_per_keyframe)

    def gather_front(self, record: torch.Tensor) -> _Pending:
        """All-gather the ranks' front records (front_record / ops.shard_front_pack), asynchronously.  .wait() -> (counts int32 [2] =
        {#rays, #opaque rays} of the WHOLE batch, far0 float32 [1] = far of the batch's first ray), derived locally from the gathered
        depths; identical on every rank."""
        world, stride = self.world_size, record.numel()
        gathered = torch.empty(world * stride, dtype=torch.float32, device=record.device)
        # front "inline": a synchronous collective - ProcessGroupNCCL enqueues it on the CURRENT stream, between the pack kernel and
        # whatever follows, so its latency (a 16 KB all-gather) is in line but nothing crosses queues; "async": on the backend's own
        # stream, hidden behind the sampler and the density forward at the price of two cross-queue hand-overs
        if self.native is not None and record.is_cuda:
            work = self.native.all_gather(gathered, record, side=self.front == "async")
        else:
            work = dist.all_gather_into_tensor(gathered, record, group=self.group, async_op=self.front == "async")

        def finish():
            if gathered.is_cuda:
                from .. import ops
                return ops.shard_front_reduce(gathered, world, stride)
            return reduce_front_records(gathered, world, stride)
        return _Pending([work], finish)

    # ---- density gradient --------------------------------------------------------------------------------------------
    def owned_range(self, n_total: int):
        """(lo, hi) of this rank's chunk of the flat parameter vector in the "reduce_scatter" form, or None when the whole vector is
        all-reduced (form "all_reduce", or a vector that does not split into 16-byte aligned equal chunks)."""
        if self.exchange != "reduce_scatter" or n_total <= 0 or n_total % (4 * self.world_size):
            return None
        chunk = n_total // self.world_size
        return self.rank * chunk, (self.rank + 1) * chunk

    def exchange_grads(self, flat: torch.Tensor, async_op: bool = True, force_all_reduce: bool = False, zero_rest: bool = True):
        """Sum the flat density gradient [MLP | tables] over the ranks.  After .wait(): form "all_reduce" - `flat` holds the sum
        everywhere; form "reduce_scatter" - flat[lo:hi] (owned_range) holds the sum of this rank's chunk; the rest of the vector
        belongs to other ranks and is zeroed when zero_rest (a caller whose next backward STORES the whole gradient - the training
        loop's overwrite mode - passes False and saves two fills of the table per iteration).  force_all_reduce: the whole sum everywhere
        whatever the configured form - for callers that hand the gradient to an optimiser of their own (Optimizer.compute_loss through
        autograd): only the training loop knows how to step a chunk and gather the parameters afterwards."""
        bf16 = self.payload == "bf16"
        sl = None if force_all_reduce else self.owned_range(flat.numel())
        if sl is None:
            buf = flat.to(torch.bfloat16) if bf16 else flat
            if self.native is not None and flat.is_cuda:
                work = self.native.all_reduce_(buf, side=async_op)
            else:
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            pending = _Pending([work], (lambda: flat.copy_(buf)) if bf16 else None)
        else:
            lo, hi = sl
            src = flat.to(torch.bfloat16) if bf16 else flat
            out = torch.empty(hi - lo, device=flat.device, dtype=src.dtype)
            if self.native is not None and flat.is_cuda:
                work = self.native.reduce_scatter(out, src, side=async_op)
            else:
                work = dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

            def finish():
                if zero_rest:
                    flat[:lo].zero_()
                    flat[hi:].zero_()
                flat[lo:hi].copy_(out)
            pending = _Pending([work], finish)
        if not async_op:
            pending.wait()
        return pending

    def gather_params(self, flat_params: torch.Tensor):
        """form "reduce_scatter": every rank has stepped its chunk of the parameters; collect the chunks (in place)."""
        sl = self.owned_range(flat_params.numel())
        if sl is None:
            return
        if self.native is not None and flat_params.is_cuda:
            self.native.all_gather(flat_params, flat_params[sl[0]:sl[1]])          # in place: the rank's chunk is its slot of the result
            return
        mine = flat_params[sl[0]:sl[1]].clone()
        dist.all_gather_into_tensor(flat_params, mine, group=self.group)

    def owned(self, window: Sequence) -> list:
        return [window[i] for i in shard_window(len(window), self.world_size, self.rank)]

    def all_reduce_grads(self, flat: torch.Tensor, async_op: bool = False):
        """Sum a flat buffer over ranks, in place (the occupancy pseudo-gradient: 64-bit fixed-point accumulators, exact)."""
        if self.native is not None and flat.is_cuda and flat.dtype in (torch.float32, torch.int64, torch.int32):
            work = self.native.all_reduce_(flat, side=async_op)
            return work if async_op else flat
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return work if async_op else flat

    def earliest_failure(self, poison: torch.Tensor) -> torch.Tensor:
        """The failure guard's {code, iteration} word (int32 [2], code 0 = none) of every rank -> the word of the EARLIEST failing
        iteration over all ranks (ties: the smallest code), as one packed MIN all-reduce so that the pair stays a pair."""
        none = torch.iinfo(torch.int64).max
        packed = torch.where(poison[0:1] != 0, poison[1:2].long() * 256 + poison[0:1].long(), torch.full((1,), none, dtype=torch.int64, device=poison.device))
        if self.native is not None and packed.is_cuda:
            self.native.all_reduce_(packed, op="min")
        else:
            dist.all_reduce(packed, op=dist.ReduceOp.MIN, group=self.group)
        failed = packed != none
        code = torch.where(failed, packed % 256, torch.zeros_like(packed))
        it = torch.where(failed, packed // 256, torch.zeros_like(packed))
        return torch.cat([code, it]).to(torch.int32)

    def owned_indices(self, n_keyframes: int) -> List[int]:
        return shard_window(n_keyframes, self.world_size, self.rank)

    def max_over_ranks(self, value: int) -> int:
        """the largest of the ranks' integers (a host value on every rank: one blocking all-reduce; used off the training loop only)"""
        t = torch.tensor([int(value)], dtype=torch.int64)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        dist.broadcast(t, src=src, group=self.group)
        return t
