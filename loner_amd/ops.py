"""Tensor-level wrappers around the C ABI (one Python function per entry point).

Every function takes/returns torch tensors that live on the HIP device, allocates outputs with
torch, and enqueues the kernel on torch's current stream.  No function here computes anything
itself - the arithmetic is in libloner_hip.so.
"""
import ctypes as C
import os

import torch

from . import hip
from .hip import _ptr, _stream, check, load, require_device

_steps_cache = {}


def linspace_table(count: int, device) -> torch.Tensor:
    """torch.linspace(0,1,count) on `device` (ray_sampling.py:27,59); cached."""
    key = (count, str(device))
    t = _steps_cache.get(key)
    if t is None:
        t = torch.linspace(0, 1, count).to(device)
        _steps_cache[key] = t
    return t


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# ---------------------------------------------------------------- diagnostics
def profile_enable(on=True):
    """Per-kernel HIP-event timing inside lnr_density_forward / lnr_density_backward (off by default)."""
    check(load().lnr_profile_enable(1 if on else 0), "lnr_profile_enable")


def profile_read():
    """-> {kernel: {"calls": n, "total_ms": t, "avg_ms": t / n}}; waits for the recorded events and clears the log."""
    cap, stride = 64, 48
    names = C.create_string_buffer(cap * stride)
    ms = (C.c_float * cap)()
    calls = (C.c_int32 * cap)()
    n = load().lnr_profile_read(names, stride, ms, calls, cap)
    out = {}
    for i in range(max(n, 0)):
        name = names.raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode()
        out[name] = {"calls": int(calls[i]), "total_ms": float(ms[i]), "avg_ms": float(ms[i]) / max(int(calls[i]), 1)}
    return out


# ---------------------------------------------------------------- density network
_workspaces = {}
_BINS_W8 = bool(os.environ.get("LNR_BINS_W8"))
_BINS = bool(os.environ.get("LNR_BINS"))            # A/B switch of the experiment scripts (tools/): the binned partition on hashed levels


def _workspace(spec, device, n_points, forward_only=False):
    """One scratch buffer per device, grown on demand (feature planes, gradient records, ... - see the header).
    forward_only: sized for lnr_density_forward alone (no backward will follow on these points)."""
    lib = load()
    need = (lib.lnr_density_workspace_forward if forward_only else lib.lnr_density_workspace)(C.byref(spec), int(n_points))
    key = str(device)
    ent = _workspaces.get(key)
    if ent is None or ent["buf"].numel() * 4 < need:
        clipped = 0 if ent is None else ent["clipped_before"] + int(ent["buf"][:1].view(torch.int32).item())
        if ent is not None:                         # the library's note about the old buffer goes with it (include/loner_hip.h)
            check(load().lnr_density_workspace_release(_ptr(ent["buf"])), "lnr_density_workspace_release")
        _workspaces.pop(key, None)
        ent = None                                  # release the old buffer before the larger one is allocated
        ent = {"buf": torch.empty((need + 3) // 4, device=device, dtype=torch.float32), "features_of": None, "clipped_before": clipped}
        check(load().lnr_density_workspace_init(_ptr(ent["buf"]), ent["buf"].numel() * 4, _stream()), "lnr_density_workspace_init")
        _workspaces[key] = ent
    return ent, need


def density_clipped_count(device) -> int:
    """Number of density outputs the forward kernels have clipped on `device` so far (non-finite values, and values beyond
    +-65504 in the fp16 mode: nerf_tcnn.py:70-78).  Reads a status word of the workspace: synchronises with the device."""
    ent = _workspaces.get(str(device))
    if ent is None:
        return 0
    return ent["clipped_before"] + int(ent["buf"][hip.STATUS_CLIPPED:hip.STATUS_CLIPPED + 1].view(torch.int32).item())


def density_forward(spec, params, pts=None, rays=None, z=None, n_rays_dev=None, forward_only=False):
    """forward_only: the caller will not call density_backward(reuse_features=True) on these points - the workspace is sized
    for the forward alone (rendering / inference: no record regions)."""
    require_device(params, pts, rays, z)
    params = _f32c(params)
    if pts is not None:
        pts = _f32c(pts).reshape(-1, 3)
        n = pts.shape[0]
        ent, need = _workspace(spec, params.device, n, forward_only)
        sigma = torch.empty(n, device=params.device, dtype=torch.float32)
        check(load().lnr_density_forward(C.byref(spec), _ptr(params), _ptr(pts), n, None, None, 0, 0, None,
                                         _ptr(sigma), _ptr(ent["buf"]), need, _stream()), "lnr_density_forward")
        ent["features_of"] = None if forward_only else (params.data_ptr(), pts.data_ptr(), 0, n)
        return sigma
    rays, z = _f32c(rays), _f32c(z)
    n, s = z.shape
    ent, need = _workspace(spec, params.device, n * s, forward_only)
    sigma = torch.empty(n, s, device=params.device, dtype=torch.float32)      # rows >= *n_rays_dev are never read
    check(load().lnr_density_forward(C.byref(spec), _ptr(params), None, 0, _ptr(rays), _ptr(z), n, s,
                                     _ptr(n_rays_dev), _ptr(sigma), _ptr(ent["buf"]), need, _stream()), "lnr_density_forward")
    ent["features_of"] = None if forward_only else (params.data_ptr(), rays.data_ptr(), z.data_ptr(), n * s)
    return sigma


def _event(ev):
    """raw hipEvent_t of a torch.cuda.Event that has been recorded at least once (torch creates the handle lazily), or NULL"""
    if ev is None:
        return None
    h = ev.cuda_event
    if not h:
        raise RuntimeError("loner_amd: the torch event has no handle yet - record it once before handing it to the library")
    return C.c_void_p(h)


def density_backward(spec, params, d_sigma, grad_params, pts=None, rays=None, z=None, n_rays_dev=None,
                     want_d_pts=False, reuse_features=False, d_rays=None, table_atomics=False, report_regions=False, bins=False, bins_w8=False, input_grad_event=None,
                     defer_weight_fold=False, overwrite_grad=False):
    """Accumulates into grad_params [n_params] (None: parameters frozen, only the input gradient is computed);
    returns d_pts ([...,3]) or None.  table_atomics: test hook (LNR_BWD_TABLE_ATOMICS).
    reuse_features: the caller asserts that the last density_forward on this device ran on the same params and
    points (and that neither changed since), so the encoded features still in the workspace are reused.
    d_rays [n_rays,13] (rays form, instead of want_d_pts): the point gradient is reduced per ray and added to it.
    overwrite_grad: grad_params receives this call's gradient instead of accumulating it (LNR_BWD_OVERWRITE_GRAD)."""
    assert not (want_d_pts and d_rays is not None)
    require_device(params, d_sigma, grad_params, pts, rays, z)
    params, d_sigma = _f32c(params), _f32c(d_sigma)
    assert grad_params is None or (grad_params.dtype == torch.float32 and grad_params.is_contiguous())
    flags = (hip.BWD_TABLE_ATOMICS if table_atomics else 0) | (hip.BWD_REPORT_REGIONS if report_regions else 0) | \
        (hip.BWD_BINS if (bins or _BINS) else 0) | (hip.BWD_BINS_W8 if (bins_w8 or _BINS_W8) else 0) | \
        (hip.BWD_DEFER_WEIGHT_FOLD if defer_weight_fold else 0) | (hip.BWD_OVERWRITE_GRAD if overwrite_grad else 0)
    n_points = (pts.numel() // 3) if pts is not None else z.numel()
    ent, need = _workspace(spec, params.device, n_points)
    if pts is not None:
        pts = _f32c(pts).reshape(-1, 3)
        n = pts.shape[0]
        reuse = int(bool(reuse_features))
        if reuse and ent["features_of"] != (params.data_ptr(), pts.data_ptr(), 0, n):
            raise RuntimeError("density_backward(reuse_features=True): workspace features belong to a different forward call")
        d_pts = torch.empty(n, 3, device=params.device, dtype=torch.float32) if want_d_pts else None
        check(load().lnr_density_backward(C.byref(spec), _ptr(params), _ptr(pts), n, None, None, 0, 0, None,
                                          _ptr(d_sigma), _ptr(grad_params), _ptr(d_pts), None, reuse, flags, _ptr(ent["buf"]), need,
                                          _event(input_grad_event), _stream()), "lnr_density_backward")
        return d_pts
    rays, z = _f32c(rays), _f32c(z)
    n, s = z.shape
    reuse = int(bool(reuse_features))
    if reuse and ent["features_of"] != (params.data_ptr(), rays.data_ptr(), z.data_ptr(), n * s):
        raise RuntimeError("density_backward(reuse_features=True): workspace features belong to a different forward call")
    d_pts = torch.empty(n, s, 3, device=params.device, dtype=torch.float32) if want_d_pts else None
    check(load().lnr_density_backward(C.byref(spec), _ptr(params), None, 0, _ptr(rays), _ptr(z), n, s,
                                      _ptr(n_rays_dev), _ptr(d_sigma), _ptr(grad_params), _ptr(d_pts), _ptr(d_rays), reuse, flags,
                                      _ptr(ent["buf"]), need, _event(input_grad_event), _stream()), "lnr_density_backward")
    return d_pts


def density_fold_weight_grads(spec, grad_params, n_points, overwrite_grad=False):
    """Adds the weight-gradient slabs a density_backward(defer_weight_fold=True) call left in the workspace to grad_params
    (on the current stream: the training loop does it on its side stream, beside the table-gradient reduce)."""
    require_device(grad_params)
    ent, need = _workspace(spec, grad_params.device, n_points)
    check(load().lnr_density_fold_weight_grads(C.byref(spec), int(n_points), _ptr(grad_params), _ptr(ent["buf"]), need,
                                               hip.BWD_OVERWRITE_GRAD if overwrite_grad else 0, _stream()),
          "lnr_density_fold_weight_grads")


# ---------------------------------------------------------------- rays
def build_lidar_rays(directions, distances, index, transform12, ray_range, scale, shift):
    """-> (rays [m,13], depths [m], keep [m] uint8) for ALL candidates."""
    require_device(directions, distances, index, transform12)
    directions, distances, transform12 = _f32c(directions), _f32c(distances), _f32c(transform12)
    index = index.to(torch.int64).contiguous()
    m = index.shape[0]
    dev = directions.device
    rays = torch.empty(m, hip.RAY_STRIDE, device=dev, dtype=torch.float32)
    depths = torch.empty(m, device=dev, dtype=torch.float32)
    keep = torch.empty(m, device=dev, dtype=torch.uint8)
    sh = (C.c_float * 3)(float(shift[0]), float(shift[1]), float(shift[2]))
    check(load().lnr_build_lidar_rays(_ptr(directions), _ptr(distances), directions.shape[1], _ptr(index), m,
                                      _ptr(transform12), float(ray_range[0]), float(ray_range[1]), float(scale), sh,
                                      _ptr(rays), _ptr(depths), _ptr(keep), _stream()), "lnr_build_lidar_rays")
    return rays, depths, keep


class WindowTables:
    """Host-side per-segment tables of a keyframe window for lnr_build_window_rays (built once per window)."""

    def __init__(self, dirs_list, dist_list, const_dist, seg_counts, seg_pose):
        n = len(dirs_list)
        self.n_seg = n
        self.keepalive = (list(dirs_list), list(dist_list))
        self.dirs = (C.c_void_p * n)(*[d.data_ptr() for d in dirs_list])
        self.dist = (C.c_void_p * n)(*[(d.data_ptr() if d is not None else None) for d in dist_list])
        self.const_dist = (C.c_float * n)(*[float(c) for c in const_dist])
        self.n_points = (C.c_int64 * n)(*[int(d.shape[1]) for d in dirs_list])
        starts = [0]
        for c in seg_counts:
            starts.append(starts[-1] + int(c))
        self.seg_start_list = starts
        self.seg_start = (C.c_int32 * (n + 1))(*starts)
        self.seg_pose = (C.c_int32 * n)(*[int(p) for p in seg_pose])
        self.total = starts[-1]


def build_window_rays(tab: WindowTables, transforms, ray_range, scale, shift, index=None, seed=0):
    """-> (rays [total,13], depths, keep, src_index) for every candidate of the window, one launch."""
    require_device(transforms, index)
    dev = transforms.device
    rays = torch.empty(tab.total, hip.RAY_STRIDE, device=dev, dtype=torch.float32)
    depths = torch.empty(tab.total, device=dev, dtype=torch.float32)
    keep = torch.empty(tab.total, device=dev, dtype=torch.uint8)
    idx_out = None
    if index is None:
        idx_out = torch.empty(tab.total, device=dev, dtype=torch.int64)
    else:
        index = index.to(torch.int64).contiguous()
    sh = (C.c_float * 3)(float(shift[0]), float(shift[1]), float(shift[2]))
    check(load().lnr_build_window_rays(tab.dirs, tab.dist, tab.const_dist, tab.n_points, tab.seg_start, tab.seg_pose, tab.n_seg,
                                       _ptr(index), _ptr(idx_out), int(seed) & (2 ** 64 - 1), _ptr(_f32c(transforms)),
                                       float(ray_range[0]), float(ray_range[1]), float(scale), sh, _ptr(rays), _ptr(depths),
                                       _ptr(keep), _stream()), "lnr_build_window_rays")
    return rays, depths, keep, (index if index is not None else idx_out)


def pose_forward(pose6):
    require_device(pose6)
    p = _f32c(pose6.detach()).reshape(-1, 6)
    T = torch.empty(p.shape[0], 12, device=p.device, dtype=torch.float32)
    check(load().lnr_pose_forward(_ptr(p), p.shape[0], _ptr(T), _stream()), "lnr_pose_forward")
    return T


def pose_backward(pose6, d_transforms, mask=None, out=None, accumulate=False, poison=None, poison_tag=0):
    """poison (int32[2] device word, optional): failure guard, see include/loner_hip.h."""
    require_device(pose6, d_transforms, mask, out, poison)
    p = _f32c(pose6.detach()).reshape(-1, 6)
    if out is None:
        out = torch.empty_like(p)
        accumulate = False
    check(load().lnr_pose_backward(_ptr(p), _ptr(_f32c(d_transforms)), _ptr(mask), p.shape[0], _ptr(out), int(accumulate),
                                   _ptr(poison), int(poison_tag), _stream()), "lnr_pose_backward")
    return out


def compact_rays(rays, depths, keep, src_index, seg_start, n_out=None, want_counts=False, front=None):
    """seg_start: python list [n_seg+1] (or the ctypes int32 array of a WindowTables).  -> (rays_out [cap,13], depths_out, src_out, out_seg_start dev int32
    [n_seg+1], n_out dev int32 [1]); only the first n_out rows are meaningful.  n_out (optional): an int32 [1] device tensor the
    live count is written into (the training loop hands in a row of its per-iteration log instead of copying into it afterwards).
    want_counts / front = (seg_order ctypes array or list, cap): the same launch also computes count_opaque's {#rays, #opaque rays} /
    writes the rank's front record (shard_front_pack) - a sixth return value, counts int32 [2] or record float32 [FRONT_HEADER + cap]."""
    require_device(rays, depths, keep, src_index)
    n_in = rays.shape[0]
    dev = rays.device
    n_seg = len(seg_start) - 1
    rays_out = torch.empty_like(rays)
    depths_out = torch.empty_like(depths)
    src_out = torch.empty_like(src_index) if src_index is not None else None
    # (both are written in full by the kernel: no fill launches - the front end of an iteration is a chain of small kernels)
    out_seg = torch.empty(n_seg + 1, device=dev, dtype=torch.int32)
    if n_out is None:
        n_out = torch.empty(1, device=dev, dtype=torch.int32)
    else:
        require_device(n_out)
        assert n_out.dtype == torch.int32 and n_out.numel() == 1 and n_out.is_contiguous()
    seg = seg_start if isinstance(seg_start, C.Array) else (C.c_int32 * (n_seg + 1))(*[int(v) for v in seg_start])
    if want_counts or front is not None:
        counts = rec = order = None
        cap = 0
        if front is not None:
            order, cap = front
            if not isinstance(order, C.Array):
                order = (C.c_int32 * n_seg)(*[int(v) for v in order])
            rec = torch.empty(hip.FRONT_HEADER + int(cap), device=dev, dtype=torch.float32)
        else:
            counts = torch.empty(2, device=dev, dtype=torch.int32)
        check(load().lnr_compact_rays_front(_ptr(rays), _ptr(depths), _ptr(keep), _ptr(src_index), n_in, seg, n_seg,
                                            _ptr(rays_out), _ptr(depths_out), _ptr(src_out), _ptr(out_seg), _ptr(n_out),
                                            _ptr(counts), order, int(cap), _ptr(rec), _stream()), "lnr_compact_rays_front")
        return rays_out, depths_out, src_out, out_seg, n_out, (rec if front is not None else counts)
    check(load().lnr_compact_rays(_ptr(rays), _ptr(depths), _ptr(keep), _ptr(src_index), n_in, seg, n_seg,
                                  _ptr(rays_out), _ptr(depths_out), _ptr(src_out), _ptr(out_seg), _ptr(n_out), _stream()),
          "lnr_compact_rays")
    return rays_out, depths_out, src_out, out_seg, n_out


def first_ray_key(rays, out_seg_start, seg_order):
    """int64 [1] on the device: (window order of this rank's first segment with a kept ray) << 32 | float bits of that ray's far,
    INT64_MAX without one (mapping/sharding.py: the far[0] of a sharded batch)."""
    require_device(rays, out_seg_start)
    n_seg = len(seg_order)
    key = torch.empty(1, device=rays.device, dtype=torch.int64)
    order = (C.c_int32 * n_seg)(*[int(v) for v in seg_order])
    check(load().lnr_first_ray_key(_ptr(rays), _ptr(out_seg_start), order, n_seg, _ptr(key), _stream()), "lnr_first_ray_key")
    return key


def shard_front_pack(rays, out_seg_start, seg_order, depths, n_rays_dev, cap, device=None):
    """A rank's "front record" for the sharded loop's one all-gather (include/loner_hip.h: lnr_shard_front_pack): float32
    [FRONT_HEADER + cap] = first-ray key | live-ray count | ground-truth depths.  rays None: a rank without keyframes."""
    dev = rays.device if rays is not None else torch.device(device)
    rec = torch.empty(hip.FRONT_HEADER + int(cap), device=dev, dtype=torch.float32)
    if rays is None:
        check(load().lnr_shard_front_pack(None, None, None, 0, None, 0, None, int(cap), _ptr(rec), _stream()), "lnr_shard_front_pack")
        return rec
    require_device(rays, out_seg_start, depths, n_rays_dev)
    n_seg = len(seg_order)
    order = seg_order if isinstance(seg_order, C.Array) else (C.c_int32 * n_seg)(*[int(v) for v in seg_order])
    check(load().lnr_shard_front_pack(_ptr(rays), _ptr(out_seg_start), order, n_seg, _ptr(_f32c(depths)), rays.shape[0], _ptr(n_rays_dev),
                                      int(cap), _ptr(rec), _stream()), "lnr_shard_front_pack")
    return rec


def shard_front_reduce(records, world, stride):
    """gathered front records [world * stride] -> (counts int32 [2] = {#rays, #opaque rays} of the WHOLE batch, far0 float [1])"""
    require_device(records)
    assert records.dtype == torch.float32 and records.numel() == world * stride
    counts = torch.empty(2, device=records.device, dtype=torch.int32)
    far0 = torch.empty(1, device=records.device, dtype=torch.float32)
    check(load().lnr_shard_front_reduce(_ptr(records), int(world), int(stride), _ptr(counts), _ptr(far0), _stream()), "lnr_shard_front_reduce")
    return counts, far0


def lidar_rays_backward(d_rays, rays, src_index, seg_start_dev, directions_list, transforms, scale):
    """-> d_transform [n_seg, 12]"""
    require_device(d_rays, rays, src_index, seg_start_dev, transforms)
    n_seg = len(directions_list)
    dev = rays.device
    out = torch.empty(n_seg, 12, device=dev, dtype=torch.float32) if rays.shape[0] > 0 else torch.zeros(n_seg, 12, device=dev)
    ptrs = (C.c_void_p * n_seg)(*[d.data_ptr() for d in directions_list])
    npts = (C.c_int64 * n_seg)(*[int(d.shape[1]) for d in directions_list])
    check(load().lnr_lidar_rays_backward(_ptr(_f32c(d_rays)), _ptr(rays), _ptr(src_index), _ptr(seg_start_dev), n_seg, ptrs,
                                         npts, _ptr(_f32c(transforms)), float(scale), _ptr(out), _stream()),
          "lnr_lidar_rays_backward")
    return out


# ---------------------------------------------------------------- samplers
def occ_interpolate(grid, pts):
    require_device(grid, pts)
    v = grid.shape[-1]
    g = _f32c(grid).reshape(v, v, v)
    p = _f32c(pts)
    out = torch.empty(p.shape[:-1], device=p.device, dtype=torch.float32)
    check(load().lnr_occ_interpolate(_ptr(g), v, _ptr(p), out.numel(), _ptr(out), _stream()), "lnr_occ_interpolate")
    return out


def sample_rays_occ(rays, grid, n_samples, perturb, u_jitter=None, u_pdf=None, seed=0, n_rays_dev=None, debug=False):
    require_device(rays, grid, u_jitter, u_pdf)
    rays = _f32c(rays)
    v = grid.shape[-1]
    g = _f32c(grid).reshape(v, v, v)
    n = rays.shape[0]
    h = n_samples // 2
    steps = linspace_table(h, rays.device)
    z = torch.empty(n, n_samples, device=rays.device, dtype=torch.float32)
    inds = probs = cdf = None
    if debug:
        inds = torch.zeros(n, h, device=rays.device, dtype=torch.int64)
        probs = torch.zeros(n, h, device=rays.device, dtype=torch.float32)
        cdf = torch.zeros(n, h - 1, device=rays.device, dtype=torch.float32)
    check(load().lnr_sample_rays_occ(_ptr(rays), n, _ptr(n_rays_dev), _ptr(g), v, n_samples, float(perturb), _ptr(steps),
                                     _ptr(_f32c(u_jitter)), _ptr(_f32c(u_pdf)), int(seed) & (2 ** 64 - 1), _ptr(z),
                                     _ptr(inds), _ptr(probs), _ptr(cdf), _stream()), "lnr_sample_rays_occ")
    if debug:
        return z, dict(inds=inds, probs=probs, cdf=cdf)
    return z


def sample_rays_uniform(rays, n_samples, perturb, u_jitter=None, seed=0, n_rays_dev=None):
    require_device(rays, u_jitter)
    rays = _f32c(rays)
    n = rays.shape[0]
    steps = linspace_table(n_samples, rays.device)
    z = torch.zeros(n, n_samples, device=rays.device, dtype=torch.float32)
    check(load().lnr_sample_rays_uniform(_ptr(rays), n, _ptr(n_rays_dev), n_samples, float(perturb), _ptr(steps),
                                         _ptr(_f32c(u_jitter)), int(seed) & (2 ** 64 - 1), _ptr(z), _stream()),
          "lnr_sample_rays_uniform")
    return z


# ---------------------------------------------------------------- rendering
def ftb_gather(rays, z, idx, n_alive, b0, block, rays_c, z_c, next_count):
    """front-to-back inference (lnr_render_ftb_gather): alive rays' records and block depths -> the compact arrays rays_c / z_c"""
    require_device(rays, z, idx, n_alive, rays_c, z_c, next_count)
    check(load().lnr_render_ftb_gather(_ptr(rays), _ptr(z), z.shape[1], _ptr(idx), _ptr(n_alive), rays.shape[0], int(b0), int(block),
                                       _ptr(rays_c), _ptr(z_c), _ptr(next_count), _stream()), "lnr_render_ftb_gather")


def ftb_composite(sigma_c, z, rays, idx, n_alive, b0, block, noise_std, seed, transmittance, depth_acc, opacity_acc, next_idx, next_count, last, noise=None):
    """front-to-back inference (lnr_render_ftb_composite): one block's contributions of the alive rays; survivors appended to next_idx"""
    require_device(sigma_c, z, rays, idx, n_alive, transmittance, depth_acc, opacity_acc, next_idx, next_count, noise)
    check(load().lnr_render_ftb_composite(_ptr(sigma_c), _ptr(z), _ptr(rays), z.shape[1], _ptr(idx), _ptr(n_alive), rays.shape[0], int(b0), int(block),
                                          _ptr(_f32c(noise) if noise is not None else None), float(noise_std), int(seed), _ptr(transmittance), _ptr(depth_acc), _ptr(opacity_acc), _ptr(next_idx),
                                          _ptr(next_count), 1 if last else 0, _stream()), "lnr_render_ftb_composite")


def render_forward(sigma, z, rays, noise=None, noise_std=0.0, seed=0, n_rays_dev=None, want_weights=True):
    """-> (depth, weights, opacity, variance); want_weights=False leaves the [n, S] weights unwritten (returns None for them)."""
    require_device(sigma, z, rays, noise)
    sigma, z, rays = _f32c(sigma), _f32c(z), _f32c(rays)
    n, s = z.shape
    dev = z.device
    depth = torch.zeros(n, device=dev); opacity = torch.zeros(n, device=dev); variance = torch.zeros(n, device=dev)
    weights = torch.zeros(n, s, device=dev) if want_weights else None
    check(load().lnr_render_forward(_ptr(sigma), _ptr(z), _ptr(rays), n, _ptr(n_rays_dev), s, _ptr(_f32c(noise)),
                                    float(noise_std), int(seed), _ptr(depth), _ptr(weights), _ptr(opacity), _ptr(variance),
                                    _stream()), "lnr_render_forward")
    return depth, weights, opacity, variance


def render_backward(sigma, z, rays, g_depth, g_weights, g_opacity, g_variance, noise=None, noise_std=0.0, seed=0,
                    n_rays_dev=None):
    require_device(sigma, z, rays)
    sigma, z, rays = _f32c(sigma), _f32c(z), _f32c(rays)
    n, s = z.shape
    d_sigma = torch.zeros(n, s, device=z.device)
    d_rays = torch.zeros(n, hip.RAY_STRIDE, device=z.device)
    check(load().lnr_render_backward(_ptr(sigma), _ptr(z), _ptr(rays), n, _ptr(n_rays_dev), s, _ptr(_f32c(noise)),
                                     float(noise_std), int(seed), _ptr(_f32c(g_depth)), _ptr(_f32c(g_weights)),
                                     _ptr(_f32c(g_opacity)), _ptr(_f32c(g_variance)), _ptr(d_sigma), _ptr(d_rays), _stream()),
          "lnr_render_backward")
    return d_sigma, d_rays


def points_grad_to_rays(d_pts, z, d_rays, n_rays_dev=None):
    require_device(d_pts, z, d_rays)
    n, s = z.shape
    check(load().lnr_points_grad_to_rays(_ptr(_f32c(d_pts)), _ptr(_f32c(z)), n, _ptr(n_rays_dev), s, _ptr(d_rays), _stream()),
          "lnr_points_grad_to_rays")
    return d_rays


# ---------------------------------------------------------------- loss pieces
def weights_gt(s, g, eps, normalise=True):
    require_device(s, g)
    s = _f32c(s)
    n, k = s.shape
    g = _f32c(g).reshape(-1)
    out = torch.empty_like(s)
    if isinstance(eps, torch.Tensor):
        require_device(eps)
        e = _f32c(eps).reshape(-1)
        check(load().lnr_weights_gt(_ptr(s), _ptr(g), _ptr(e), 0.0, int(normalise), n, k, _ptr(out), _stream()), "lnr_weights_gt")
    else:
        check(load().lnr_weights_gt(_ptr(s), _ptr(g), None, float(eps), int(normalise), n, k, _ptr(out), _stream()), "lnr_weights_gt")
    return out


def logits_grad(s, g, margin=2.0, l_free=0.25, l_occ=2.5):
    require_device(s, g)
    s = _f32c(s)
    n, k = s.shape
    out = torch.empty_like(s)
    check(load().lnr_logits_grad(_ptr(s), _ptr(_f32c(g).reshape(-1)), n, k, float(margin), float(l_free), float(l_occ),
                                 _ptr(out), _stream()), "lnr_logits_grad")
    return out


def count_opaque(rays, depth_gt, n_rays_dev=None, far0=None):
    """-> int32 [2] = {#rays, #opaque rays}.  far0 (device float [1], optional): the `far` every depth is compared with
    (the reference's far[0] quirk) when `rays` is only a shard of the batch."""
    require_device(rays, depth_gt, far0)
    counts = torch.empty(2, device=rays.device, dtype=torch.int32)       # overwritten, also for an empty batch
    check(load().lnr_count_opaque(_ptr(_f32c(rays)), _ptr(_f32c(depth_gt)), rays.shape[0], _ptr(n_rays_dev), _ptr(far0),
                                  _ptr(counts), _stream()), "lnr_count_opaque")
    return counts


def los_loss_fused(sigma, z, rays, depth_gt, scale, cfg: hip.LossConfig, counts, noise=None, noise_std=0.0, seed=0,
                   n_rays_dev=None, want_stats=False, want_weights=False, loss_out=None, far0=None, poison=None, poison_tag=0,
                   zero_dead_rows=True):
    """zero_dead_rows=False: d_rays rows at and beyond *n_rays_dev are left uninitialised (the kernel writes every column of every live
    row; the training loop never reads the others and saves a fill launch per iteration)."""
    require_device(sigma, z, rays, depth_gt, counts, noise, far0, poison)
    sigma, z, rays, depth_gt = _f32c(sigma), _f32c(z), _f32c(rays), _f32c(depth_gt)
    n, s = z.shape
    dev = z.device
    if loss_out is None:
        loss_out = torch.zeros(8, device=dev)
    d_sigma = torch.empty(n, s, device=dev)
    d_rays = (torch.zeros if (zero_dead_rows and n_rays_dev is not None) else torch.empty)(n, hip.RAY_STRIDE, device=dev)
    stats = torch.zeros(n, 8, device=dev) if want_stats else None
    w = torch.zeros(n, s, device=dev) if want_weights else None
    partials = torch.empty(((n + hip.LOSS_RAYS_PER_BLOCK - 1) // hip.LOSS_RAYS_PER_BLOCK) * 8, device=dev)
    check(load().lnr_los_loss_fused(_ptr(sigma), _ptr(z), _ptr(rays), _ptr(depth_gt), n, _ptr(n_rays_dev), s,
                                    _ptr(_f32c(noise)), float(noise_std), int(seed), float(scale), C.byref(cfg), _ptr(counts),
                                    _ptr(far0), _ptr(loss_out), _ptr(d_sigma), _ptr(d_rays), _ptr(stats), _ptr(w), _ptr(partials),
                                    _ptr(poison), int(poison_tag), _stream()),
          "lnr_los_loss_fused")
    return loss_out, d_sigma, d_rays, stats, w


# ---------------------------------------------------------------- optimisers
def adam_step(params, grads, exp_avg, exp_avg_sq, lr, step, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, zero_grad=True, poison=None):
    require_device(params, grads, exp_avg, exp_avg_sq, poison)
    check(load().lnr_adam_step(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), params.numel(), float(lr),
                               float(betas[0]), float(betas[1]), float(eps), int(step), float(grad_scale), int(zero_grad),
                               _ptr(poison), _stream()), "lnr_adam_step")


def occ_grid_step(grid, rays, z, depth_gt, scale, lr, margin=2.0, l_free=0.25, l_occ=2.5, grad_buf=None, n_rays_dev=None):
    """grad_buf: int64 [V^3] fixed-point accumulator (see the header) or None for the in-place float-atomic update."""
    require_device(grid, rays, z, depth_gt, grad_buf)
    assert grad_buf is None or (grad_buf.dtype == torch.int64 and grad_buf.is_contiguous())
    v = grid.shape[-1]
    n, s = z.shape
    check(load().lnr_occ_grid_step(_ptr(grid), v, _ptr(_f32c(rays)), _ptr(_f32c(z)), _ptr(_f32c(depth_gt)), n, _ptr(n_rays_dev),
                                   s, float(scale), float(lr), float(margin), float(l_free), float(l_occ), _ptr(grad_buf),
                                   _stream()), "lnr_occ_grid_step")


def occ_grid_apply(grid, grad_buf, lr, zero_grad=True, poison=None):
    require_device(grid, grad_buf, poison)
    assert grad_buf.dtype == torch.int64 and grad_buf.is_contiguous()
    check(load().lnr_occ_grid_apply(_ptr(grid), _ptr(grad_buf), grid.numel(), float(lr), int(zero_grad), _ptr(poison), _stream()),
          "lnr_occ_grid_apply")


def rng_draws(which, seed, n_rays, n_per_ray, device="cuda"):
    """The in-kernel generator's draws as a tensor [n_rays, n_per_ray] (hip.DRAW_JITTER / DRAW_PDF / DRAW_NOISE / DRAW_RAY_INDEX + segment)."""
    out = torch.empty(int(n_rays), int(n_per_ray), device=device, dtype=torch.float32)
    check(load().lnr_rng_draws(int(which), int(seed) & (2 ** 64 - 1), int(n_rays), int(n_per_ray), _ptr(out), _stream()), "lnr_rng_draws")
    return out


def selftest_mfma(device="cuda"):
    """max abs error of the three MFMA fragment-layout checks (fp32 16x16x4, fp16 16x16x32, the three-term bf16 split on 16x16x32);
    0.0 when all layouts hold (and the split product of the test operands is exact)."""
    out = torch.full((3,), -1.0, device=device)
    check(load().lnr_selftest_mfma(_ptr(out), _stream()), "lnr_selftest_mfma")
    return float(out.abs().max().item()) if bool((out >= 0).all()) else -1.0
