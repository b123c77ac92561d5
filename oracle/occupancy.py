"""Occupancy-grid lookup and update -- CPU oracle (test infrastructure).

Follows the reference:
  * OccupancyGridModel.interpolate  src/models/model_tcnn.py:122-131
    (torch grid_sample, 3-D 'bilinear', align_corners=False, zero padding)
  * get_logits_grad                 src/models/losses.py:54-62
  * Optimizer._step_occupancy_grid  src/mapping/optimizer.py:598-609
"""
import numpy as np
import torch


def trilinear_lookup_np(grid: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """grid: [V,V,V] indexed [z,y,x] float32; pts: [...,3] (x,y,z) in [-1,1].

    Reproduces grid_sample's arithmetic order so results are bit-identical to
    torch CPU: unnormalise ix=((x+1)*W-1)/2, corner weights as products of
    distances to the opposite corner, accumulate the 8 corners in the order
    (z0y0x0, z0y0x1, z0y1x0, z0y1x1, z1y0x0, ...), skipping out-of-range
    corners.
    """
    f32 = np.float32
    g = np.ascontiguousarray(grid, dtype=f32)
    D, H, W = g.shape
    p = np.asarray(pts, dtype=f32)
    shape = p.shape[:-1]
    p = p.reshape(-1, 3)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    ix = ((x + f32(1)) * f32(W) - f32(1)) / f32(2)
    iy = ((y + f32(1)) * f32(H) - f32(1)) / f32(2)
    iz = ((z + f32(1)) * f32(D) - f32(1)) / f32(2)
    x0 = np.floor(ix); y0 = np.floor(iy); z0 = np.floor(iz)
    x1 = x0 + f32(1); y1 = y0 + f32(1); z1 = z0 + f32(1)
    # weight of the *low* corner along an axis is distance to the high corner
    wx0 = x1 - ix; wx1 = ix - x0
    wy0 = y1 - iy; wy1 = iy - y0
    wz0 = z1 - iz; wz1 = iz - z0
    out = np.zeros(p.shape[0], f32)
    for (zc, wz) in ((z0, wz0), (z1, wz1)):
        for (yc, wy) in ((y0, wy0), (y1, wy1)):
            for (xc, wx) in ((x0, wx0), (x1, wx1)):
                # torch forms the corner weight as (wx*wy)*wz
                w = (wx * wy) * wz
                ok = (xc >= 0) & (xc < W) & (yc >= 0) & (yc < H) & (zc >= 0) & (zc < D)
                xi = np.clip(xc, 0, W - 1).astype(np.int64)
                yi = np.clip(yc, 0, H - 1).astype(np.int64)
                zi = np.clip(zc, 0, D - 1).astype(np.int64)
                v = g[zi, yi, xi]
                out = np.where(ok, out + v * w, out)
    return out.reshape(shape)


def trilinear_lookup(grid: torch.Tensor, pts: torch.Tensor) -> torch.Tensor:
    """Differentiable (w.r.t. grid) lookup; grid [1,1,V,V,V], pts [N,B,3]."""
    n, b, _ = pts.shape
    sampled = torch.nn.functional.grid_sample(
        grid, pts.reshape(1, 1, n, b, 3), mode="bilinear", align_corners=False)
    return sampled.reshape(n, b)


def logits_pseudo_grad(s_metres: torch.Tensor, g_metres: torch.Tensor,
                       margin: float = 2.0, free: float = 0.25, occ: float = 2.5) -> torch.Tensor:
    """losses.py:54-62.  +free in front of the surface (s < g-margin), -occ
    within `margin` of it, 0 behind; the step functions are 0 at 0."""
    x = s_metres - g_metres
    zero = torch.zeros((), dtype=x.dtype, device=x.device)
    step = lambda t: torch.heaviside(t, zero)
    return free * step(-x - margin) - occ * step(x + margin) * step(margin - x)


def grid_step(grid: torch.Tensor, points: torch.Tensor, s_metres: torch.Tensor,
              g_metres: torch.Tensor, lr: float) -> torch.Tensor:
    """One SGD step of the occupancy logits (optimizer.py:598-609).

    grid [1,1,V,V,V]; points [N,S,3]; s_metres [N,S]; g_metres [N,1].
    Returns the updated grid (new tensor)."""
    g = grid.detach().clone().requires_grad_(True)
    logits = trilinear_lookup(g, points.detach())
    logits.backward(gradient=logits_pseudo_grad(s_metres, g_metres))
    return (g.detach() - lr * g.grad).detach()
