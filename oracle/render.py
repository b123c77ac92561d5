"""Volume rendering of density samples -- CPU oracle (test infrastructure), torch.

Follows the reference's raw2outputs (src/models/rendering_tcnn.py:71-147) for
the lidar configuration it is called with (sigma_only=True, softplus=False,
far given, ret_var=True) and render_rays' sample placement (:233-241).
"""
import torch


def sample_points(rays: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """rays [N,13], z [N,S] -> xyz [N,S,3]  (rendering_tcnn.py:241)."""
    return rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]


def composite(sigma: torch.Tensor, z: torch.Tensor, dirs: torch.Tensor, far: torch.Tensor,
              noise=None):
    """sigma [N,S], z [N,S], dirs [N,3], far [N,1], noise [N,S] or None.

    Returns dict(depth [N], weights [N,S], opacity [N], variance [N]).
    The last interval is 1e10 long; intervals are scaled by |dir|;
    alpha = 1-exp(-delta*relu(sigma+noise)); transmittance uses 1-alpha+1e-10;
    the unexplained mass (1-sum w) is placed at `far`."""
    gaps = z[:, 1:] - z[:, :-1]
    gaps = torch.cat([gaps, 1e10 * torch.ones_like(gaps[:, :1])], dim=-1)
    gaps = gaps * torch.linalg.vector_norm(dirs[:, None, :], dim=-1)
    dens = sigma if noise is None else sigma + noise
    alpha = 1.0 - torch.exp(-gaps * torch.relu(dens))
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], dim=-1), dim=-1)[:, :-1]
    weights = alpha * trans
    opacity = weights.sum(-1)
    depth = (torch.cat([weights, 1.0 - weights.sum(dim=1, keepdim=True)], dim=1)
             * torch.cat([z, far], dim=-1)).sum(-1)
    variance = (weights * (depth.view(-1, 1) - z) ** 2).sum(dim=1)
    return dict(depth=depth, weights=weights, opacity=opacity, variance=variance)
