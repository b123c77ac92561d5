"""Loss helpers with the reference's names and signatures (src/models/losses.py:29-62), on HIP."""
import torch

from .. import ops


def get_weights_gt(sampled_depth: torch.Tensor, gt_depth: torch.Tensor, eps, norm: bool = True) -> torch.Tensor:
    """Truncated-Gaussian target weights.  sampled_depth [N,S], gt_depth [N,1], eps float or [N,1]."""
    return ops.weights_gt(sampled_depth, gt_depth, eps, normalise=norm)


def get_logits_grad(z_vals: torch.Tensor, depth: torch.Tensor, eps=2, l_free=0.25, l_occ=2.5) -> torch.Tensor:
    """Occupancy pseudo-gradient per sample.  z_vals [N,S], depth [N,1] (both metres)."""
    return ops.logits_grad(z_vals, depth, margin=eps, l_free=l_free, l_occ=l_occ)
