"""Line-of-sight depth loss -- CPU oracle (test infrastructure), torch.

Follows the reference:
  * get_weights_gt                       src/models/losses.py:29-51
  * calculate_KL/JS_divergence           src/mapping/optimizer.py:614-626
  * Optimizer.compute_loss (lidar part)  src/mapping/optimizer.py:437-595
"""
import math
from dataclasses import dataclass

import torch


def target_weights(s: torch.Tensor, g: torch.Tensor, eps, normalise: bool = True) -> torch.Tensor:
    """Truncated-Gaussian target over sample depths.  s [N,S] metres, g [N,1],
    eps float or [N,1].  Operation order kept as in losses.py:38-50 (the clip
    points are formed as (g-eps-g)/sigma, not as -3)."""
    sd = eps / 3
    lo = (g - eps - g) / sd
    hi = (g + eps - g) / sd
    pdf = lambda t: 1.0 / math.sqrt(2 * math.pi) * torch.exp(-0.5 * (t ** 2))
    cdf = lambda t: 0.5 * (1 + torch.erf(t / math.sqrt(2)))
    dens = pdf((s - g) / sd) / sd / (cdf(hi) - cdf(lo))
    zero = torch.zeros_like(s)
    inside = torch.heaviside(s - (g - eps), zero) * torch.heaviside((g + eps) - s, zero)
    out = inside * dens
    if normalise:
        out = out / (out.sum(dim=1, keepdim=True) + 1e-6)
    return out


def gaussian_kl(m1, s1, m2, s2):
    return torch.log(s2 / s1) + (s1 * s1 + (m1 - m2) ** 2) / (2 * (s2 * s2)) - 0.5


def gaussian_js(m1, s1, m2, s2):
    mm = 0.5 * (m1 + m2)
    sm = 0.5 * torch.sqrt(s1 ** 2 + s2 ** 2)
    return 0.5 * gaussian_kl(m1, s1, mm, sm) + 0.5 * gaussian_kl(m2, s2, mm, sm)


@dataclass
class LossConfig:
    """Keys of model_config.loss (cfg/model_config/default_model_config.yaml:42-63)."""
    selection: str = "L1_JS"          # L1_JS | L2_JS | L1_LOS | L2_LOS
    min_js: float = 1.0
    max_js: float = 10.0
    js_alpha: float = 1.0
    los_lambda: float = 1000.0
    depth_lambda: float = 0.005
    min_eps: float = 0.5
    eps0: float = 3.0                 # LOS variants: starting tolerance
    eps_decay_rate: float = 0.95
    eps_decay_steps: float = 1.0
    decay_eps: bool = True


def ray_masks(rays: torch.Tensor, depth_gt: torch.Tensor, far0=None):
    """optimizer.py:460-463 including the broadcast quirk: an [N,1] > [N]
    comparison whose column 0 is kept, i.e. every depth is compared with the
    FIRST ray's far value.  far0: that value when `rays` is only a shard of
    the batch the reference would have evaluated at once (default: this
    batch's own first ray)."""
    far0 = rays[0, -1] if far0 is None else far0
    transparent = depth_gt.reshape(-1) > far0
    opaque = (depth_gt.reshape(-1) > 0) & ~transparent
    return opaque


def lidar_loss(rendered: dict, z: torch.Tensor, rays: torch.Tensor, depth_gt: torch.Tensor,
               scale, cfg: LossConfig, iteration: int = 0, far0=None):
    """rendered: output of render.composite (grad-carrying); z [N,S] (detached
    sample depths); depth_gt [N].  Returns (loss, aux) where aux carries the
    per-ray intermediates the tests compare."""
    opaque = ray_masks(rays, depth_gt, far0)
    s = z * scale
    g = depth_gt.reshape(-1, 1) * scale
    w = rendered["weights"]
    wsum = torch.sum(w, dim=1)
    mean = torch.sum(s * w, dim=1) / (wsum + 1e-10)
    var = torch.sum((s - mean[:, None]) ** 2 * w, dim=1) / (wsum + 1e-10) + 1e-10
    std = torch.sqrt(var)
    js = gaussian_js(g, cfg.min_eps / 3.0, mean[:, None], std[:, None]).reshape(-1)

    depth_m = rendered["depth"][:, None] * scale
    term_depth = cfg.depth_lambda * torch.nn.functional.mse_loss(depth_m[opaque, 0], g[opaque, 0])
    loss = term_depth

    if cfg.selection in ("L1_JS", "L2_JS"):
        score = js.detach().clone()
        score[score < cfg.min_js] = 0
        score[score > cfg.max_js] = cfg.max_js
        eps = (cfg.min_eps * (1 + cfg.js_alpha * score))[:, None]
        depth_eps = float(eps.mean())
    elif cfg.selection in ("L1_LOS", "L2_LOS"):
        if cfg.decay_eps:
            depth_eps = max(cfg.eps0 * cfg.eps_decay_rate ** (iteration / cfg.eps_decay_steps), cfg.min_eps)
        else:
            depth_eps = cfg.eps0
        eps = depth_eps
    else:
        raise ValueError(cfg.selection)
    target = target_weights(s, g, eps)
    target[~opaque, :] = 0
    if cfg.selection.startswith("L1"):
        los = torch.nn.functional.l1_loss(w, target)
    else:
        los = torch.nn.functional.mse_loss(w, target)
    term_los = cfg.los_lambda * los
    term_opacity = torch.abs(rendered["opacity"][opaque] - 1).mean()
    loss = loss + term_los
    loss = loss + term_opacity
    aux = dict(terms=(term_depth, term_los, term_opacity), opaque=opaque, mean=mean.detach(), std=std.detach(), js=js.detach(),
               eps=eps if isinstance(eps, float) else eps.detach(), target=target.detach(),
               depth_eps=depth_eps)
    return loss, aux
