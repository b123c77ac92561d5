"""Ray samplers with the reference's interface (src/models/ray_sampling.py:18-92), on HIP.

get_samples(rays[N,13], N_samples, perturb) -> z_vals[N,N_samples] (detached, as in the reference,
whose sort runs under no_grad).  The two optional keyword arguments carry explicit random draws
(the reference calls torch.rand at :38/:72 and rendering_tcnn.py:48); when omitted the kernels use
their counter-based generator seeded from torch's CPU generator (one host draw, no device sync).
For parity tests a `draws` object with the reference's draw order (jitter / pdf / noise) can be attached with
`set_draws`; `render_rays` then also takes the density noise from it.
"""
import torch

from .. import ops


def _host_seed() -> int:
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class _DrawsMixin:
    _draws = None

    def set_draws(self, draws):
        """draws: object with jitter(n, h) / pdf(n, h) / noise(n, s) returning CPU tensors, or None for the in-kernel generator"""
        self._draws = draws


class UniformRaySampler(_DrawsMixin):
    def __init__(self):
        pass

    def get_samples(self, rays, N_samples, perturb, u_jitter=None, n_rays_dev=None):
        if u_jitter is None and self._draws is not None and perturb > 0:
            u_jitter = self._draws.jitter(rays.shape[0], N_samples).to(rays.device)
        return ops.sample_rays_uniform(rays.detach(), N_samples, perturb, u_jitter=u_jitter,
                                       seed=_host_seed() if u_jitter is None else 0, n_rays_dev=n_rays_dev)


class OccGridRaySampler(_DrawsMixin):
    def __init__(self):
        self._occ_gamma = None

    def update_occ_grid(self, occ_gamma):
        self._occ_gamma = occ_gamma

    def get_samples(self, rays, N_samples, perturb, u_jitter=None, u_pdf=None, n_rays_dev=None):
        if self._occ_gamma is None:
            raise RuntimeError("OccGridRaySampler: update_occ_grid() has not been called")
        if self._draws is not None:
            if u_jitter is None and perturb > 0:
                u_jitter = self._draws.jitter(rays.shape[0], N_samples // 2).to(rays.device)
            if u_pdf is None:
                u_pdf = self._draws.pdf(rays.shape[0], N_samples // 2).to(rays.device)
        need_seed = u_jitter is None or u_pdf is None
        return ops.sample_rays_occ(rays.detach(), self._occ_gamma, N_samples, perturb, u_jitter=u_jitter, u_pdf=u_pdf,
                                   seed=_host_seed() if need_seed else 0, n_rays_dev=n_rays_dev)
