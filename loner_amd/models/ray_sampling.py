"""Ray samplers with the reference's interface (src/models/ray_sampling.py:18-92), on HIP.

get_samples(rays[N,13], N_samples, perturb) -> z_vals[N,N_samples] (detached, as in the reference,
whose sort runs under no_grad).  The two optional keyword arguments carry explicit random draws
(the reference calls torch.rand at :38/:72 and rendering_tcnn.py:48); when omitted the kernels use
their counter-based generator seeded from torch's CPU generator (one host draw, no device sync).
"""
import torch

from .. import ops


def _host_seed() -> int:
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class UniformRaySampler:
    def __init__(self):
        pass

    def get_samples(self, rays, N_samples, perturb, u_jitter=None, n_rays_dev=None):
        return ops.sample_rays_uniform(rays.detach(), N_samples, perturb, u_jitter=u_jitter,
                                       seed=_host_seed() if u_jitter is None else 0, n_rays_dev=n_rays_dev)


class OccGridRaySampler:
    def __init__(self):
        self._occ_gamma = None

    def update_occ_grid(self, occ_gamma):
        self._occ_gamma = occ_gamma

    def get_samples(self, rays, N_samples, perturb, u_jitter=None, u_pdf=None, n_rays_dev=None):
        if self._occ_gamma is None:
            raise RuntimeError("OccGridRaySampler: update_occ_grid() has not been called")
        need_seed = u_jitter is None or u_pdf is None
        return ops.sample_rays_occ(rays.detach(), self._occ_gamma, N_samples, perturb, u_jitter=u_jitter, u_pdf=u_pdf,
                                   seed=_host_seed() if need_seed else 0, n_rays_dev=n_rays_dev)
