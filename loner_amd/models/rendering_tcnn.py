"""Volume rendering with the reference's entry points (src/models/rendering_tcnn.py), on HIP.

`render_rays` keeps the reference's signature and result dictionary (:192-266).  Forward and
backward run in libloner_hip.so: density network (lnr_density_forward/backward), compositing
(lnr_render_forward/backward) and the ray-record gradient (lnr_points_grad_to_rays); torch autograd
only connects the pieces (one autograd.Function), so a loss written in torch on top of the result
dictionary - e.g. the reference's own Optimizer.compute_loss - back-propagates to the density
parameters and to the ray records (and from there to the keyframe poses).
"""
import torch

from .. import ops


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, sigma_only=True, num_colors=3,
                softplus=False, far=None, ret_var=False, noise=None):
    """rendering_tcnn.py:71-147 for the lidar configuration (sigma_only, far given).  Forward only."""
    if not sigma_only or softplus or far is None:
        raise NotImplementedError("raw2outputs: only the lidar configuration (sigma_only=True, far given) is supported")
    n = z_vals.shape[0]
    rays = torch.zeros(n, 13, device=z_vals.device)
    rays[:, 3:6] = rays_d
    rays[:, 12] = far.reshape(-1)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (noise is None and raw_noise_std > 0) else 0
    depth, weights, opacity, variance = ops.render_forward(raw[..., 0], z_vals, rays, noise=noise,
                                                           noise_std=raw_noise_std, seed=seed)
    return torch.tensor([-1.]), depth, weights, opacity, (variance if ret_var else None)


class _RenderRays(torch.autograd.Function):
    """(rays, params) -> (depth, weights, opacity, variance); z, noise are constants."""

    @staticmethod
    def forward(ctx, rays, params, z, spec, noise, noise_std, seed):
        rays_c = rays.detach().float().contiguous()
        sigma = ops.density_forward(spec, params.detach(), rays=rays_c, z=z)
        depth, weights, opacity, variance = ops.render_forward(sigma, z, rays_c, noise=noise, noise_std=noise_std, seed=seed)
        ctx.save_for_backward(rays_c, params, z, sigma, noise if noise is not None else torch.empty(0))
        ctx.spec, ctx.noise_std, ctx.seed, ctx.has_noise = spec, noise_std, seed, noise is not None
        return depth, weights, opacity, variance

    @staticmethod
    def backward(ctx, g_depth, g_weights, g_opacity, g_variance):
        rays, params, z, sigma, noise = ctx.saved_tensors
        noise = noise if ctx.has_noise else None
        d_sigma, d_rays = ops.render_backward(sigma, z, rays, g_depth, g_weights, g_opacity, g_variance, noise=noise,
                                              noise_std=ctx.noise_std, seed=ctx.seed)
        want_rays = ctx.needs_input_grad[0]
        grad_params = torch.zeros_like(params)
        d_pts = ops.density_backward(ctx.spec, params.detach(), d_sigma, grad_params, rays=rays, z=z, want_d_pts=want_rays)
        if want_rays:
            ops.points_grad_to_rays(d_pts, z, d_rays)
        return (d_rays if want_rays else None), (grad_params if ctx.needs_input_grad[1] else None), None, None, None, None, None


def render_rays(rays, ray_sampler, nerf_model, ray_range, scale_factor, N_samples=64, retraw=False, perturb=0,
                white_bkgd=False, raw_noise_std=0., netchunk=32768, num_colors=3, sigma_only=False, DEBUG=False,
                detach_sigma=True, return_variance=False, noise=None):
    """rendering_tcnn.py:192-266.  Only sigma_only=True (LiDAR) is on the mapping path."""
    if not sigma_only:
        raise NotImplementedError("render_rays: colour rendering (camera=True) is not part of the LiDAR mapping path")
    z_vals = ray_sampler.get_samples(rays, N_samples, perturb)
    draws = getattr(ray_sampler, "_draws", None)
    if noise is None and draws is not None and raw_noise_std > 0:      # parity hook: the randn of rendering_tcnn.py:104
        noise = (draws.noise(z_vals.shape[0], N_samples) * raw_noise_std).to(z_vals.device)
    net = nerf_model._model_sigma
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (noise is None and raw_noise_std > 0) else 0
    depth, weights, opacity, variance = _RenderRays.apply(rays, net.params, z_vals, net.spec, noise, float(raw_noise_std), seed)
    result = {'rgb_fine': torch.tensor([-1.]), 'depth_fine': depth, 'weights_fine': weights, 'opacity_fine': opacity}
    if return_variance:
        result["variance"] = variance
    if retraw:
        result['samples_fine'] = z_vals
        result['points_fine'] = rays[:, None, 0:3] + rays[:, None, 3:6] * z_vals[:, :, None]
    return result


def inference(model, xyz_, dir_, sigma_only=False, netchunk=32768, detach_sigma=True, meshing=False):
    """rendering_tcnn.py:149-187: density at explicit points."""
    n_rays, n_samples = xyz_.shape[0:2]
    out = model(xyz_.reshape(-1, 3), None, sigma_only, detach_sigma)
    return out if meshing else out.view(n_rays, n_samples, -1)
