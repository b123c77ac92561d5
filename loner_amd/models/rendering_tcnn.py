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


class _Raw2Outputs(torch.autograd.Function):
    """(sigma [N,S], rays_d [N,3], far [N]) -> (depth, weights, opacity, variance): compositing by lnr_render_forward, its analytic
    backward by lnr_render_backward.  z_vals and the noise are constants (the reference's z_vals leave get_samples detached)."""

    @staticmethod
    def forward(ctx, sigma, rays_d, far, z, noise, noise_std, seed):
        n = z.shape[0]
        rays = torch.zeros(n, 13, device=z.device)
        rays[:, 3:6] = rays_d.detach()
        rays[:, 12] = far.detach().reshape(-1)
        sigma_c = sigma.detach().float().contiguous()
        depth, weights, opacity, variance = ops.render_forward(sigma_c, z, rays, noise=noise, noise_std=noise_std, seed=seed)
        ctx.save_for_backward(sigma_c, z, rays, noise if noise is not None else torch.empty(0))
        ctx.noise_std, ctx.seed, ctx.has_noise, ctx.far_shape = noise_std, seed, noise is not None, far.shape
        return depth, weights, opacity, variance

    @staticmethod
    def backward(ctx, g_depth, g_weights, g_opacity, g_variance):
        sigma, z, rays, noise = ctx.saved_tensors
        d_sigma, d_rays = ops.render_backward(sigma, z, rays, g_depth, g_weights, g_opacity, g_variance,
                                              noise=noise if ctx.has_noise else None, noise_std=ctx.noise_std, seed=ctx.seed)
        return (d_sigma if ctx.needs_input_grad[0] else None, d_rays[:, 3:6] if ctx.needs_input_grad[1] else None,
                d_rays[:, 12].reshape(ctx.far_shape) if ctx.needs_input_grad[2] else None, None, None, None, None)


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, sigma_only=True, num_colors=3,
                softplus=False, far=None, ret_var=False, noise=None):
    """rendering_tcnn.py:71-147 for the lidar configuration (sigma_only, far given).  Differentiable like the reference's (plain
    autograd there): gradients reach `raw` (the densities), `rays_d` (through |d| in the interval lengths, :100) and `far` (through
    the depth's background term, :126-130).  `z_vals` are constants here - in the reference's pipeline they leave get_samples
    detached (ray_sampling.py:75-90) - so a z_vals that requires a gradient is refused rather than silently ignored."""
    if not sigma_only or softplus or far is None:
        raise NotImplementedError("raw2outputs: only the lidar configuration (sigma_only=True, far given) is supported")
    if torch.is_grad_enabled() and z_vals.requires_grad:
        raise NotImplementedError("raw2outputs: a gradient with respect to z_vals is not implemented (the reference's samplers return "
                                  "detached depths); pass z_vals.detach()")
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (noise is None and raw_noise_std > 0) else 0
    depth, weights, opacity, variance = _Raw2Outputs.apply(raw[..., 0], rays_d, far, z_vals.detach().float().contiguous(),
                                                           noise, float(raw_noise_std), seed)
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
