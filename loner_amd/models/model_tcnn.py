"""Model / OccupancyGridModel with the reference's interface (src/models/model_tcnn.py), on HIP."""
import torch
import torch.nn as nn

from .. import ops
from .nerf_tcnn import DecoupledNeRF
from .rendering_tcnn import inference, render_rays


class Model(nn.Module):
    """Holds all trainable variables (model_tcnn.py:24-105)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        if cfg.model_type == 'nerf_decoupled':
            self.nerf_model = DecoupledNeRF(cfg.nerf_config, cfg.num_colors)
        else:
            raise NotImplementedError()

    def get_rgb_parameters(self, ignore_requires_grad=False):
        all_params = list(self.nerf_model._model_intensity.parameters()) + list(self.nerf_model._pos_encoding.parameters()) + \
            ([] if self.nerf_model._dir_encoding is None else list(self.nerf_model._dir_encoding.parameters()))
        return all_params if ignore_requires_grad else [p for p in all_params if p.requires_grad]

    def get_rgb_mlp_parameters(self):
        return list(self.nerf_model._model_intensity.parameters())

    def get_rgb_feature_parameters(self):
        params = list(self.nerf_model._pos_encoding.parameters()) + \
            ([] if self.nerf_model._dir_encoding is None else list(self.nerf_model._dir_encoding.parameters()))
        return [p for p in params if p.requires_grad]

    def get_sigma_parameters(self, ignore_requires_grad=False):
        all_params = list(self.nerf_model._model_sigma.parameters())
        return all_params if ignore_requires_grad else [p for p in all_params if p.requires_grad]

    def freeze_sigma_head(self, should_freeze=True):
        for p in self.get_sigma_parameters(True):
            p.requires_grad = not should_freeze

    def freeze_rgb_head(self, should_freeze=True):
        for p in self.get_rgb_parameters(True):
            p.requires_grad = not should_freeze

    def inference_points(self, xyz_, dir_, sigma_only):
        return inference(self.nerf_model, xyz_, dir_, netchunk=0, sigma_only=sigma_only, meshing=True)

    # rays per launch of the no-grad route: up to 2^23 samples per launch (4096 rays at 2048 samples: 1 GB of feature planes)
    _POINTS_PER_LAUNCH = 1 << 23

    def _sample_counts(self, testing):
        r = self.cfg.render
        return (r.N_samples_test, 0.) if testing else (r.N_samples_train, r.perturb)

    def _render_no_grad(self, rays, ray_sampler, n_samples, perturb, want_weights):
        """Rendering without autograd (Model.forward under no_grad / for inputs that need no gradient, render_depth): the whole
        batch goes through sampler -> density forward -> compositing in launches of _POINTS_PER_LAUNCH samples, whatever
        cfg.render.chunk says (the reference's chunk loop, model_tcnn.py:81-101, bounds ITS memory; results do not depend on
        it).  Nothing is kept for a backward pass: forward-only workspace, no [N,S] weights unless asked for.
        Parity hook: a `draws` object on the sampler is consumed chunk by chunk in the reference's order (per chunk: sampler
        draws, then the density noise), so recorded reference draws replay bit for bit."""
        rays = rays.detach().float().contiguous()
        net = self.nerf_model._model_sigma
        noise_std = float(self.cfg.render.raw_noise_std)
        n = rays.shape[0]
        draws = getattr(ray_sampler, "_draws", None)
        occ = hasattr(ray_sampler, "update_occ_grid")
        pre = None
        if draws is not None:
            u1, u2, nz = [], [], []
            for lo in range(0, n, self.cfg.render.chunk):
                m = min(self.cfg.render.chunk, n - lo)
                if perturb > 0:
                    u1.append(draws.jitter(m, n_samples // 2 if occ else n_samples))
                if occ:
                    u2.append(draws.pdf(m, n_samples // 2))
                if noise_std > 0:
                    nz.append(draws.noise(m, n_samples) * noise_std)
            cat = lambda xs: torch.cat(xs).to(rays.device) if xs else None
            pre = (cat(u1), cat(u2), cat(nz))
        step = max(64, self._POINTS_PER_LAUNCH // int(n_samples))
        out = {"depth": [], "opacity": [], "variance": [], "weights": [], "z": []}
        # Several launches and the in-kernel generator: the sampler of launch i + 1 runs on a second stream beside the density forward of
        # launch i (whose waves spend most of their cycles waiting for table lines: the sampler's sort fills them) - 2.3 of a scan's 31.7 ms
        # hidden.  Same kernels, same arguments, and the host generator is asked for its seeds in the same order as before
        # (sampler i, noise i, sampler i + 1, ...), so the result does not change.
        ahead = pre is None and rays.is_cuda and n > step
        if ahead:
            main = torch.cuda.current_stream(rays.device)
            # one side stream per device (a stream belongs to the device it was created on).  The sampler reads the occupancy grid the
            # ray_sampler holds: nothing updates it while a render is in flight - the optimiser's occupancy step and this route are
            # both issued by the one mapper thread, and the side stream has joined the main stream again when this call returns.
            streams = self.__dict__.setdefault("_sampler_streams", {})
            side = streams.get(rays.device)
            if side is None:
                side = streams[rays.device] = torch.cuda.Stream(rays.device)
            side.wait_stream(main)                   # the rays (and the occupancy grid) come from the main stream

            def sample_ahead(lo):
                r_ = rays[lo:lo + step]
                with torch.cuda.stream(side):
                    z_ = ray_sampler.get_samples(r_, n_samples, perturb)
                    ev_ = torch.cuda.Event()
                    ev_.record(side)
                return r_, z_, ev_
            coming = sample_ahead(0)
        for lo in range(0, n, step):
            r = rays[lo:lo + step]
            kw = {}
            noise = None
            if pre is not None:
                if pre[0] is not None:
                    kw["u_jitter"] = pre[0][lo:lo + step]
                if pre[1] is not None:
                    kw["u_pdf"] = pre[1][lo:lo + step]
                noise = pre[2][lo:lo + step] if pre[2] is not None else None
            if ahead:
                r, z, ev = coming
                seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if noise_std > 0 else 0
                coming = sample_ahead(lo + step) if lo + step < n else None
                main.wait_event(ev)
                z.record_stream(main)
            else:
                z = ray_sampler.get_samples(r, n_samples, perturb, **kw)
                seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (noise is None and noise_std > 0) else 0
            sigma = ops.density_forward(net.spec, net.params.detach(), rays=r, z=z, forward_only=True)
            depth, weights, opacity, variance = ops.render_forward(sigma, z, r, noise=noise, noise_std=noise_std, seed=seed,
                                                                   want_weights=want_weights)
            for k, v in (("depth", depth), ("opacity", opacity), ("variance", variance), ("weights", weights), ("z", z)):
                if v is not None and (k != "z" or self.cfg.render.retraw):
                    out[k].append(v)
        self.nerf_model.warn_if_clipped(rays.device)
        return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in out.items() if v}

    @torch.no_grad()
    def render_depth(self, rays, ray_sampler, scale_factor=None, testing=True, front_to_back=None):
        """Rendered depth per ray and nothing else (what compute_l1_depth, renderer_lidar and the meshing consumers read from the
        result dictionary: analysis/compute_l1_depth.py:56-58): the lean form of forward(testing=True, camera=False).
        front_to_back (None: cfg.render.front_to_back, default False): the opt-in route that stops evaluating a ray once its
        transmittance is below 2^-24 (_render_depth_front_to_back) - the same depth to ~1e-7 relative, about half the network
        evaluations on a trained map."""
        n_samples, perturb = self._sample_counts(testing)
        if rays.shape[0] == 0:
            return torch.empty(0, device=rays.device, dtype=torch.float32)
        if front_to_back is None:
            front_to_back = bool(self.cfg.render.get("front_to_back", False)) if hasattr(self.cfg.render, "get") else False
        if front_to_back and n_samples % self._FTB_BLOCK == 0 and n_samples > self._FTB_BLOCK:
            return self._render_depth_front_to_back(rays, ray_sampler, n_samples, perturb)
        return self._render_no_grad(rays, ray_sampler, n_samples, perturb, want_weights=False)["depth"]

    _FTB_BLOCK = 256          # samples per block along the ray (lnr_render_ftb_composite: one wave per ray, four samples per lane)
    _FTB_RAYS = 16384         # rays per chunk: a block of a chunk is one density-forward launch of at most 2^22 samples

    def _render_depth_front_to_back(self, rays, ray_sampler, n_samples, perturb):
        """The reference evaluates the density network at all N_samples_test depths of every ray (model_tcnn.py:73-105); a front-to-back
        composite stops contributing once the transmittance is gone.  Per chunk of rays: the sampler draws all depths (indices and depths
        are the default route's), then block after block of 256 samples along the rays the network runs on the rays still alive
        (include/loner_hip.h: lnr_render_ftb_*; the live count stays on the device - no host round trip, every block is launched).
        The sampler of chunk i + 1 runs on a second stream beside the blocks of chunk i, as in _render_no_grad."""
        rays = rays.detach().float().contiguous()
        net = self.nerf_model._model_sigma
        params = net.params.detach()
        noise_std = float(self.cfg.render.raw_noise_std)
        n, dev, B = rays.shape[0], rays.device, self._FTB_BLOCK
        step = self._FTB_RAYS
        # parity hook, as in _render_no_grad: a `draws` object on the sampler is consumed in the reference's order (per cfg.render.chunk of
        # rays: sampler draws, then the density noise) - both routes then see the same random numbers, whatever their launch sizes
        draws = getattr(ray_sampler, "_draws", None)
        pre = None
        if draws is not None:
            occ = hasattr(ray_sampler, "update_occ_grid")
            u1, u2, nz = [], [], []
            for lo in range(0, n, self.cfg.render.chunk):
                m_ = min(self.cfg.render.chunk, n - lo)
                if perturb > 0:
                    u1.append(draws.jitter(m_, n_samples // 2 if occ else n_samples))
                if occ:
                    u2.append(draws.pdf(m_, n_samples // 2))
                if noise_std > 0:
                    nz.append(draws.noise(m_, n_samples) * noise_std)
            cat = lambda xs: torch.cat(xs).to(dev) if xs else None
            pre = (cat(u1), cat(u2), cat(nz))
        main = torch.cuda.current_stream(dev)
        streams = self.__dict__.setdefault("_sampler_streams", {})
        side = streams.get(dev)
        if side is None:
            side = streams[dev] = torch.cuda.Stream(dev)
        side.wait_stream(main)

        def sample_ahead(lo):
            r_ = rays[lo:lo + step]
            kw = {}
            if pre is not None:
                if pre[0] is not None:
                    kw["u_jitter"] = pre[0][lo:lo + step]
                if pre[1] is not None:
                    kw["u_pdf"] = pre[1][lo:lo + step]
            with torch.cuda.stream(side):
                z_ = ray_sampler.get_samples(r_, n_samples, perturb, **kw)
                ev_ = torch.cuda.Event()
                ev_.record(side)
            return r_, z_, ev_
        coming = sample_ahead(0)
        out = []
        bufs = self.__dict__.setdefault("_ftb_buffers", {})
        for lo in range(0, n, step):
            r, z, ev = coming
            noise = pre[2][lo:lo + step].contiguous() if (pre is not None and pre[2] is not None) else None
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (noise_std > 0 and noise is None) else 0
            coming = sample_ahead(lo + step) if lo + step < n else None
            main.wait_event(ev)
            z.record_stream(main)
            m = r.shape[0]
            key = (dev, m)
            if key not in bufs:
                bufs.clear()                                  # (one chunk size at a time: a scan is full chunks and one ragged tail)
                bufs[key] = dict(rays_c=torch.empty(m, 13, device=dev), z_c=torch.empty(m, B, device=dev),
                                 idx=[torch.empty(m, device=dev, dtype=torch.int32) for _ in range(2)],
                                 cnt=[torch.empty(1, device=dev, dtype=torch.int32) for _ in range(2)])
            b = bufs[key]
            T = torch.ones(m, device=dev)
            dacc, oacc = torch.zeros(m, device=dev), torch.zeros(m, device=dev)
            b["idx"][0].copy_(torch.arange(m, device=dev, dtype=torch.int32))
            b["cnt"][0].fill_(m)
            cur = 0
            n_blocks = n_samples // B
            for k in range(n_blocks):
                idx, cnt, nidx, ncnt = b["idx"][cur], b["cnt"][cur], b["idx"][1 - cur], b["cnt"][1 - cur]
                ops.ftb_gather(r, z, idx, cnt, k * B, B, b["rays_c"], b["z_c"], ncnt)
                sigma_c = ops.density_forward(net.spec, params, rays=b["rays_c"], z=b["z_c"], n_rays_dev=cnt, forward_only=True)
                ops.ftb_composite(sigma_c, z, r, idx, cnt, k * B, B, noise_std, seed, T, dacc, oacc, nidx, ncnt, last=k == n_blocks - 1, noise=noise)
                cur = 1 - cur
            out.append(dacc + (1.0 - oacc) * r[:, 12])
        self.nerf_model.warn_if_clipped(dev)
        return out[0] if len(out) == 1 else torch.cat(out, 0)

    def forward(self, rays, ray_sampler, scale_factor, testing=False, camera=True, detach_sigma=True, return_variance=False):
        """Batched rendering with the reference's signature and result dictionary (model_tcnn.py:70-105).  When a gradient can
        flow (grad mode on and the rays or the density parameters require one) each cfg.render.chunk of rays goes through the
        differentiable render_rays; otherwise the batch takes the forward-only route in a few large launches."""
        if camera:
            raise NotImplementedError("Model.forward: colour rendering (camera=True) is not part of the LiDAR mapping path")
        if rays.shape[0] == 0:
            return {}                 # the reference's chunk loop does not run and its result dictionary stays empty (model_tcnn.py:81-105)
        n_samples, perturb = self._sample_counts(testing)
        sig = self.nerf_model._model_sigma.params
        if not (torch.is_grad_enabled() and (rays.requires_grad or sig.requires_grad)):
            r = self._render_no_grad(rays, ray_sampler, n_samples, perturb, want_weights=True)
            results = {'rgb_fine': torch.tensor([-1.]).repeat(-(-rays.shape[0] // self.cfg.render.chunk)), 'depth_fine': r["depth"],
                       'weights_fine': r["weights"], 'opacity_fine': r["opacity"]}
            if return_variance:
                results["variance"] = r["variance"]
            if self.cfg.render.retraw:
                results['samples_fine'] = r["z"]
                results['points_fine'] = rays[:, None, 0:3] + rays[:, None, 3:6] * r["z"][:, :, None]
            return results
        parts = [render_rays(rays[i:i + self.cfg.render.chunk, :], ray_sampler, self.nerf_model, self.cfg.ray_range,
                             scale_factor, N_samples=n_samples, retraw=self.cfg.render.retraw, perturb=perturb,
                             white_bkgd=self.cfg.render.white_bkgd, raw_noise_std=self.cfg.render.raw_noise_std,
                             netchunk=self.cfg.render.netchunk, num_colors=self.cfg.num_colors, sigma_only=True,
                             detach_sigma=detach_sigma, return_variance=return_variance)
                 for i in range(0, rays.shape[0], self.cfg.render.chunk)]
        return {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}


class OccupancyGridModel(nn.Module):
    """V^3 grid of occupancy log-odds (model_tcnn.py:108-131)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        v = cfg.voxel_size
        self.occupancy_grid = nn.Parameter(torch.zeros(1, 1, v, v, v))

    def forward(self):
        return self.occupancy_grid

    @staticmethod
    def interpolate(occupancy_grid, ray_bin_centers, mode='bilinear'):
        """Trilinear lookup of the grid at points [n_rays, n_bins, 3] -> [n_rays, n_bins]."""
        if mode != 'bilinear':
            raise NotImplementedError("only trilinear ('bilinear') interpolation is supported")
        return ops.occ_interpolate(occupancy_grid.detach(), ray_bin_centers.detach())
