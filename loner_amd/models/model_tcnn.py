"""Model / OccupancyGridModel with the reference's interface (src/models/model_tcnn.py), on HIP."""
from collections import defaultdict

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

    def forward(self, rays, ray_sampler, scale_factor, testing=False, camera=True, detach_sigma=True, return_variance=False):
        """Batched rendering in chunks of cfg.render.chunk rays (model_tcnn.py:70-105)."""
        if testing:
            n_samples, perturb = self.cfg.render.N_samples_test, 0.
        else:
            n_samples, perturb = self.cfg.render.N_samples_train, self.cfg.render.perturb
        results = defaultdict(list)
        for i in range(0, rays.shape[0], self.cfg.render.chunk):
            chunk = render_rays(rays[i:i + self.cfg.render.chunk, :], ray_sampler, self.nerf_model, self.cfg.ray_range,
                                scale_factor, N_samples=n_samples, retraw=self.cfg.render.retraw, perturb=perturb,
                                white_bkgd=self.cfg.render.white_bkgd, raw_noise_std=self.cfg.render.raw_noise_std,
                                netchunk=self.cfg.render.netchunk, num_colors=self.cfg.num_colors, sigma_only=(not camera),
                                detach_sigma=detach_sigma, return_variance=return_variance)
            for k, v in chunk.items():
                results[k] += [v]
        for k, v in results.items():
            results[k] = torch.cat(v, 0)
        return results


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
