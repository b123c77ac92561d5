"""One keyframe-window optimisation, end to end -- CPU oracle (test infrastructure).

Restates Optimizer._do_iterate_optimizer (src/mapping/optimizer.py:194-424)
for the lidar-only configuration the reference runs: per iteration, per
keyframe draw ray indices (:287-303), build rays (:305), render + loss
(:352 -> :437-595), backward (:366), Adam over [sigma params, poses]
(:257-269, :376-380), and every N_iters_acc-th global step the occupancy-grid
SGD step (:382-384 -> :598-609).  Adam is re-created for every call, exactly
as the reference does.

Random numbers come from a `draws` object so that a test can (a) use torch's
global generator in the reference's call order (A.9 of SURVEY.md) or (b) replay
recorded tensors into both this oracle and the HIP path.

Also used, timed, as the `cpu_baseline` ("port") leg of bench.py.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import loss as L
from . import network as NW
from . import occupancy as OC
from . import poses as P
from . import rays as R
from . import render as RD
from . import sampling as SP
from . import torch_sampling as TS


@dataclass(eq=False)
class OracleKeyframe:
    directions: torch.Tensor            # [3,n] sensor-frame unit vectors
    distances: torch.Tensor             # [n] metres
    pose6: torch.Tensor                 # [6] = [t, axis-angle]
    anchored: bool = False
    sky_directions: Optional[torch.Tensor] = None
    time: float = 0.0                   # start time of the scan (Frame.get_time): decides "latest keyframe"


class TorchDraws:
    """Draws in the reference's order from torch's global generator."""
    def ray_index(self, n_points, count):
        return torch.randint(n_points, (count,))

    def sky_index(self, n_sky, count):
        return torch.randint(0, n_sky, (count,))

    def jitter(self, n, h):
        return torch.rand(n, h)

    def pdf(self, n, h):
        return torch.rand(n, h)

    def noise(self, n, s):
        return torch.randn(n, s)


class AdamState:
    """torch.optim.Adam (betas 0.9/0.999, eps 1e-8, no weight decay), restated."""
    def __init__(self, tensors, lrs):
        self.tensors, self.lrs = tensors, lrs
        self.m = [torch.zeros_like(t) for t in tensors]
        self.v = [torch.zeros_like(t) for t in tensors]
        self.t = 0

    def step(self, grads, b1=0.9, b2=0.999, eps=1e-8):
        self.t += 1
        c1 = 1 - b1 ** self.t
        c2 = math.sqrt(1 - b2 ** self.t)
        for p, g, m, v, lr in zip(self.tensors, grads, self.m, self.v, self.lrs):
            if g is None:
                continue
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / c2).add_(eps)
            p.addcdiv_(m, denom, value=-(lr / c1))


@dataclass
class MapperConfig:
    n_rays: int = 512
    n_sky: int = 0
    n_samples: int = 512
    perturb: float = 1.0
    noise_std: float = 1.0
    ray_range: tuple = (1.0, 50.0)
    lr_sigma: float = 1e-2
    lr_pose: float = 1e-3
    occ_lr: float = 1e-4
    occ_every: int = 10
    use_occupancy: bool = True
    loss: L.LossConfig = field(default_factory=L.LossConfig)


class OracleMapper:
    def __init__(self, spec: NW.NetworkSpec, params: torch.Tensor, scale: float, shift,
                 cfg: MapperConfig, grid_size: int = 100, device="cpu", sampler="numpy"):
        """sampler "numpy": the rounding-exact emulation of torch's CPU kernels (sampling.py; parity tests, CPU only);
        "torch": the reference's own torch op sequence (torch_sampling.py; any device - the op-for-op baseline legs of bench.py).
        With device != "cpu" the keyframes' tensors must live on that device too."""
        self.spec, self.cfg = spec, cfg
        self.device, self.sampler = torch.device(device), sampler
        assert sampler == "torch" or self.device.type == "cpu"
        self.params = params.clone().float().to(self.device)
        self.scale = torch.tensor(float(scale), device=self.device)
        self.shift = torch.as_tensor(shift, dtype=torch.float32).to(self.device)
        self.grid = torch.zeros(1, 1, grid_size, grid_size, grid_size, device=self.device)
        self.global_step = 0
        self.draws = TorchDraws()
        self.trace = []

    # ---- one forward (+loss) over an already built ray batch -------------------------
    def forward_loss(self, rays, depths, iteration=0, draws=None, z_override=None):
        draws = draws or self.draws
        cfg = self.cfg
        n = rays.shape[0]
        half = cfg.n_samples // 2
        dev = self.device
        if self.sampler == "torch":
            if cfg.use_occupancy:
                u1 = draws.jitter(n, half).to(dev) if cfg.perturb > 0 else None
                z = TS.sample_occupancy(rays.detach(), self.grid, cfg.n_samples, cfg.perturb, u1, draws.pdf(n, half).to(dev))
            else:
                u1 = draws.jitter(n, cfg.n_samples).to(dev) if cfg.perturb > 0 else None
                z = TS.sample_uniform(rays.detach(), cfg.n_samples, cfg.perturb, u1)
        else:
            rays_np = rays.detach().numpy()
            if cfg.use_occupancy:
                u1 = draws.jitter(n, half) if cfg.perturb > 0 else None
                u2 = draws.pdf(n, half)
                z = SP.sample_occupancy(rays_np, self.grid[0, 0].numpy(), cfg.n_samples, cfg.perturb,
                                        None if u1 is None else u1.numpy(), u2.numpy())
            else:
                u1 = draws.jitter(n, cfg.n_samples) if cfg.perturb > 0 else None
                z = SP.sample_uniform(rays_np, cfg.n_samples, cfg.perturb, None if u1 is None else u1.numpy())
            z = torch.from_numpy(z)
        z = z if z_override is None else z_override
        xyz = RD.sample_points(rays, z)
        sigma = NW.density(self.spec, self.params, xyz.reshape(-1, 3)).reshape(n, cfg.n_samples)
        noise = (draws.noise(n, cfg.n_samples) * cfg.noise_std).to(dev) if cfg.noise_std > 0 else None
        out = RD.composite(sigma, z, rays[:, 3:6], rays[:, -1:], noise)
        loss, aux = L.lidar_loss(out, z, rays, depths, self.scale, cfg.loss, iteration)
        aux.update(z=z, xyz=xyz.detach(), out=out)
        return loss, aux

    # ---- the optimisation loop ---------------------------------------------------------
    def iterate(self, window: List[OracleKeyframe], n_iters: int, freeze_poses=False,
                freeze_sigma=False, draws=None, latest_kf_only=False, stop_after_s=None, min_iters=0, sync=None):
        """stop_after_s / min_iters / sync (bench.py's bounded baseline legs): leave the loop once that much wall time has passed
        (after at least min_iters iterations; sync() is called per iteration to finish the device's work first) - ONE optimisation
        phase with ONE Adam, cut short, instead of many one-iteration phases that would each start a fresh Adam.
        self.last_iterations says how many iterations ran."""
        import time as _time
        t_start = _time.time()
        self.last_iterations = 0
        draws = draws or self.draws
        cfg = self.cfg
        if len(window) == 1:
            window[0].anchored = True
        if latest_kf_only:              # optimizer.py:239-247: first keyframe with the strictly largest start time
            latest = window[0]
            for kf in window:
                if kf.time > latest.time:
                    latest = kf
            window = [latest]
        rr = torch.tensor(cfg.ray_range, device=self.device)
        self.params.requires_grad_(not freeze_sigma)
        free = [kf for kf in window if not kf.anchored and not freeze_poses]
        for kf in window:
            kf.pose6.requires_grad_(kf in free)
        tensors, lrs = [], []
        if not freeze_sigma:
            tensors.append(self.params); lrs.append(cfg.lr_sigma)
        for kf in free:
            tensors.append(kf.pose6); lrs.append(cfg.lr_pose)
        adam = AdamState(tensors, lrs)
        self.last_adam = adam            # (tests read the moments of the phase that just ran)
        n_valid = 0
        for it in range(n_iters):
            rays_all, depth_all = [], []
            for kf in window:
                idx = draws.ray_index(kf.distances.shape[0], cfg.n_rays).to(self.device)
                sky_idx = None
                if cfg.n_sky > 0 and kf.sky_directions is not None and kf.sky_directions.numel() > 0:
                    sky_idx = draws.sky_index(kf.sky_directions.shape[1], cfg.n_sky).to(self.device)
                T = P.transform_from_pose6(kf.pose6) if kf.pose6.requires_grad \
                    else P.transform_from_pose6(kf.pose6.detach())
                r, d = R.keyframe_ray_records(kf.directions, kf.distances, idx, T, rr, self.scale,
                                              self.shift, kf.sky_directions, sky_idx)
                rays_all.append(r); depth_all.append(d)
            rays = torch.cat(rays_all).float()
            depths = torch.cat(depth_all).float()
            n_valid += rays.shape[0]
            loss, aux = self.forward_loss(rays, depths, it, draws)
            assert not torch.isnan(loss), "NaN loss"
            grads = torch.autograd.grad(loss, tensors, allow_unused=True)
            with torch.no_grad():
                adam.step(grads)
            self.trace.append(float(loss.detach()))
            if cfg.use_occupancy and self.global_step % cfg.occ_every == 0:
                g = depths.reshape(-1, 1) * self.scale
                self.grid = OC.grid_step(self.grid, aux["xyz"], aux["z"] * self.scale, g, cfg.occ_lr)
            self.global_step += 1
            self.last_iterations = it + 1
            if stop_after_s is not None:
                if sync is not None:
                    sync()
                if it + 1 >= min_iters and _time.time() - t_start >= stop_after_s:
                    break
        self.params.requires_grad_(False)
        for kf in window:
            kf.pose6.requires_grad_(False)
        return n_valid
