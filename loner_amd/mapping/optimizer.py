"""Optimizer: the per-keyframe optimisation loop of the mapping thread, on the MI355X.

Same class surface as the reference's src/mapping/optimizer.py (constructor :74-76,
iterate_optimizer :144, _do_iterate_optimizer :194, compute_loss :437, should_enable_lidar :427,
_step_occupancy_grid :598, and the privates the Mapper reads: _keyframe_count, _global_step, _model,
_optimizer, _occupancy_grid_model, _occupancy_grid_optimizer - mapper.py:104-130,161-175), so
it drops into the reference's Mapper unchanged.  What differs is where the work happens:

  reference (per iteration)                      here
  --------------------------------------------   ------------------------------------------------
  CPU randint + CPU ray build + H2D copy          index draw, gather/rotate/clip and compaction on
  (optimizer.py:285-340)                          the GPU from HBM-resident keyframe buffers
  ~20 torch ops materialising [N,S] tensors       sampler / density fwd / fused render+loss+backward
  (ray_sampling.py, rendering_tcnn.py,            / density bwd kernels chained on one HIP stream
   optimizer.py:437-595) + autograd
  loss.item() and eps.cpu() syncs (:354,:503)     no host sync inside the loop; the NaN / finite
  per-iteration checks before step (:368-374,     checks are made on the device in every iteration (a
  :590)                                           "poison" word the step kernels obey) and raised
                                                  once per phase
  torch.optim.Adam over 7.4 M params              one fused Adam kernel (lnr_adam_step)

Random numbers: by default the kernels' counter-based generator is used (seeded per iteration from
torch's CPU generator).  For parity tests a `draws` object with the reference's draw order
(per-keyframe randint, then rand, rand, randn) can be injected via `set_draws`.
"""
import os
import time
from dataclasses import dataclass
from typing import List, Tuple

import torch

from .. import hip, ops
from ..common.pose_utils import WorldCube, tensor_to_transform
from ..common.ray_utils import device_scan
from ..models.model_tcnn import Model, OccupancyGridModel
from ..models.ray_sampling import OccGridRaySampler, UniformRaySampler


@dataclass
class OptimizationSettings:
    """Parameters of one optimisation phase (optimizer.py:41-60)."""
    num_iterations: int = 1
    freeze_poses: bool = False
    latest_kf_only: bool = False
    freeze_sigma_mlp: bool = False
    freeze_rgb_mlp: bool = False

    def from_dict(dict):
        get = lambda k, d: dict[k] if k in dict else d
        return OptimizationSettings(get("num_iterations", 1), get("freeze_poses", False), get("latest_kf_only", False),
                                    get("freeze_sigma_mlp", False), get("freeze_rgb_mlp", False))


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (betas 0.9/0.999, eps 1e-8, no weight decay) with the update done by
    lnr_adam_step.  State keys match torch's Adam so that state_dict() round-trips (mapper.py:161-175)."""

    def __init__(self, param_groups, poison=None):
        super().__init__(param_groups, dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8))
        self.poison = poison          # failure guard of the optimisation phase (int32[2] device word, include/loner_hip.h) or None

    @torch.no_grad()
    def step(self, zero_grad=True, groups=None, ranges=None):
        """groups: indices of the parameter groups to step (default: all).  ranges: [(lo, hi), ...] element ranges of the
        (flat) parameters to step instead of all of them - the sharded form in which a rank owns a slice of the hash tables
        (mapping/sharding.py); moments outside the ranges are left alone, the step count advances once."""
        return self.step_now(zero_grad, groups, ranges)

    @torch.no_grad()
    def step_now(self, zero_grad=True, groups=None, ranges=None):
        """step() without torch.optim.Optimizer's per-call wrapper (profiler record, pre / post hooks: ~15 us of host time per call, two
        calls per iteration - a tenth of a one-keyframe rank's host budget).  The training loop calls this; `step` is the public form."""
        if groups is not None and len(groups) == 0:
            return
        for gi, group in enumerate(self.param_groups):
            if groups is not None and gi not in groups:
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                flat = (p.data.view(-1), p.grad.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1))
                for lo, hi in (ranges if ranges is not None else [(0, flat[0].numel())]):
                    ops.adam_step(flat[0][lo:hi], flat[1][lo:hi], flat[2][lo:hi], flat[3][lo:hi],
                                  group["lr"], st["step"], betas=group["betas"], eps=group["eps"], zero_grad=zero_grad,
                                  poison=self.poison)


class _LidarLossFn(torch.autograd.Function):
    """loss = compute_loss(rays, params): the HIP path computes the loss AND its gradients in the forward
    call; backward hands them to autograd (scaled by the incoming gradient)."""

    @staticmethod
    def forward(ctx, rays, params, opt, depths, iteration_idx):
        out = opt._loss_and_grads(rays.detach().float().contiguous(), depths, params, iteration_idx,
                                  want_ray_grads=rays.requires_grad, want_param_grads=params.requires_grad)
        ctx.save_for_backward(out["d_rays"] if out["d_rays"] is not None else torch.empty(0),
                              out["grad_params"] if out["grad_params"] is not None else torch.empty(0))
        ctx.has = (out["d_rays"] is not None, out["grad_params"] is not None)
        ctx.rays_device = rays.device
        return out["loss"][0].clone()

    @staticmethod
    def backward(ctx, g):
        d_rays, grad_params = ctx.saved_tensors
        return ((g * d_rays).to(ctx.rays_device) if ctx.has[0] else None), (g * grad_params if ctx.has[1] else None), None, None, None


class Optimizer:
    def __init__(self, settings, calibration, world_cube: WorldCube, device, use_gt_poses: bool = False,
                 lidar_only: bool = True, enable_sky_segmentation: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError("loner_amd.mapping.Optimizer needs an MI355X/HIP device; there is no CPU path")
        hip.load()                                           # fail early and loudly if the library is missing
        self._settings = settings
        self._calibration = calibration
        self._device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self._use_gt_poses = use_gt_poses
        self._optimization_settings = OptimizationSettings()
        self._lidar_only = lidar_only
        self._model_config = settings.model_config
        self._scale_factor = world_cube.scale_factor
        self._data_prep_device = 'cpu' if settings.data_prep_on_cpu else self._device
        self._world_cube = world_cube.to(self._data_prep_device)
        self._ray_range = torch.Tensor(list(self._model_config.model.ray_range)).to(self._data_prep_device)
        self._ray_range_f = [float(v) for v in self._ray_range.reshape(-1).cpu()]       # (host floats: read in every iteration)
        self._scale_f = float(world_cube.scale_factor)
        self._shift_f = [float(v) for v in torch.as_tensor(world_cube.shift).reshape(-1).cpu()]

        self._model = Model(self._model_config.model).to(self._device)

        if self._settings.samples_selection.strategy == 'OGM':
            self._occupancy_grid_model = OccupancyGridModel(self._model_config.model.occ_model).to(self._device)
            self._occupancy_grid = self._occupancy_grid_model()
            occ_params = [p for p in self._occupancy_grid_model.parameters() if p.requires_grad]
            self._occupancy_grid_optimizer = torch.optim.SGD(occ_params, lr=self._model_config.model.occ_model.lr)
            self._ray_sampler = OccGridRaySampler()
            self._ray_sampler.update_occ_grid(self._occupancy_grid.detach())
        elif self._settings.samples_selection.strategy == 'UNIFORM':
            self._ray_sampler = UniformRaySampler()
        else:
            raise RuntimeError(f"Can't find samples_selection strategy: {self._settings.samples_selection.strategy}")
        if not self._lidar_only:
            raise NotImplementedError("camera rays are not enabled in the reference either (optimizer.py:433-434)")

        self._keyframe_count = 0
        self._global_step = 0
        self._keyframe_schedule = self._settings["keyframe_schedule"]
        self._optimizer = None
        self._num_lidar_samples = self._settings.num_samples.lidar
        self._enable_sky_segmentation = enable_sky_segmentation
        self._progress_bar = None

        self._draws = None           # None: in-kernel generator; else an object with the reference's draw order
        self._dist = None            # sharded-window context (mapping/sharding.py), None on one GPU
        self._results_lidar = None
        self._depth_eps = None
        self._grad_buf = None
        self._pipeline = True         # False: everything on one stream (same results; the tests compare the two)
        self._side_stream = None      # second stream + event of the pipelined loop, created on first use
        self._grad_event = None
        self._poison = None          # failure guard of the running phase (device int32[2]), None outside a phase
        self.last_failure = None
        self._pc = None               # per-phase cache of the configuration values read in every iteration (_consts)
        self._pending_density = None  # (all-reduce handle or None, group index, lr): density Adam step deferred by the training loop
        self._defer_density_step = True   # False: step right away (same arithmetic; the tests compare the two)
        self._overwrite_grads = True      # False: the density gradient is accumulated and zeroed by Adam (same results; the tests compare the two)
        self.last_stats = {}

    # -------------------------------------------------------------------------------------------
    def _consts(self):
        """Configuration values the loop reads in EVERY iteration, looked up once per optimisation phase (the settings object wraps
        every nested dictionary anew on each attribute access: ~35 lookups per iteration were 20 us of a one-keyframe rank's 350 us of
        host time, which is what bounds that rank once two collectives are enqueued - DESIGN.md section 5).  Dropped at the start of
        every phase and by anyone who edits the configuration between calls (self._pc = None)."""
        pc = self._pc
        if pc is None:
            render, lc = self._model_config.model.render, self._model_config.loss
            if lc.loss_selection not in hip.LOSS_SELECTIONS:
                raise ValueError(f"Can't use unknown Loss {lc.loss_selection}")
            ogm = self._settings.samples_selection.strategy == 'OGM'
            pc = self._pc = dict(
                S=render.N_samples_train, perturb=render.perturb, noise_std=float(render.raw_noise_std), ogm=ogm,
                occ_every=int(self._model_config.model.occ_model.N_iters_acc) if ogm else 0,
                strategy=self._settings.rays_selection.strategy,
                loss=dict(selection=hip.LOSS_SELECTIONS[lc.loss_selection], min_js=lc.JS_loss.min_js_score, max_js=lc.JS_loss.max_js_score,
                          js_alpha=lc.JS_loss.alpha, depth_lambda=lc.depthloss_lambda, min_eps=lc.min_depth_eps,
                          decay_los=bool(lc.decay_los_lambda), los_lambda=lc.los_lambda,
                          los_rate=lc.los_lambda_decay_rate if lc.decay_los_lambda else None,
                          los_steps=lc.los_lambda_decay_steps if lc.decay_los_lambda else None,
                          min_los=lc.min_los_lambda if lc.decay_los_lambda else None,
                          decay_eps=bool(lc.decay_depth_eps), depth_eps=lc.depth_eps,
                          eps_rate=lc.depth_eps_decay_rate if lc.decay_depth_eps else None,
                          eps_steps=lc.depth_eps_decay_steps if lc.decay_depth_eps else None))
        return pc

    def set_draws(self, draws):
        """Inject host random draws (objects with ray_index/sky_index/jitter/pdf/noise, see oracle.mapping_step)."""
        self._draws = draws

    def set_distributed(self, dist_ctx):
        self._dist = dist_ctx

    # -------------------------------------------------------------------------------------------
    def iterate_optimizer(self, keyframe_window: List, optimizer_settings: OptimizationSettings = None) -> float:
        """Run the iteration schedule selected by the keyframe count (optimizer.py:144-192)."""
        cumulative_kf_idx = 0
        for item in self._keyframe_schedule:
            kf_count = item["num_keyframes"]
            iteration_schedule = item["iteration_schedule"]
            cumulative_kf_idx += kf_count
            if cumulative_kf_idx >= self._keyframe_count + 1 or kf_count == -1:
                break
        num_its = sum(i["num_iterations"] for i in iteration_schedule)
        if self._settings.debug.profile_optimizer:
            # optimizer.py:158-176: a torch profiler around the phase (one step per iteration: wait 1, warm up 1, record the rest),
            # TensorBoard traces under <log_directory>/profile/tensorboard_optimizer/, no timing.csv row.  The profiler's CUDA activity
            # is the HIP activity on ROCm; the loop takes its single-stream form while a profiler is attached.
            from torch.profiler import ProfilerActivity, profile, schedule, tensorboard_trace_handler
            prof_dir = f"{self._settings.log_directory}/profile"
            os.makedirs(prof_dir, exist_ok=True)
            prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], profile_memory=True, record_shapes=True,
                           with_stack=True, with_modules=True, schedule=schedule(wait=1, warmup=1, active=max(num_its - 2, 1)),
                           on_trace_ready=tensorboard_trace_handler(f"{prof_dir}/tensorboard_optimizer/"))
            prof.start()
            start_time = time.time()
            result = self._do_iterate_optimizer(keyframe_window, iteration_schedule, prof, optimizer_settings=optimizer_settings)
            torch.cuda.synchronize(self._device)
            elapsed_time = time.time() - start_time
            prof.stop()
            print(f"Elapsed Time: {elapsed_time}. Per Iteration: {elapsed_time / num_its}, Its/Sec: {num_its / elapsed_time}")
        else:
            start_time = time.time()
            result = self._do_iterate_optimizer(keyframe_window, iteration_schedule, optimizer_settings=optimizer_settings)
            torch.cuda.synchronize(self._device)
            elapsed_time = time.time() - start_time
            os.makedirs(self._settings.log_directory, exist_ok=True)
            with open(f"{self._settings.log_directory}/timing.csv", 'a+') as f:
                f.write(f"{num_its},{elapsed_time}\n")
            if self._progress_bar is None:
                print(f"Elapsed Time: {elapsed_time}. Per Iteration: {elapsed_time / num_its}, Its/Sec: {num_its / elapsed_time}")
        self._keyframe_count += 1
        return result

    def should_enable_lidar(self) -> bool:
        return not self._optimization_settings.freeze_sigma_mlp or not self._optimization_settings.freeze_poses

    def should_enable_camera(self) -> bool:
        return False

    # -------------------------------------------------------------------------------------------
    def _do_iterate_optimizer(self, keyframe_window: List, iteration_schedule, profiler=None,
                              optimizer_settings: OptimizationSettings = None) -> float:
        if len(keyframe_window) == 1:           # (sharded: every rank is handed the whole window, so all agree)
            keyframe_window[0].is_anchored = True
        if len(iteration_schedule) > 1 and self._settings.skip_pose_refinement:
            iteration_schedule = iteration_schedule[1:]
        losses_log, depth_eps_log = [], []
        if optimizer_settings is not None:
            iteration_schedule = [None]

        for iteration_config in iteration_schedule:
            if optimizer_settings is None:
                os_ = self._optimization_settings
                os_.freeze_poses = iteration_config["freeze_poses"] or self._settings.freeze_poses or self._use_gt_poses
                os_.latest_kf_only = iteration_config["latest_kf_only"] if "latest_kf_only" in iteration_config else False
                os_.freeze_rgb_mlp = iteration_config["freeze_rgb_mlp"]
                os_.freeze_sigma_mlp = iteration_config["freeze_sigma_mlp"]
                os_.num_iterations = iteration_config["num_iterations"]
            else:
                self._optimization_settings = optimizer_settings
            os_ = self._optimization_settings
            os_.freeze_poses = os_.freeze_poses or self._settings.freeze_poses or self._use_gt_poses
            self._pc = None
            pc = self._consts()

            if self._settings.samples_selection.strategy == 'OGM':
                self._ray_sampler.update_occ_grid(self._occupancy_grid.detach())
            self._model.freeze_sigma_head(os_.freeze_sigma_mlp)
            self._model.freeze_rgb_head(True)
            optimize_poses = not os_.freeze_poses

            # The phase is decided on the WHOLE window (latest keyframe, anchoring); in the sharded mode every rank is handed the
            # same window and keeps the keyframes i mod G == rank of the active list (mapping/sharding.py).  A rank may end up
            # with none (window smaller than the world size: the first keyframes of every run): it still joins every collective.
            if os_.latest_kf_only:
                active_all = [max(keyframe_window, key=lambda kf: float(kf.get_time()))]
            else:
                active_all = list(keyframe_window)
            active = self._dist.owned(active_all) if self._dist is not None else active_all
            # positions of this rank's keyframes in the window (the sharded far[0] agreement needs them)
            self._active_order = self._dist.owned_indices(len(active_all)) if self._dist is not None else list(range(len(active_all)))
            # depth slots of the sharded loop's front record (mapping/sharding.py): the candidate rays of the fullest rank
            n_sky_cfg = self._settings.num_samples.sky if self._enable_sky_segmentation else 0
            self._front_cap = self._dist.front_capacity(len(active_all), self._num_lidar_samples + n_sky_cfg) if self._dist is not None else 0
            for kf in active:
                if not kf.is_anchored:
                    kf.get_lidar_pose().set_fixed(not optimize_poses)
            tracking = (not os_.freeze_poses) and os_.freeze_rgb_mlp and os_.freeze_sigma_mlp

            sigma_params = self._model.get_sigma_parameters()
            groups = []
            density_group = None
            if not tracking and sigma_params:
                density_group = len(groups)
                groups.append({'params': sigma_params, 'lr': self._model_config.train.lrate_sigma_mlp})
            # poses are optimised on the device as one [K,6] tensor (rows of fixed/anchored keyframes get zero gradient).
            # Which pose a keyframe's rays are built from is the reference's choice: the ground-truth pose when the optimiser was
            # constructed with use_gt_poses (keyframe.py:83-86 via optimizer.py:305), else the lidar pose.  A pose that is NOT optimised
            # in this phase enters as the 4x4 matrix its get_transformation_matrix() hands out - which is what the reference's ray
            # builder reads (pose.py:140-144: for a fixed pose that is the matrix cached at construction, whatever happened to the
            # 6-vector since) - not as a matrix re-derived from get_pose_tensor().
            pose_objs = [(kf._frame._gt_lidar_pose if self._use_gt_poses else kf.get_lidar_pose()) for kf in active]
            free_list = [bool(optimize_poses and not kf.is_anchored) for kf in active]
            pose_cpu = [p.get_pose_tensor() for p in pose_objs]
            if active and all(p.device.type == "cpu" for p in pose_cpu):
                pose_dev = torch.stack([p.detach().float() for p in pose_cpu]).to(self._device).contiguous()      # one upload
            elif active:
                pose_dev = torch.stack([p.detach().to(self._device, torch.float32) for p in pose_cpu]).contiguous()
            else:
                pose_dev = torch.zeros(0, 6, device=self._device)
            free_rows = torch.tensor(free_list, device=self._device).to(torch.uint8)
            tab = self._window_tables(active) if active else None
            if tab is not None and not all(free_list):
                with torch.no_grad():
                    fixed = [(torch.zeros(12) if fr else p.get_transformation_matrix().detach().float().cpu()[:3, :4].reshape(12))
                             for p, fr in zip(pose_objs, free_list)]
                tab.fixed_T12 = torch.stack(fixed).to(self._device).contiguous()
                tab.free_col = free_rows.bool()[:, None]
                tab.any_free = any(free_list)
            elif tab is not None:
                tab.fixed_T12 = None
            any_free = bool(optimize_poses and any(not kf.is_anchored for kf in active))
            pose_dev.requires_grad_(any_free)
            if any_free:
                groups.append({'params': [pose_dev], 'lr': self._model_config.train.lrate_pose})
            # always an Adam, like the reference (Mapper.build_ckpt reads its state_dict unconditionally, mapper.py:161-175)
            # failure guard: {code, iteration} written by the loss / pose-gradient kernels of the iteration that fails; every
            # step kernel after it does nothing (the reference raises before optimizer.step(), optimizer.py:368-376,590)
            poison = torch.zeros(2, device=self._device, dtype=torch.int32)
            self._poison = poison
            self._idle_grad_cleared = False              # (_join_without_rays: an idle rank's zero contribution, cleared once per phase)
            self._optimizer = HipAdam(groups if groups else [{'params': []}], poison=poison)
            gamma = float(self._model_config.train.lrate_gamma)
            base_lrs = [g['lr'] for g in groups]
            # the density step may be deferred (see below) when the density parameters are trained in this phase
            if os_.freeze_sigma_mlp or not self._defer_density_step:
                density_group = None
            n_it = os_.num_iterations
            loss_log = torch.zeros(max(n_it, 1), 8, device=self._device)
            valid_log = torch.zeros(max(n_it, 1), device=self._device, dtype=torch.int32)

            # Pipelined loop (one GPU, keyframes to work on): the input gradient of an iteration is complete when the encode backward
            # ends, while the table-gradient reduce, the weight-gradient fold and the (deferred) density Adam step still follow.
            # Everything that depends on the input gradient only - pose gradient, pose step, occupancy step - and the whole front
            # end of the NEXT iteration (pose -> [R|t], ray draw / build / compaction, loss normalisers, sampler: ~0.2 ms of small,
            # latency-bound kernels) runs on a second stream beside that tail; the streams meet again in front of the next density
            # forward.  Same kernels, same arguments, same order of the random draws: the results do not change.
            # (measured on the bench window: 2.35 -> 2.33 ms per iteration - the reduce occupies every wave slot, so the side stream
            # mostly fills its tail; with a single keyframe the two extra stream hand-overs cost more than they hide: 0.50 -> 0.51 ms)
            pipelined = len(active) >= 2 and self._dist is None and self._pipeline and profiler is None and self.should_enable_lidar()
            if pipelined:
                main = torch.cuda.current_stream(self._device)
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(self._device)     # (a high-priority stream measured the same: 2.32 ms)
                    self._grad_event = torch.cuda.Event()
                    self._grad_event.record(main)                   # (torch creates the handle at the first record)
                side, ev = self._side_stream, self._grad_event
                sp = sigma_params[0] if sigma_params else self._model.nerf_model._model_sigma.params

                def front_end(it_next):
                    b = self._build_window_rays(active, pose_dev, tab, n_out=valid_log[it_next:it_next + 1])
                    b["front"] = self._sample_front(b["rays"], b["depths"], b["n_dev"], pre=b)
                    return b
                batch = front_end(0) if n_it > 0 else None
                for it_idx in range(n_it):
                    out = self._loss_and_grads(batch["rays"], batch["depths"], sp, it_idx, want_ray_grads=any_free,
                                               want_param_grads=not os_.freeze_sigma_mlp, n_rays_dev=batch["n_dev"],
                                               loss_out=loss_log[it_idx], accumulate_into_param_grad=True, want_stats=False,
                                               defer_grad_wait=True, poison=poison, front=batch["front"], input_grad_event=ev,
                                               defer_weight_fold=density_group is not None)
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        # (only when the density step is deferred: it then waits for this stream; a step taken right away on the main
                        # stream must find the folded gradient there)
                        if out["grad_params"] is not None and density_group is not None:
                            # the weight-gradient slabs of the MLP backward, folded here instead of behind the table-gradient reduce
                            ops.density_fold_weight_grads(self._model.nerf_model._model_sigma.spec, out["grad_params"],
                                                          batch["rays"].shape[0] * batch["front"]["z"].shape[1],
                                                          overwrite_grad=self._overwrite_grads)
                        if any_free:
                            self._pose_backward(batch, out["d_rays"], pose_dev, free_rows, poison=poison, poison_tag=it_idx)
                        if groups:
                            for g, lr0 in zip(self._optimizer.param_groups, base_lrs):
                                g['lr'] = lr0 * (gamma ** it_idx)
                            dg = density_group if density_group is not None else \
                                (0 if (sigma_params and not tracking and not os_.freeze_sigma_mlp) else None)
                            self._optimizer.step_now(zero_grad=True, groups=tuple(i for i in range(len(groups)) if i != dg))
                        else:
                            dg = None
                        if pc["ogm"] and self._global_step % pc["occ_every"] == 0:
                            self._step_occupancy_grid()
                        if it_idx + 1 < n_it:
                            batch = front_end(it_idx + 1)
                    if groups and dg is not None:
                        if density_group is None:
                            main.wait_stream(side)                        # (the pose check of this iteration may still mark the failure word)
                            self._step_density(out["grad_work"], dg)      # not deferred: behind the reduce, on the main stream
                        else:
                            self._pending_density = (out["grad_work"], density_group, self._optimizer.param_groups[density_group]['lr'])
                    main.wait_stream(side)
                    self._global_step += 1
                    if self._progress_bar is not None:
                        self._progress_bar.update()
                n_loop = 0
            else:
                n_loop = n_it
            for it_idx in range(n_loop):
                if not self.should_enable_lidar():
                    break
                if active:
                    batch = self._build_window_rays(active, pose_dev, tab, n_out=valid_log[it_idx:it_idx + 1])
                    out = self._loss_and_grads(batch["rays"], batch["depths"], sigma_params[0] if sigma_params else
                                               self._model.nerf_model._model_sigma.params, it_idx, want_ray_grads=any_free,
                                               want_param_grads=not os_.freeze_sigma_mlp, n_rays_dev=batch["n_dev"],
                                               loss_out=loss_log[it_idx], accumulate_into_param_grad=True, want_stats=False,
                                               defer_grad_wait=True, poison=poison, shard_segments=(batch["seg_start"], tab.seg_order_c), pre=batch)
                else:
                    out = self._join_without_rays(sigma_params[0] if sigma_params else None, want_param_grads=not os_.freeze_sigma_mlp)
                if any_free:
                    self._pose_backward(batch, out["d_rays"], pose_dev, free_rows, poison=poison, poison_tag=it_idx)
                if groups:
                    for g, lr0 in zip(self._optimizer.param_groups, base_lrs):
                        g['lr'] = lr0 * (gamma ** it_idx)
                    if density_group is None:
                        dg = 0 if (sigma_params and not tracking and not os_.freeze_sigma_mlp) else None     # stepped right away
                        self._optimizer.step_now(zero_grad=True, groups=tuple(i for i in range(len(groups)) if i != dg))
                        if dg is not None:
                            self._step_density(out["grad_work"], dg)
                    else:
                        # The density step is deferred to just before the next density forward (_flush_density_step): nothing in
                        # between reads the density parameters - pose gradient and pose step, occupancy step, the next batch's ray
                        # build, compaction, counts and sampling - so in the sharded mode all of that runs beside the gradient
                        # all-reduce instead of behind it.  Same arithmetic, same order per parameter group.
                        self._optimizer.step_now(zero_grad=True, groups=tuple(i for i in range(len(groups)) if i != density_group))
                        self._pending_density = (out["grad_work"], density_group, self._optimizer.param_groups[density_group]['lr'])
                if self.should_enable_lidar() and pc["ogm"] and self._global_step % pc["occ_every"] == 0:
                    self._step_occupancy_grid()
                self._global_step += 1
                if profiler is not None:
                    profiler.step()
                if self._progress_bar is not None:
                    self._progress_bar.update()

            self._flush_density_step()
            # ---- one host sync per phase.  The checks the reference makes in every iteration (:368-374,:590) were made on the
            # device, by the kernels of that iteration; the state below is the one the failing iteration started from ----
            if self._dist is not None:
                # A failure on any rank is everybody's.  The guarantee "parameters as the failing iteration began" is a SINGLE-GPU one:
                # the word is per rank until here, so the healthy ranks kept stepping (with the failing rank's non-finite gradient in
                # their all-reduce) - the run is lost either way.  What all ranks agree on is WHICH failure to report: the earliest
                # iteration, with the code that belongs to it (one packed value, so code and iteration cannot come from different ranks).
                poison.copy_(self._dist.earliest_failure(poison))
            # Everything the host needs from the phase travels in ONE device -> host copy (round 3 made eight small ones, each a
            # sync: ~1 ms per phase together with the per-keyframe pose copies, 2 % of a 20-iteration phase): the failure word, the
            # per-iteration live-ray counts and loss terms, the poses, a finite-poses flag and the origin check of the last batch.
            n_log = max(n_it, 1)
            res = self._results_lidar
            if active and n_it and res is not None:
                last = res["rays"]
                n_live_t = res["n_rays_dev"].reshape(1).float() if res["n_rays_dev"] is not None else \
                    torch.full((1,), float(last.shape[0]), device=self._device)
                row = torch.arange(last.shape[0], device=self._device, dtype=torch.float32)[:, None]
                outside = ((last[:, :3].abs() > 1) & (row < n_live_t)).any().reshape(1).float()
            else:
                outside = torch.zeros(1, device=self._device)
            pose_ok = torch.isfinite(pose_dev.detach()).all().reshape(1).float()
            # (the integer words - failure code and iteration, live-ray counts - travel as their BIT PATTERNS inside the float32 buffer,
            # so they are exact whatever their size; the pieces are split by the sizes of the tensors that were concatenated)
            pieces = [poison.view(torch.float32), valid_log.view(torch.float32), loss_log.detach().reshape(-1),
                      pose_dev.detach().reshape(-1).float(), pose_ok, outside]
            host = torch.cat(pieces).cpu().split([p_.numel() for p_ in pieces])
            code, failed_it = (int(v) for v in host[0].view(torch.int32))
            valid_host = host[1].view(torch.int32).float()
            loss_terms_host = host[2].reshape(loss_log.shape).clone()
            pose_host = host[3].reshape(pose_dev.shape)
            pose_finite, origins_outside = bool(host[4][0] != 0), bool(host[5][0] != 0)
            self._poison = None
            self._model.nerf_model.warn_if_clipped(self._device)      # nerf_tcnn.py:70-78, once per phase instead of per forward
            loss_host = loss_terms_host[:, 0]

            def hand_poses_back():
                with torch.no_grad():
                    for k, p_ in enumerate(pose_cpu):
                        if free_list[k]:        # (the others were not stepped: their tensors are left exactly as they are)
                            p_.data.copy_(pose_host[k].to(p_.device))
            if code != 0:
                hand_poses_back()               # the poses of the last good iteration, like every other phase end
                self.last_failure = {"code": code, "iteration": failed_it}
                if code == hip.POISON_NAN_LOSS:
                    raise AssertionError("NaN Loss Encountered")
                if code == hip.POISON_POSE_GRAD:
                    raise RuntimeError("Fatal: Encountered invalid gradient in pose.")
                raise RuntimeError("Fatal: Encountered invalid pose tensor.")
            if n_it and torch.isnan(loss_host).any():
                raise AssertionError("NaN Loss Encountered")
            if any_free and not pose_finite:
                raise RuntimeError("Fatal: Encountered invalid pose tensor.")
            # the reference asserts on every ray build that the ray origins lie inside the world cube (ray_utils.py:301-303);
            # here once per phase, on the last batch (all rays of a keyframe share its origin)
            if origins_outside:
                raise AssertionError("ray origins are outside the world cube")
            hand_poses_back()
            sigma = self._model.nerf_model._model_sigma.params
            if sigma.grad is not None and os_.freeze_sigma_mlp:
                sigma.grad = None
            elif sigma.grad is not None and self._overwrite_grads and n_it:
                sigma.grad.zero_()              # as after the last zero_grad of the phase (one fill per phase instead of one per iteration)
            losses_log.append(loss_host.tolist())
            depth_eps_log.append((loss_terms_host[:, 4] / valid_host.clamp(min=1)).tolist())
            self._depth_eps = depth_eps_log[-1][-1] if n_it else None
            self.last_stats = {"n_valid_rays": int(valid_host.sum().item()), "iterations": n_it, "loss_terms": loss_terms_host}

        if self._settings.debug.log_losses:
            for name, logs in (("losses", losses_log), ("depth_eps", depth_eps_log)):
                d = f"{self._settings.log_directory}/{name}/keyframe_{self._keyframe_count}"
                os.makedirs(d, exist_ok=True)
                for i, log in enumerate(logs):
                    with open(f"{d}/phase_{i}.csv", 'w+') as f:
                        f.write("\n".join(str(v) for v in log))
        return None

    # -------------------------------------------------------------------------------------------
    def _window_tables(self, active):
        """Static per-window segment tables (built once per optimisation phase): per keyframe one lidar segment and,
        when sky segmentation is on, one sky segment (keyframe.py:91-100)."""
        dev = self._device
        n_lidar = self._num_lidar_samples
        n_sky = self._settings.num_samples.sky if self._enable_sky_segmentation else 0
        rr = [float(self._ray_range[0]), float(self._ray_range[1])]
        dirs_l, dist_l, const_l, counts, poses, is_sky, lens = [], [], [], [], [], [], []
        for k, kf in enumerate(active):
            scan = kf.get_lidar_scan()
            dirs, dist = device_scan(scan, dev)
            dirs_l.append(dirs); dist_l.append(dist); const_l.append(0.0); counts.append(n_lidar); poses.append(k)
            is_sky.append(False); lens.append(dirs.shape[1])
            sky = scan.sky_rays
            if n_sky > 0 and sky is not None and sky.nelement() > 0:
                sdirs = sky.detach().to(dev, torch.float32).contiguous()
                dirs_l.append(sdirs); dist_l.append(None); const_l.append(rr[1] + 1.0); counts.append(n_sky); poses.append(k)
                is_sky.append(True); lens.append(sdirs.shape[1])
        tab = ops.WindowTables(dirs_l, dist_l, const_l, counts, poses)
        tab.is_sky, tab.seg_kf, tab.lens, tab.dirs_list = is_sky, poses, lens, dirs_l
        # window position of every segment (lidar before sky inside a keyframe): the order of the single-GPU batch
        order_of = getattr(self, "_active_order", None) or list(range(len(active)))
        tab.seg_order = [2 * order_of[k] + (1 if sky_ else 0) for k, sky_ in zip(poses, is_sky)]
        import ctypes as _C
        tab.seg_order_c = (_C.c_int32 * len(tab.seg_order))(*tab.seg_order)      # (for lnr_shard_front_pack, every iteration)
        tab.fixed_T12 = None           # set per phase: the matrices of the poses that are not optimised (_do_iterate_optimizer)
        tab.seg_kf_dev = torch.tensor(poses, device=dev)
        tab.lidar_seg_mask = torch.tensor([0.0 if s else 1.0 for s in is_sky], device=dev)[:, None]
        tab.has_sky = any(is_sky)
        return tab

    def _draw_window_indices(self, active, tab):
        """Host-injected draws (parity tests) or non-RANDOM strategies; None -> drawn inside the build kernel."""
        strat = self._consts()["strategy"]
        if strat == 'RANDOM' and self._draws is None:
            return None
        out = []
        for s in range(tab.n_seg):
            kf = active[tab.seg_kf[s]]
            count = tab.seg_start_list[s + 1] - tab.seg_start_list[s]
            if tab.is_sky[s]:
                idx = self._draws.sky_index(tab.lens[s], count) if self._draws is not None else \
                    torch.randint(0, tab.lens[s], (count,))
            elif strat == 'RANDOM':
                idx = self._draws.ray_index(tab.lens[s], count)
            elif strat == 'MASK':
                m = kf.get_lidar_scan().mask.nonzero(as_tuple=True)[0]
                idx = m[torch.randint(len(m), (count,))]
            elif strat == 'FIXED':
                idx = torch.arange(count)
            else:
                raise RuntimeError(f"Can't find rays_selection strategy: {strat}")
            out.append(idx.to(self._device))
        return torch.cat(out)

    def _build_window_rays(self, active, pose_dev, tab, n_out=None):
        """optimizer.py:285-340 on the device: pose -> [R|t], index draw + ray build for the whole window in one
        launch, one order-preserving compaction.  n_out: where the live ray count goes (a row of the phase's log)."""
        if tab.fixed_T12 is not None and not tab.any_free:
            T12 = tab.fixed_T12                            # nothing is optimised (e.g. the anchored keyframe of a one-keyframe shard)
        else:
            T12 = ops.pose_forward(pose_dev)
            if tab.fixed_T12 is not None:                  # poses that are not optimised in this phase: the matrix their Pose hands out
                T12 = torch.where(tab.free_col, T12, tab.fixed_T12)
        rr = self._ray_range_f
        index = self._draw_window_indices(active, tab)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if index is None else 0
        rays_c, depths_c, keep, src_c = ops.build_window_rays(tab, T12, rr, self._scale_f, self._shift_f, index=index, seed=seed)
        # (the table's ctypes arrays: built once per phase.)  The compaction launch also delivers what the loss needs next from the compacted
        # batch - the normalisers {#rays, #opaque rays}, or this rank's front record in the sharded loop: one launch instead of two
        if self._dist is not None:
            rays, depths, src, out_seg, n_dev, rec = ops.compact_rays(rays_c, depths_c, keep, src_c, tab.seg_start, n_out=n_out,
                                                                      front=(tab.seg_order_c, self._front_cap))
            return dict(rays=rays, depths=depths, src=src, seg_start=out_seg, n_dev=n_dev, T12=T12, tab=tab, front_record=rec)
        rays, depths, src, out_seg, n_dev, counts = ops.compact_rays(rays_c, depths_c, keep, src_c, tab.seg_start, n_out=n_out, want_counts=True)
        return dict(rays=rays, depths=depths, src=src, seg_start=out_seg, n_dev=n_dev, T12=T12, tab=tab, counts=counts)

    def _pose_backward(self, batch, d_rays, pose_dev, free_mask_u8, poison=None, poison_tag=0):
        """dL/drays -> dL/d[R|t] per segment (HIP) -> dL/dpose6 (HIP, analytic axis-angle Jacobian)."""
        tab = batch["tab"]
        seg_T = batch["T12"] if not tab.has_sky else batch["T12"][tab.seg_kf_dev]
        dT_seg = ops.lidar_rays_backward(d_rays, batch["rays"], batch["src"], batch["seg_start"], tab.dirs_list, seg_T, self._scale_f)
        if tab.has_sky:          # sky rays are built from a detached pose (keyframe.py:93): drop their contribution
            dT = torch.zeros(pose_dev.shape[0], 12, device=self._device).index_add_(0, tab.seg_kf_dev, dT_seg * tab.lidar_seg_mask)
        else:
            dT = dT_seg
        if pose_dev.grad is None:
            pose_dev.grad = ops.pose_backward(pose_dev, dT, mask=free_mask_u8, poison=poison, poison_tag=poison_tag)
        else:
            ops.pose_backward(pose_dev, dT, mask=free_mask_u8, out=pose_dev.grad, accumulate=True, poison=poison, poison_tag=poison_tag)

    # -------------------------------------------------------------------------------------------
    def _loss_config(self, iteration_idx) -> hip.LossConfig:
        lc = self._consts()["loss"]
        cfg = hip.LossConfig()
        cfg.selection = lc["selection"]
        cfg.min_js, cfg.max_js, cfg.js_alpha = lc["min_js"], lc["max_js"], lc["js_alpha"]
        if lc["decay_los"]:
            cfg.los_lambda = max(lc["los_lambda"] * (lc["los_rate"] ** ((self._global_step + 1) / lc["los_steps"])), lc["min_los"])
        else:
            cfg.los_lambda = lc["los_lambda"]
        cfg.depth_lambda = lc["depth_lambda"]
        cfg.min_eps = lc["min_eps"]
        if lc["decay_eps"]:
            cfg.fixed_eps = max(lc["depth_eps"] * (lc["eps_rate"] ** (iteration_idx / lc["eps_steps"])), lc["min_eps"])
        else:
            cfg.fixed_eps = lc["depth_eps"]
        return cfg

    def _sample_front(self, rays, depths, n_rays_dev, draws=None, shard_segments=None, pre=None):
        """The part of an iteration that does not touch the density parameters: loss normalisers and sample depths for `rays`
        (optimizer.py:437-470 up to the network call).  -> dict(counts, front_work, z, seed, far0).  shard_segments (sharded loop):
        (compacted segment starts on the device, window position of every segment) of this rank's batch.  pre: the batch dictionary of
        _build_window_rays, whose compaction launch has already computed the counts / written the front record."""
        draws = draws if draws is not None else self._draws
        pc = self._consts()
        S, perturb, ogm = pc["S"], pc["perturb"], pc["ogm"]
        dev = self._device
        n = rays.shape[0]
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if draws is None else 0
        u1 = u2 = None
        if draws is not None:
            if perturb > 0:
                u1 = draws.jitter(n, S // 2 if ogm else S).to(dev)
            if ogm:
                u2 = draws.pdf(n, S // 2).to(dev)
        # Sharded: the loss divides by the GLOBAL counts and compares every ground-truth depth with far[0], the first ray of the WHOLE
        # batch (optimizer.py:460-461,488-489,569-578).  One all-gather of the ranks' front records {first-ray key, live count, depths}
        # delivers all three (mapping/sharding.py); it is pure latency and runs behind the sampler and the density forward - awaited
        # right before the loss kernel, its first consumer.
        counts = far0 = front_work = None
        if self._dist is not None:
            if shard_segments is not None:
                seg_start, seg_order, cap = shard_segments[0], shard_segments[1], self._front_cap
            else:       # a caller outside the training loop (compute_loss): the batch is one segment at this rank's position
                seg_start, seg_order = torch.tensor([0, n], device=dev, dtype=torch.int32), [self._dist.rank]
                cap = self._dist.max_over_ranks(n)
            rec = pre.get("front_record") if (pre is not None and shard_segments is not None) else None
            if rec is None:
                rec = ops.shard_front_pack(rays, seg_start, seg_order, depths, n_rays_dev, cap)
            front_work = self._dist.gather_front(rec)
        else:
            counts = pre.get("counts") if pre is not None else None
            if counts is None:
                counts = ops.count_opaque(rays, depths, n_rays_dev=n_rays_dev)
        if ogm:
            z = ops.sample_rays_occ(rays, self._occupancy_grid.detach(), S, perturb, u_jitter=u1, u_pdf=u2, seed=seed,
                                    n_rays_dev=n_rays_dev)
        else:
            z = ops.sample_rays_uniform(rays, S, perturb, u_jitter=u1, seed=seed, n_rays_dev=n_rays_dev)
        return dict(counts=counts, front_work=front_work, z=z, seed=seed, far0=far0)

    def _loss_and_grads(self, rays, depths, params, iteration_idx, want_ray_grads, want_param_grads, n_rays_dev=None,
                        loss_out=None, accumulate_into_param_grad=False, draws=None, want_stats=True, defer_grad_wait=False,
                        poison=None, front=None, input_grad_event=None, defer_weight_fold=False, shard_segments=None, pre=None):
        """sample -> density -> fused render+loss(+backward) -> density backward.  rays [N,13], depths [N] on device.
        front: the result of _sample_front for these rays when the caller already ran it (the pipelined training loop);
        input_grad_event: recorded by the density backward as soon as d_rays is complete (ops.density_backward)."""
        draws = draws if draws is not None else self._draws
        pc = self._consts()
        S, noise_std = pc["S"], pc["noise_std"]
        spec = self._model.nerf_model._model_sigma.spec
        dev = self._device
        n = rays.shape[0]
        if front is None:
            front = self._sample_front(rays, depths, n_rays_dev, draws, shard_segments=shard_segments, pre=pre)
        counts, front_work, z, seed, far0 = front["counts"], front["front_work"], front["z"], front["seed"], front["far0"]
        noise = None
        if draws is not None and noise_std > 0:
            noise = (draws.noise(n, S) * noise_std).to(dev)
        p = params.detach()
        self._flush_density_step()            # the previous iteration's (deferred) density Adam step lands here
        sigma = ops.density_forward(spec, p, rays=rays, z=z, n_rays_dev=n_rays_dev)
        if front_work is not None:
            counts, far0 = front_work.wait()
        loss, d_sigma, d_rays, stats, _ = ops.los_loss_fused(sigma, z, rays, depths, self._scale_f, self._loss_config(iteration_idx),
                                                            counts, noise=noise, noise_std=noise_std, seed=seed + 1,
                                                            n_rays_dev=n_rays_dev, want_stats=want_stats, loss_out=loss_out, far0=far0,
                                                            poison=poison, poison_tag=iteration_idx, zero_dead_rows=not defer_grad_wait)
        grad_params = None
        grad_work = None
        if want_param_grads or want_ray_grads:
            if not want_param_grads:
                grad_params = None            # frozen parameters (tracking / freeze_sigma phases): input gradient only - no table-
                                              # gradient records, no reduce, no 30 MB zero buffer (lnr_density_backward, grad_params NULL)
            elif accumulate_into_param_grad:
                if params.grad is None:
                    params.grad = torch.zeros_like(params)
                grad_params = params.grad
            else:
                grad_params = torch.zeros_like(p)
            # the point gradient is reduced per ray inside the backward and added to d_rays (no [N,S,3] tensor)
            # the training loop steps after every backward: the gradient buffer then RECEIVES the gradient (no read in the reduce, no
            # zeroing in the optimiser: 60 MB of HBM traffic per iteration); every other caller accumulates, as autograd expects
            overwrite = self._overwrite_grads and defer_grad_wait and accumulate_into_param_grad and grad_params is not None
            ops.density_backward(spec, p, d_sigma, grad_params, rays=rays, z=z, n_rays_dev=n_rays_dev,
                                 reuse_features=True, d_rays=d_rays if want_ray_grads else None, input_grad_event=input_grad_event,
                                 defer_weight_fold=defer_weight_fold and grad_params is not None, overwrite_grad=overwrite)
            if self._dist is not None and want_param_grads:
                # only the training loop (defer_grad_wait) steps a slice and gathers the parameters; every other caller
                # (compute_loss -> autograd -> an optimiser of its own) gets the whole sum whatever the exchange form
                # (overwrite mode: the next backward stores every float of the gradient - the chunks of other ranks need no zeroing)
                grad_work = self._dist.exchange_grads(grad_params, async_op=True, force_all_reduce=not defer_grad_wait,
                                                      zero_rest=not overwrite)
                if not defer_grad_wait:
                    grad_work.wait()
                    grad_work = None
        elif input_grad_event is not None:
            input_grad_event.record()
        self._results_lidar = {"rays": rays, "depths": depths, "samples_fine": z, "n_rays_dev": n_rays_dev, "stats": stats}
        return dict(loss=loss, d_rays=d_rays if want_ray_grads else None,
                    grad_params=grad_params if want_param_grads else None, stats=stats, z=z, grad_work=grad_work)

    def _join_without_rays(self, params, want_param_grads):
        """Sharded mode, this rank owns no keyframe of the active window: take part in every collective of an iteration
        (far[0] broadcast, loss normalisers, density gradient) with zero contributions, in the order _loss_and_grads issues
        them, so that the other ranks neither block nor see different sums."""
        dev = self._device
        front_work = self._dist.gather_front(ops.shard_front_pack(None, None, (), None, None, self._front_cap, device=dev))
        self._flush_density_step()
        front_work.wait()
        grad_work = None
        if want_param_grads and params is not None:
            if params.grad is None:
                params.grad = torch.zeros_like(params)
            elif self._overwrite_grads:
                # no backward of this rank overwrites what the previous exchange left in the buffer: contribute zeros again.  The
                # all-reduce form leaves the sum in the whole vector; the reduce-scatter form only reads the buffer and writes this
                # rank's chunk (and the phase starts from an all-zero gradient: _do_iterate_optimizer's phase end).
                # The rest of the buffer must be zero as well - it is after a phase that ended normally, but not after an autograd
                # backward through compute_loss() without a zero_grad, nor after anything else that wrote params.grad in between
                #  - so the first idle iteration of a phase clears the whole vector: one fill per phase, not per iteration.
                sl = self._dist.owned_range(params.grad.numel())
                if sl is None or not self._idle_grad_cleared:
                    params.grad.zero_()
                    self._idle_grad_cleared = True
                else:
                    params.grad.view(-1)[sl[0]:sl[1]].zero_()
            grad_work = self._dist.exchange_grads(params.grad.view(-1), async_op=True, zero_rest=not self._overwrite_grads)
        self._results_lidar = None
        return dict(loss=None, d_rays=None, grad_params=params.grad if (want_param_grads and params is not None) else None,
                    stats=None, z=None, grad_work=grad_work)

    def _flush_density_step(self):
        """Apply the density Adam step the training loop deferred (after its gradient all-reduce, if any, has finished)."""
        if self._pending_density is None:
            return
        work, group, lr = self._pending_density
        self._pending_density = None
        self._optimizer.param_groups[group]['lr'] = lr
        self._step_density(work, group)

    def _step_density(self, work, group):
        """Adam step of the density parameters once their gradient exchange (if any) has finished.  Sharded with the
        "reduce_scatter" exchange a rank steps ITS chunk of the flat parameter vector, then the chunks are gathered."""
        if work is not None:
            work.wait()
        ranges = None
        if self._dist is not None:
            sl = self._dist.owned_range(self._model.nerf_model._model_sigma.params.numel())
            if sl is not None:
                ranges = [sl]
        # (overwrite mode: the next backward stores its gradient over this one - nothing to zero)
        self._optimizer.step_now(zero_grad=not self._overwrite_grads, groups=(group,), ranges=ranges)
        if ranges is not None:
            self._dist.gather_params(self._model.nerf_model._model_sigma.params.data.view(-1))

    def compute_loss(self, camera_samples: Tuple[torch.Tensor, torch.Tensor], lidar_samples: Tuple[torch.Tensor, torch.Tensor],
                     iteration_idx: int, override_enables: bool = False, tracking=False) -> torch.Tensor:
        """Differentiable lidar loss for (rays [N,13], depths [N]) - optimizer.py:437-595.  The returned scalar
        back-propagates into the density parameters and into `rays` (hence into keyframe poses)."""
        if not ((override_enables or self.should_enable_lidar()) and lidar_samples is not None):
            print("Warning: zero loss")
            return torch.zeros((), device=self._device)
        rays, depths = lidar_samples
        rays = rays.reshape(-1, rays.shape[-1])
        depths = depths.reshape(-1).to(self._device).float().contiguous()
        self._pc = None                  # a public entry point: re-read the configuration (the training loop caches it per phase)
        params = self._model.nerf_model._model_sigma.params
        rays_dev = rays if rays.device == self._device else rays.to(self._device)
        loss = _LidarLossFn.apply(rays_dev, params, self, depths, iteration_idx)
        stats = self._results_lidar["stats"]
        self._results_lidar.update(depth_fine=stats[:, 0], opacity_fine=stats[:, 1], variance=stats[:, 2])
        self._depth_eps = float(stats[:, 6].mean().item())
        assert not torch.isnan(loss), "NaN Loss Encountered"
        return loss

    def compute_loss_api(self, lidar_samples: Tuple[torch.Tensor, torch.Tensor], iteration_idx: int) -> torch.Tensor:
        """API-parity mode of compute_loss: the loss evaluated the way the reference evaluates it (optimizer.py:437-595) -
        Model.forward -> the result dictionary with its [N,S] weights / samples and [N,S,3] points materialised in HBM (model_tcnn.py:70-105)
        -> torch ops on that dictionary -> torch autograd back through render_rays.  Same value and gradients as compute_loss, which
        computes all of it in one fused kernel pass without the dictionary; this is the form a caller gets who keeps the reference's own
        loss code on top of this package's Model, and the one bench.py times as `api_parity_mode`."""
        lc = self._model_config.loss
        self._pc = None
        rays, depths = lidar_samples
        rays = rays.reshape(-1, rays.shape[-1])
        rays = rays if rays.device == self._device else rays.to(self._device)
        scale = self._scale_f
        g = depths.reshape(-1, 1).to(self._device).float() * scale                          # ground-truth depths, metres
        # the reference's (N,1) > (N,) broadcast followed by [..., 0]: every depth is compared with far of the FIRST ray (:460-461)
        transparent = depths.reshape(-1).to(self._device) > rays[0, 12].detach()
        opaque = (depths.reshape(-1).to(self._device) > 0) & ~transparent
        res = self._model(rays, self._ray_sampler, self._scale_factor, camera=False, return_variance=True)
        s = res["samples_fine"] * scale                                                      # sample depths, metres
        w = res["weights_fine"]
        w_sum = w.sum(1) + 1e-10
        mean = (s * w).sum(1) / w_sum
        var = ((s - mean[:, None]) ** 2 * w).sum(1) / w_sum + 1e-10
        std = var.sqrt()
        js = self.calculate_JS_divergence(g, lc.min_depth_eps / 3., mean[:, None], std[:, None]).reshape(-1)
        loss = lc.depthloss_lambda * torch.nn.functional.mse_loss((res["depth_fine"] * scale)[opaque], g[opaque, 0])
        cfg = self._loss_config(iteration_idx)
        if lc.loss_selection in ("L1_JS", "L2_JS"):
            js = torch.where(js < lc.JS_loss.min_js_score, torch.zeros_like(js), js).clamp(max=lc.JS_loss.max_js_score)
            eps = (lc.min_depth_eps * (1 + lc.JS_loss.alpha * js))[:, None].detach()
            self._depth_eps = float(eps.mean().item())
        else:
            eps = cfg.fixed_eps
            self._depth_eps = eps
        from ..models.losses import get_weights_gt
        w_gt = get_weights_gt(s.detach(), g, eps)
        w_gt = w_gt * opaque[:, None]
        los = (w - w_gt).abs().mean() if lc.loss_selection in ("L1_JS", "L1_LOS") else ((w - w_gt) ** 2).mean()
        loss = loss + cfg.los_lambda * los + (res["opacity_fine"][opaque] - 1).abs().mean()
        res.update(rays=rays.detach(), depths=depths.reshape(-1).to(self._device).float(), n_rays_dev=None, stats=None)
        self._results_lidar = res
        assert not torch.isnan(loss), "NaN Loss Encountered"
        return loss

    # -------------------------------------------------------------------------------------------
    def _step_occupancy_grid(self):
        """optimizer.py:598-609; must follow a loss evaluation (uses its rays / samples / depths)."""
        res = self._results_lidar
        if res is None and self._dist is None:
            raise RuntimeError("_step_occupancy_grid called before compute_loss")
        occ = self._model_config.model.occ_model
        grid = self._occupancy_grid_model.occupancy_grid
        # the pseudo-gradient is accumulated in 64-bit fixed point (exact integer atomics: the step is reproducible, and in the
        # sharded mode the ranks' accumulators add up exactly in one integer all-reduce), then applied to the logits
        if self._grad_buf is None:
            self._grad_buf = torch.zeros(grid.numel(), device=grid.device, dtype=torch.int64)
        if res is not None:            # (None: a rank without rays in the sharded mode contributes zeros to the all-reduce)
            ops.occ_grid_step(grid.data, res["rays"], res["samples_fine"], res["depths"], self._scale_f, occ.lr,
                              grad_buf=self._grad_buf, n_rays_dev=res["n_rays_dev"])
        if self._dist is not None:
            self._dist.all_reduce_grads(self._grad_buf)
        ops.occ_grid_apply(grid.data, self._grad_buf, occ.lr, zero_grad=True, poison=self._poison)
        self._occupancy_grid = self._occupancy_grid_model()
        self._ray_sampler.update_occ_grid(self._occupancy_grid.detach())

    def calculate_KL_divergence(self, mean1, std1, mean2, std2):
        var1, var2 = std1 * std1, std2 * std2
        return torch.log(std2 / std1) + (var1 + (mean1 - mean2) ** 2) / (2 * var2) - 0.5

    def calculate_JS_divergence(self, mean1, std1, mean2, std2):
        mean_m = 0.5 * (mean1 + mean2)
        std_m = 0.5 * torch.sqrt(std1 ** 2 + std2 ** 2)
        return 0.5 * self.calculate_KL_divergence(mean1, std1, mean_m, std_m) + \
            0.5 * self.calculate_KL_divergence(mean2, std2, mean_m, std_m)
