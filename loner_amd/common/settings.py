"""Settings container + the default configuration of the mapping hot path.

The reference loads YAML into an attrdict-based `Settings` (src/common/settings.py:51-75).  Only
the *keys* the hot path reads are part of the boundary; `Settings` below
gives the same access patterns (attribute access, item access, nested dicts become Settings, lists
become tuples on attribute access) so that either this class or the reference's own object can be
handed to `Optimizer`.  `default_optimizer_settings()` restates the values of cfg/defaults.yaml
(mapper.optimizer), cfg/model_config/default_model_config.yaml and
cfg/nerf_config/default_nerf_hash.yaml.
"""
import copy
import os

import yaml


class Settings(dict):
    def __getattr__(self, key):
        try:
            value = self[key]
        except KeyError:
            raise AttributeError(key)
        return self._wrap(value)

    def __setattr__(self, key, value):
        self[key] = value

    @classmethod
    def _wrap(cls, value):
        if isinstance(value, dict) and not isinstance(value, Settings):
            return cls(value)
        if isinstance(value, list):
            return tuple(cls._wrap(v) for v in value)
        return value

    def __deepcopy__(self, memo):
        return Settings(copy.deepcopy(dict(self), memo))

    @staticmethod
    def load_from_file(filename: str) -> "Settings":
        class _Loader(yaml.SafeLoader):
            pass

        root = os.path.dirname(os.path.abspath(filename))

        def _include(loader, node):
            with open(os.path.join(root, loader.construct_scalar(node))) as f:
                return yaml.load(f, _Loader)
        _Loader.add_constructor("!include", _include)
        with open(filename) as f:
            return Settings(yaml.load(f, _Loader))


DEBUG_FLAGS = ("pytorch_detect_anomaly", "draw_comp_graph", "draw_rays_eps", "write_ray_point_clouds", "store_ray",
               "write_frame_point_clouds", "write_icp_point_clouds", "profile_optimizer", "use_groundtruth_poses",
               "draw_loss_distribution", "log_losses", "profile", "draw_samples", "visualize_loss", "log_times")


def default_nerf_config() -> dict:
    return {
        "enable_view_dependence": True,
        "dir_encoding_intensity": {"degree": 4, "otype": "SphericalHarmonics"},
        "intensity_network": {"activation": "ReLU", "n_hidden_layers": 4, "n_neurons": 64, "otype": "FullyFusedMLP",
                              "output_activation": "None"},
        "pos_encoding_intensity": {"base_resolution": 16, "log2_hashmap_size": 19, "n_features_per_level": 2,
                                   "n_levels": 16, "otype": "HashGrid"},
        "pos_encoding_sigma": {"base_resolution": 16, "log2_hashmap_size": 18, "n_features_per_level": 2,
                               "n_levels": 16, "otype": "HashGrid"},
        "sigma_network": {"activation": "ReLU", "n_hidden_layers": 1, "n_neurons": 64, "otype": "FullyFusedMLP",
                          "output_activation": "None"},
    }


def default_model_config(ray_range=(1, 50)) -> dict:
    return {
        "data": {"ray_range": list(ray_range)},
        "model": {
            "num_colors": 3, "model_type": "nerf_decoupled", "nerf_config": default_nerf_config(),
            "ray_range": list(ray_range),
            "render": {"N_samples_train": 512, "N_samples_test": 2048, "retraw": True, "lindisp": False, "perturb": 1.0,
                       "white_bkgd": False, "raw_noise_std": 1.0, "chunk": 16384, "netchunk": 0},
            "occ_model": {"voxel_size": 100, "lr": 0.0001, "N_iters_acc": 10},
        },
        "train": {"lrate_sigma_mlp": 0.01, "lrate_rgb": 0.01, "lrate_pose": 0.001, "lrate_gamma": 1.0, "decay_rate": 0.001,
                  "pose_lrate_gamma": 1.0, "rgb_weight_decay": 1e-5, "sigma_weight_decay": 0.0},
        "loss": {"loss_selection": "L1_JS", "JS_loss": {"min_js_score": 1.0, "max_js_score": 10.0, "alpha": 1.0},
                 "decay_los_lambda": False, "los_lambda": 1000.0, "min_los_lambda": 10.0, "los_lambda_decay_rate": 0.001,
                 "los_lambda_decay_steps": 15000, "decay_depth_eps": True, "depth_eps": 3.0, "min_depth_eps": 0.5,
                 "depth_eps_decay_rate": 0.95, "depth_eps_decay_steps": 1, "depthloss_lambda": 0.005},
    }


def default_optimizer_settings(ray_range=(1, 50), log_directory="/tmp/loner_amd_logs") -> Settings:
    """= settings.mapper.optimizer after Mapper.__init__ added debug and log_directory (mapper.py:60-61)."""
    return Settings({
        "freeze_poses": False, "data_prep_on_cpu": True, "enabled": True, "detach_rgb_from_poses": True,
        "detach_rgb_from_sigma": False, "skip_pose_refinement": True,
        "num_samples": {"lidar": 512, "sky": 64},
        "rays_selection": {"strategy": "RANDOM"},
        "samples_selection": {"strategy": "OGM"},
        "keyframe_schedule": [
            {"num_keyframes": 1, "iteration_schedule": [
                {"num_iterations": 1000, "freeze_poses": True, "freeze_sigma_mlp": False, "freeze_rgb_mlp": True}]},
            {"num_keyframes": -1, "iteration_schedule": [
                {"num_iterations": 50, "freeze_poses": False, "latest_kf_only": True, "freeze_sigma_mlp": True, "freeze_rgb_mlp": True},
                {"num_iterations": 50, "freeze_poses": False, "freeze_sigma_mlp": False, "freeze_rgb_mlp": True}]},
        ],
        "model_config": default_model_config(ray_range),
        "debug": {k: False for k in DEBUG_FLAGS},
        "log_directory": log_directory,
    })
