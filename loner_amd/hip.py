"""ctypes binding of libloner_hip.so (include/loner_hip.h).

PyTorch is used only for device memory and streams: tensors are handed to the C ABI as
``data_ptr()`` + sizes + the current HIP stream.  There is NO fallback: if the library is missing
or a call fails, a RuntimeError is raised (reference convention: optimizer.py:119,296,370,374).

The library is loaded lazily so that objects holding a `Hip` handle stay picklable across the
reference's `mp.set_start_method('spawn')` process boundary (src/loner.py:59,188,205).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LNR_LIB_PATH: the experiment scripts under tools/ point this at development builds, e.g. -DLNR_ABLATE)
LIB_PATH = os.environ.get("LNR_LIB_PATH") or os.path.join(_HERE, "_lib", "libloner_hip.so")

MAX_LEVELS = 32
RAY_STRIDE = 13
LOSS_RAYS_PER_BLOCK = 4      # LNR_LOSS_RAYS_PER_BLOCK
FRONT_HEADER = 4             # LNR_FRONT_HEADER

ENCODINGS = {"HashGrid": 0, "Grid": 0, "Frequency": 1}
ACTIVATIONS = {"None": 0, "ReLU": 1, "Sine": 2, "LeakyReLU": 3, "Exponential": 4, "Sigmoid": 5,
               "Squareplus": 6, "Softplus": 7, "Tanh": 8}
LOSS_SELECTIONS = {"L1_JS": 0, "L2_JS": 1, "L1_LOS": 2, "L2_LOS": 3}
PRECISIONS = {"fp32": 0, "float32": 0, "fp16": 1, "half": 1, "float16": 1, "fp32_chain": 2}
POS_ROUNDINGS = {"fma": 0, "mul_add": 1}
BWD_TABLE_ATOMICS = 1        # LNR_BWD_TABLE_ATOMICS
BWD_REPORT_REGIONS = 2       # LNR_BWD_REPORT_REGIONS
BWD_DEFER_WEIGHT_FOLD = 16   # LNR_BWD_DEFER_WEIGHT_FOLD
BWD_BINS = 4                 # LNR_BWD_BINS
BWD_BINS_W8 = 8              # LNR_BWD_BINS_W8
BWD_OVERWRITE_GRAD = 32      # LNR_BWD_OVERWRITE_GRAD
WORKSPACE_STATUS_BYTES, STATUS_CLIPPED = 256, 0            # LNR_WORKSPACE_STATUS_BYTES, LNR_STATUS_CLIPPED
DRAW_JITTER, DRAW_PDF, DRAW_NOISE, DRAW_RAY_INDEX = 0, 1, 2, 16      # LNR_DRAW_*
POISON_NAN_LOSS, POISON_POSE_GRAD, POISON_POSE = 1, 2, 3      # LNR_POISON_* codes of the failure guard (int32[2] device word)


class NetSpec(C.Structure):
    _fields_ = [("encoding", C.c_int32), ("n_levels", C.c_int32), ("n_features", C.c_int32),
                ("log2_table", C.c_int32), ("base_res", C.c_int32), ("per_level_scale", C.c_float),
                ("n_frequencies", C.c_int32), ("activation", C.c_int32), ("n_neurons", C.c_int32),
                ("n_hidden", C.c_int32), ("precision", C.c_int32), ("pos_rounding", C.c_int32),
                ("enc_dim", C.c_int32), ("in_dim", C.c_int32), ("n_mlp_params", C.c_int32),
                ("n_params", C.c_int64),
                ("level_scale", C.c_float * MAX_LEVELS), ("level_res", C.c_uint32 * MAX_LEVELS),
                ("level_size", C.c_uint32 * MAX_LEVELS), ("level_offset", C.c_uint32 * MAX_LEVELS),
                ("level_hashed", C.c_uint32 * MAX_LEVELS)]


class LossConfig(C.Structure):
    _fields_ = [("selection", C.c_int32), ("min_js", C.c_float), ("max_js", C.c_float), ("js_alpha", C.c_float),
                ("los_lambda", C.c_float), ("depth_lambda", C.c_float), ("min_eps", C.c_float),
                ("fixed_eps", C.c_float)]


P = C.c_void_p
_SIGNATURES = {
    "lnr_last_error": (C.c_char_p, []),
    "lnr_version": (C.c_int, []),
    "lnr_profile_enable": (C.c_int, [C.c_int32]),
    "lnr_profile_read": (C.c_int, [P, C.c_int32, P, P, C.c_int32]),
    "lnr_net_spec_finalize": (C.c_int, [C.POINTER(NetSpec)]),
    "lnr_density_workspace": (C.c_size_t, [C.POINTER(NetSpec), C.c_int64]),
    "lnr_density_workspace_forward": (C.c_size_t, [C.POINTER(NetSpec), C.c_int64]),
    "lnr_density_workspace_init": (C.c_int, [P, C.c_size_t, P]),
    "lnr_density_workspace_release": (C.c_int, [P]),
    "lnr_density_forward": (C.c_int, [C.POINTER(NetSpec), P, P, C.c_int64, P, P, C.c_int32, C.c_int32, P, P, P, C.c_size_t, P]),
    "lnr_density_backward": (C.c_int, [C.POINTER(NetSpec), P, P, C.c_int64, P, P, C.c_int32, C.c_int32, P, P, P, P, P,
                                       C.c_int32, C.c_int32, P, C.c_size_t, P, P]),
    "lnr_density_fold_weight_grads": (C.c_int, [C.POINTER(NetSpec), C.c_int64, P, P, C.c_size_t, C.c_int32, P]),
    "lnr_build_lidar_rays": (C.c_int, [P, P, C.c_int64, P, C.c_int32, P, C.c_float, C.c_float, C.c_float,
                                       C.POINTER(C.c_float), P, P, P, P]),
    "lnr_build_window_rays": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, P, P, C.c_uint64, P, C.c_float, C.c_float,
                                        C.c_float, C.POINTER(C.c_float), P, P, P, P]),
    "lnr_pose_forward": (C.c_int, [P, C.c_int32, P, P]),
    "lnr_pose_backward": (C.c_int, [P, P, P, C.c_int32, P, C.c_int32, P, C.c_int32, P]),
    "lnr_compact_rays": (C.c_int, [P, P, P, P, C.c_int32, C.POINTER(C.c_int32), C.c_int32, P, P, P, P, P, P]),
    "lnr_compact_rays_front": (C.c_int, [P, P, P, P, C.c_int32, C.POINTER(C.c_int32), C.c_int32, P, P, P, P, P, P, C.POINTER(C.c_int32), C.c_int32, P, P]),
    "lnr_lidar_rays_backward": (C.c_int, [P, P, P, P, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), P,
                                          C.c_float, P, P]),
    "lnr_occ_interpolate": (C.c_int, [P, C.c_int32, P, C.c_int64, P, P]),
    "lnr_sample_rays_occ": (C.c_int, [P, C.c_int32, P, P, C.c_int32, C.c_int32, C.c_float, P, P, P, C.c_uint64,
                                      P, P, P, P, P]),
    "lnr_sample_rays_uniform": (C.c_int, [P, C.c_int32, P, C.c_int32, C.c_float, P, P, C.c_uint64, P, P]),
    "lnr_render_forward": (C.c_int, [P, P, P, C.c_int32, P, C.c_int32, P, C.c_float, C.c_uint64, P, P, P, P, P]),
    "lnr_render_backward": (C.c_int, [P, P, P, C.c_int32, P, C.c_int32, P, C.c_float, C.c_uint64, P, P, P, P, P, P, P]),
    "lnr_render_ftb_gather": (C.c_int, [P, P, C.c_int32, P, P, C.c_int32, C.c_int32, C.c_int32, P, P, P, P]),
    "lnr_render_ftb_composite": (C.c_int, [P, P, P, C.c_int32, P, P, C.c_int32, C.c_int32, C.c_int32, P, C.c_float, C.c_uint64, P, P, P, P, P, C.c_int32, P]),
    "lnr_points_grad_to_rays": (C.c_int, [P, P, C.c_int32, P, C.c_int32, P, P]),
    "lnr_weights_gt": (C.c_int, [P, P, P, C.c_float, C.c_int32, C.c_int32, C.c_int32, P, P]),
    "lnr_logits_grad": (C.c_int, [P, P, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, P, P]),
    "lnr_count_opaque": (C.c_int, [P, P, C.c_int32, P, P, P, P]),
    "lnr_los_loss_fused": (C.c_int, [P, P, P, P, C.c_int32, P, C.c_int32, P, C.c_float, C.c_uint64, C.c_float,
                                     C.POINTER(LossConfig), P, P, P, P, P, P, P, P, P, C.c_int32, P]),
    "lnr_adam_step": (C.c_int, [P, P, P, P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                C.c_float, C.c_int32, P, P]),
    "lnr_occ_grid_step": (C.c_int, [P, C.c_int32, P, P, P, C.c_int32, P, C.c_int32, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_float, P, P]),
    "lnr_occ_grid_apply": (C.c_int, [P, P, C.c_int64, C.c_float, C.c_int32, P, P]),
    "lnr_selftest_mfma": (C.c_int, [P, P]),
    "lnr_first_ray_key": (C.c_int, [P, P, C.POINTER(C.c_int32), C.c_int32, P, P]),
    "lnr_rng_draws": (C.c_int, [C.c_int32, C.c_uint64, C.c_int32, C.c_int32, P, P]),
    "lnr_shard_front_pack": (C.c_int, [P, P, C.POINTER(C.c_int32), C.c_int32, P, C.c_int32, P, C.c_int32, P, P]),
    "lnr_shard_front_reduce": (C.c_int, [P, C.c_int32, C.c_int32, P, P, P]),
    "lnr_comm_available": (C.c_int, []),
    "lnr_comm_unique_id": (C.c_int, [P, C.c_size_t]),
    "lnr_comm_init": (C.c_int, [P, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "lnr_comm_destroy": (C.c_int, [P]),
    "lnr_comm_all_reduce": (C.c_int, [P, P, C.c_size_t, C.c_int32, C.c_int32, P]),
    "lnr_comm_reduce_scatter": (C.c_int, [P, P, P, C.c_size_t, C.c_int32, P]),
    "lnr_comm_all_gather": (C.c_int, [P, P, P, C.c_size_t, P]),
    "lnr_comm_broadcast": (C.c_int, [P, P, C.c_size_t, C.c_int32, P]),
}
COMM_ID_BYTES = 128
COMM_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.int64: 2, torch.int32: 3, torch.uint8: 4}
COMM_OPS = {"sum": 0, "min": 1, "max": 2}

_lib = None


def declared_symbols():
    return sorted(_SIGNATURES)


def load():
    """Load the shared library (idempotent).  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found - build it with `python -m loner_amd.build` "
                               "(there is no CPU fallback for the MI355X hot path)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _ptr(t):
    """device pointer of a tensor (None -> NULL); checks dtype-agnostic contiguity."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise RuntimeError("loner_amd: tensor passed to the HIP library must be contiguous")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _stream(device=None):
    """torch's current stream on `device` (default: the current device) as a hipStream_t.  Called once per library call, ~25 times
    per mapping iteration: torch.cuda.current_stream() builds a Stream object (8 us), the raw query is a plain C call."""
    if _raw_stream is not None and (device is None or isinstance(device, int)):
        return C.c_void_p(_raw_stream(_raw_device() if device is None else device))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def check(rc, what=""):
    if rc != 0:
        msg = load().lnr_last_error().decode(errors="replace")
        raise RuntimeError(f"loner_amd HIP call failed ({what}, status {rc}): {msg}")


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("loner_amd: expected a tensor on the MI355X (cuda/hip device), got "
                               f"{t.device}; the hot path has no CPU implementation")


def make_net_spec(encoding_config: dict, network_config: dict) -> NetSpec:
    """tinycudann-style (encoding_config, network_config) -> finalized NetSpec
    (schema of cfg/nerf_config/default_nerf_hash.yaml in the reference)."""
    s = NetSpec()
    otype = str(encoding_config.get("otype", "HashGrid"))
    if otype not in ENCODINGS:
        raise RuntimeError(f"unsupported encoding otype {otype!r} (supported: {sorted(ENCODINGS)})")
    s.encoding = ENCODINGS[otype]
    s.n_levels = int(encoding_config.get("n_levels", 16))
    s.n_features = int(encoding_config.get("n_features_per_level", 2))
    s.log2_table = int(encoding_config.get("log2_hashmap_size", 19))
    s.base_res = int(encoding_config.get("base_resolution", 16))
    s.per_level_scale = float(encoding_config.get("per_level_scale", 2.0))
    s.n_frequencies = int(encoding_config.get("n_frequencies", 12))
    act = str(network_config.get("activation", "ReLU"))
    if act not in ACTIVATIONS:
        raise RuntimeError(f"unsupported activation {act!r}")
    out_act = str(network_config.get("output_activation", "None"))
    if out_act != "None":
        raise RuntimeError(f"output_activation {out_act!r} is not supported (the reference uses None)")
    s.activation = ACTIVATIONS[act]
    s.n_neurons = int(network_config.get("n_neurons", 64))
    s.n_hidden = int(network_config.get("n_hidden_layers", 1))
    # extensions to the tinycudann schema (absent keys = defaults): arithmetic of the network, grid-position rounding
    prec = str(network_config.get("precision", "fp32"))
    if prec not in PRECISIONS:
        raise RuntimeError(f"unsupported precision {prec!r} (supported: {sorted(PRECISIONS)})")
    s.precision = PRECISIONS[prec]
    pos = str(encoding_config.get("pos_rounding", "fma"))
    if pos not in POS_ROUNDINGS:
        raise RuntimeError(f"unsupported pos_rounding {pos!r} (supported: {sorted(POS_ROUNDINGS)})")
    s.pos_rounding = POS_ROUNDINGS[pos]
    check(load().lnr_net_spec_finalize(C.byref(s)), "lnr_net_spec_finalize")
    return s
