/*
 * loner_hip.h -- C ABI of libloner_hip.so, the MI355X (gfx950) implementation of the
 * LONER mapping-thread hot path.
 *
 * The reference (umautobots/LONER) is pure Python: its "native" layer on this path is
 * torch ops plus the tinycudann CUDA extension, reached through Python classes.  It has
 * no FFI of its own, so every entry point below cites the reference *function* it
 * replaces (paths relative to the reference root).  The Python classes that mirror the
 * reference's interface (loner_amd/models/..., loner_amd/mapping/...) bind these symbols
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - tensors are dense, row-major, float32 unless stated; the caller owns all buffers;
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*);
 *   - return value 0 = success, negative = LnrStatus error (nothing was enqueued);
 *   - the library keeps no global state and is re-entrant per stream;
 *   - `n_rays_dev` (nullable): if non-null the kernels read the live ray count from
 *     device memory (<= the `n_rays` capacity given by value) so that a window whose
 *     ray count depends on data (rays dropped by the cube test) needs no host sync.
 *
 * Ray record (13 floats): [origin(3) dir(3) viewdir(3) 0 0 near far]  (ray_utils.py:307-310)
 */
#ifndef LONER_HIP_H
#define LONER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LNR_RAY_STRIDE 13
#define LNR_LOSS_RAYS_PER_BLOCK 4
#define LNR_MAX_LEVELS 32

typedef enum LnrStatus {
    LNR_OK = 0,
    LNR_ERR_INVALID_ARG = -1,
    LNR_ERR_UNSUPPORTED = -2,   /* legal config the kernels do not cover (message via lnr_last_error) */
    LNR_ERR_LAUNCH = -3,        /* hipGetLastError() != hipSuccess after a launch */
    LNR_ERR_WORKSPACE = -4      /* workspace too small */
} LnrStatus;

typedef enum LnrEncoding { LNR_ENC_HASHGRID = 0, LNR_ENC_FREQUENCY = 1 } LnrEncoding;

/* Arithmetic of the density network.  The reference runs tinycudann in half precision (fp16 parameters copy, fp16 encoded
 * features, FullyFusedMLP on tensor cores: cfg/nerf_config/default_nerf_hash.yaml:20-31, src/models/nerf_tcnn.py:35-38).
 *   LNR_PREC_F32: everything in fp32; stricter than the reference, the default.  Matrix products of the reference's default shape
 *                 class (32 encoded features -> <= 64 ReLU neurons -> 1) run on the bf16 matrix pipe with every fp32 operand split
 *                 into three bf16 terms (x = x0 + x1 + x2 exactly) and the six largest partial products accumulated in fp32:
 *                 error per product below 2^-24 relative (the order of fp32's own rounding; exact for operands of <= 16 significant
 *                 bits), ~2.5x the speed of the fp32 MFMA.  Every other network: v_mfma_f32_16x16x4_f32 (exact fp32 fma chains).
 *   LNR_PREC_F32_CHAIN: fp32 with exact fma chains (v_mfma_f32_16x16x4_f32) for every network, the default class included.
 *   LNR_PREC_F16: the reference's storage types - encoded features and MLP weights rounded to fp16, matrix products on
 *                 v_mfma_f32_16x16x32_f16 with fp32 accumulation (the reference accumulates in fp16), fp32 master
 *                 parameters and fp32 gradients (the reference: fp16 atomics with a loss scale of 128). */
typedef enum LnrPrecision { LNR_PREC_F32 = 0, LNR_PREC_F16 = 1, LNR_PREC_F32_CHAIN = 2 } LnrPrecision;

/* Grid position of the hash-grid lookup, pos = x*scale + 0.5:  LNR_POS_FMA rounds once (tiny-cuda-nn's fmaf, the
 * default), LNR_POS_MUL_ADD rounds the product and the sum separately (what un-contracted code would do).  The two
 * differ by one fp32 ulp of pos = 1/32 cell on the finest default level. */
typedef enum LnrPosRounding { LNR_POS_FMA = 0, LNR_POS_MUL_ADD = 1 } LnrPosRounding;

/* Failure guard.  The reference checks, in EVERY iteration, the loss for NaN (inside compute_loss, optimizer.py:590) and - after
 * backward, before optimizer.step() - every pose gradient and pose tensor for non-finite values (optimizer.py:368-374); an
 * exception leaves the step of that iteration unreached.  Without a host sync per iteration the same contract is kept on the
 * device: `poison_dev` (nullable everywhere) is an int32[2] word {code, tag}, zeroed by the caller at the start of a phase.
 * lnr_los_loss_fused and lnr_pose_backward set it (first event wins; tag = the caller's iteration index) and lnr_adam_step /
 * lnr_occ_grid_apply do nothing once it is non-zero, so parameters, poses and the occupancy grid stay at the values they had
 * when the failing iteration began; the host reads the word once per phase and raises the reference's error. */
#define LNR_POISON_NAN_LOSS 1    /* AssertionError("NaN Loss Encountered") */
#define LNR_POISON_POSE_GRAD 2   /* RuntimeError("Fatal: Encountered invalid gradient in pose.") */
#define LNR_POISON_POSE 3        /* RuntimeError("Fatal: Encountered invalid pose tensor.") */

/* lnr_density_backward flags */
#define LNR_BWD_TABLE_ATOMICS 1   /* test hook: every table-gradient record goes to the 64-bit overflow accumulators (atomics) */
#define LNR_BWD_BINS 4            /* hashed levels take the binned partition (whole-line appends; same sums, measured ~3 % slower than the scan partition: DESIGN.md section 8) */
#define LNR_BWD_BINS_W8 8         /* A-B hook: the binned partition with 512-thread workgroups / 1 KB bins instead of 256 / 512 bytes */
#define LNR_BWD_DEFER_WEIGHT_FOLD 16 /* leave the per-workgroup weight-gradient slabs in the workspace: the caller adds them to grad_params
                                       with lnr_density_fold_weight_grads (same points capacity) - e.g. on another stream, beside the
                                       table-gradient reduce, instead of behind it */
#define LNR_BWD_REPORT_REGIONS 2  /* diagnostic: print to stderr how full the record regions ran (synchronises the stream) */
#define LNR_BWD_OVERWRITE_GRAD 32 /* grad_params RECEIVES this call's gradient instead of accumulating it: the table-gradient reduce writes
                                       every float of its slice (zeros included) without reading it, the weight-gradient fold stores
                                       instead of adding.  A training loop that steps after every backward then needs neither the 30 MB
                                       read here nor the optimiser's zeroing of the gradient (lnr_adam_step zero_grad = 0): 60 MB of HBM
                                       traffic per iteration.  With LNR_BWD_DEFER_WEIGHT_FOLD pass the same flag to
                                       lnr_density_fold_weight_grads. */

typedef enum LnrActivation {
    LNR_ACT_NONE = 0, LNR_ACT_RELU = 1, LNR_ACT_SINE = 2, LNR_ACT_LEAKY_RELU = 3,
    LNR_ACT_EXPONENTIAL = 4, LNR_ACT_SIGMOID = 5, LNR_ACT_SQUAREPLUS = 6,
    LNR_ACT_SOFTPLUS = 7, LNR_ACT_TANH = 8
} LnrActivation;

/* Density network = input encoding + bias-free MLP; the config schema is tinycudann's as
 * used by src/models/nerf_tcnn.py:29-38 (cfg/nerf_config/default_nerf_hash.yaml keys
 * pos_encoding_sigma / sigma_network).  Fill the first block, call lnr_net_spec_finalize. */
typedef struct LnrNetSpec {
    /* -- configuration -- */
    int32_t encoding;          /* LnrEncoding */
    int32_t n_levels;          /* HashGrid */
    int32_t n_features;        /* HashGrid: features per level, 1 | 2 | 4 | 8 */
    int32_t log2_table;        /* HashGrid: log2_hashmap_size */
    int32_t base_res;          /* HashGrid: base_resolution */
    float   per_level_scale;   /* HashGrid */
    int32_t n_frequencies;     /* Frequency */
    int32_t activation;        /* LnrActivation of the hidden layers */
    int32_t n_neurons;         /* hidden width, multiple of 16, <= 256 */
    int32_t n_hidden;          /* hidden layers, >= 1 */
    int32_t precision;         /* LnrPrecision (0 = fp32) */
    int32_t pos_rounding;      /* LnrPosRounding (0 = fma) */
    /* -- derived by lnr_net_spec_finalize -- */
    int32_t enc_dim;           /* encoding outputs */
    int32_t in_dim;            /* enc_dim rounded up to 16 (padding inputs are the constant 1) */
    int32_t n_mlp_params;      /* H*in_dim + (n_hidden-1)*H*H + 16*H */
    int64_t n_params;          /* MLP matrices first ([out][in] row-major), then encoding tables */
    float    level_scale[LNR_MAX_LEVELS];
    uint32_t level_res[LNR_MAX_LEVELS];
    uint32_t level_size[LNR_MAX_LEVELS];    /* entries */
    uint32_t level_offset[LNR_MAX_LEVELS];  /* entries, from the start of the encoding block */
    uint32_t level_hashed[LNR_MAX_LEVELS];
} LnrNetSpec;

/* Loss configuration = model_config.loss (cfg/model_config/default_model_config.yaml:42-63). */
typedef struct LnrLossConfig {
    int32_t selection;     /* 0 L1_JS, 1 L2_JS, 2 L1_LOS, 3 L2_LOS  (optimizer.py:493-534,568-574) */
    float min_js, max_js, js_alpha;
    float los_lambda;      /* already decayed by the caller if decay_los_lambda (optimizer.py:448-452) */
    float depth_lambda;
    float min_eps;         /* min_depth_eps */
    float fixed_eps;       /* LOS variants: the (decayed) depth_eps for this iteration */
} LnrLossConfig;

const char* lnr_last_error(void);          /* host; thread-local text for the last negative status */
int  lnr_version(void);

/* Optional per-kernel timing of the entry points that launch several kernels (density forward / backward): when
 * enabled, HIP events are recorded on the caller's stream around each internal launch.  lnr_profile_read waits for the
 * recorded events, returns the number of distinct kernels (names [n][name_stride] chars, summed milliseconds, calls)
 * and clears the log.  Diagnostics only; off by default. */
int lnr_profile_enable(int32_t on);
int lnr_profile_read(char* names, int32_t name_stride, float* total_ms, int32_t* calls, int32_t capacity);

/* ---- density network ------------------------------------------------------------------------- */
int lnr_net_spec_finalize(LnrNetSpec* spec /*host, in/out*/);

/* Scratch both density calls need, sized for up to n_points points per call: feature planes [enc_dim][n_points],
 * their gradient, per-level d/dx planes, per-workgroup weight-gradient slabs and the record regions of the
 * table-gradient partition.  The content between calls only matters for `reuse_features` below.
 * Limits: n_points * max(n_features_per_level, 4) < 2^30 per call, encoding table < 2^30 floats (32-bit byte offsets). */
size_t lnr_density_workspace(const LnrNetSpec* spec /*host*/, int64_t n_points);
/* The part of it lnr_density_forward alone needs (status words + feature planes): rendering / inference callers
 * (Model.forward(testing=True), analysis/compute_l1_depth.py:42-64) never pay for the backward's record regions. */
size_t lnr_density_workspace_forward(const LnrNetSpec* spec /*host*/, int64_t n_points);

/* The first LNR_WORKSPACE_STATUS_BYTES of a workspace are int32 status words the kernels write and the caller may read (with
 * the device in sync) and reset; lnr_density_workspace_init zeroes them - call it once after allocating a workspace.
 *   [LNR_STATUS_CLIPPED]  number of density outputs lnr_density_forward clipped since the last reset: non-finite values - and, with
 *                         LNR_PREC_F16, values beyond +-65504, which the reference's fp16 network returns as +-inf - are replaced
 *                         by the extremes of the network's dtype, NaN by 0, as DecoupledNeRF.forward does with nan_to_num
 *                         (nerf_tcnn.py:70-78); the caller prints the reference's "Clipping infinite outputs" warning once. */
#define LNR_WORKSPACE_STATUS_BYTES 256
#define LNR_STATUS_CLIPPED 0
#define LNR_STATUS_OVF_LEVEL0 16      /* [16 .. 16 + LNR_MAX_LEVELS): internal - the call stamp of the last lnr_density_backward that used a level's
                                          64-bit overflow accumulators (those are kept all-zero between calls instead of cleared per call).
                                          Between two density calls on a workspace its content belongs to the library: a caller that writes
                                          into it - or hands it to another network / batch size - is fine (that is detected by layout), one
                                          that scribbles over it from outside must call lnr_density_workspace_init again. */
/* REQUIRED once for every allocation that is used as a workspace - also for a new allocation that happens to get the address of a
 * released one: the library keeps a small host-side note per workspace ADDRESS (layout signature, call stamp, "overflow accumulators
 * known zero"), which lnr_density_workspace_init resets.  Without it a recycled address would be believed to hold zeroed accumulators.
 * lnr_density_workspace_release drops the note when a workspace is freed (optional but tidy: the notes are bounded - beyond 64 of them
 * every note is dropped, which only costs the next backward on each workspace one clear).  One workspace serves one stream at a time:
 * forward and backward of a call pair, and successive calls, are ordered by the stream they are issued on. */
int lnr_density_workspace_init(void* workspace, size_t workspace_bytes, void* stream);
int lnr_density_workspace_release(void* workspace);

/* sigma = MLP(enc((xyz+1)/2))[0]           replaces tinycudann forward at nerf_tcnn.py:63-72
 * Points are given either explicitly (pts != NULL, [n_points,3] in the world cube [-1,1]) or
 * implicitly as rays [n_rays,13] + z [n_rays,n_samples] (xyz = o + d*z, rendering_tcnn.py:241).
 * Two launches: level-major encoding into the workspace's feature planes, then the MLP on those planes. */
int lnr_density_forward(const LnrNetSpec* spec /*host*/, const float* params,
                        const float* pts, int64_t n_points,
                        const float* rays, const float* z, int32_t n_rays, int32_t n_samples,
                        const int32_t* n_rays_dev,
                        float* sigma /*[n_points] or [n_rays*n_samples]*/,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the above                     replaces tinycudann backward (loss.backward(), optimizer.py:366)
 * grad_params [n_params] is ACCUMULATED into (caller zeroes it; lnr_adam_step can re-zero it); NULL = parameters frozen
 * (tracking phase, optimizer.py:239-259): only the input gradient is computed, no table-gradient records, no reduce.
 * d_pts (nullable) [*,3] receives dL/dxyz per point (needed only when poses are optimised).
 * d_rays (nullable, rays form only, instead of d_pts) [n_rays,13]: dL/dxyz is reduced over the samples of each ray and
 * ADDED to the ray-record gradient (origin cols 0:3 += sum dL/dxyz, direction cols 3:6 += sum z dL/dxyz) - what
 * lnr_points_grad_to_rays does with d_pts, without materialising d_pts (64-bit fixed-point sums: reproducible).
 * reuse_features != 0: the workspace still holds the feature planes lnr_density_forward wrote for the SAME
 * spec, params and points (tinycudann keeps its forward activations the same way); 0 re-encodes first. */
int lnr_density_backward(const LnrNetSpec* spec /*host*/, const float* params,
                         const float* pts, int64_t n_points,
                         const float* rays, const float* z, int32_t n_rays, int32_t n_samples,
                         const int32_t* n_rays_dev,
                         const float* d_sigma, float* grad_params, float* d_pts, float* d_rays,
                         int32_t reuse_features, int32_t flags, void* workspace, size_t workspace_bytes,
                         void* input_grad_event /* hipEvent_t, nullable: recorded on `stream` as soon as d_pts / d_rays are complete,
                                                   before the table-gradient reduce - the pose tail and the next batch's ray build and
                                                   sampling can then run on another stream beside the rest of this call */,
                         void* stream);

/* grad_params[0 : n_mlp_params] += (flags & LNR_BWD_OVERWRITE_GRAD: =) the weight-gradient slabs a lnr_density_backward call with
 * LNR_BWD_DEFER_WEIGHT_FOLD left in `workspace` (n_points: the n_points / n_rays * n_samples of that call).  Fixed summation order:
 * reproducible. */
int lnr_density_fold_weight_grads(const LnrNetSpec* spec /*host*/, int64_t n_points, float* grad_params,
                                  void* workspace, size_t workspace_bytes, int32_t flags, void* stream);

/* ---- rays ------------------------------------------------------------------------------------- */
/* LidarRayDirections.build_lidar_rays (ray_utils.py:269-322) + get_far_val (:31-60) for one
 * keyframe: gather, rotate, normalise, clip to the cube.  Writes ALL n_index candidates plus a
 * keep flag (the `far > near + 1/scale` test, :318-322).  transform: 12 floats = rows of [R|t]. */
int lnr_build_lidar_rays(const float* directions /*[3,n_points]*/, const float* distances /*[n_points]*/,
                         int64_t n_points, const int64_t* index /*[n_index]*/, int32_t n_index,
                         const float* transform /*[12]*/, float range_min, float range_max,
                         float scale, const float* shift /*host [3]*/,
                         float* rays /*[n_index,13]*/, float* depths /*[n_index]*/,
                         uint8_t* keep /*[n_index]*/, void* stream);

/* The same for a whole keyframe window in one launch (optimizer.py:285-340): segment s covers candidates
 * [seg_start[s], seg_start[s+1]) of keyframe pose row seg_pose[s]; distances[s] == NULL means the constant
 * const_distance[s] (sky rays, sensors.py:162-167).  index (device, concatenated) may be NULL: the indices are
 * then drawn in-kernel (the torch.randint of optimizer.py:288,301) and returned in index_out.  All per-segment
 * arrays are HOST arrays of length n_seg (seg_start: n_seg+1); transforms [n_poses,12] is on the device. */
int lnr_build_window_rays(const float* const* directions, const float* const* distances, const float* const_distance,
                          const int64_t* n_points, const int32_t* seg_start, const int32_t* seg_pose, int32_t n_seg,
                          const int64_t* index, int64_t* index_out, uint64_t seed, const float* transforms,
                          float range_min, float range_max, float scale, const float* shift /*host [3]*/,
                          float* rays, float* depths, uint8_t* keep, void* stream);

/* tensor_to_transform (pose_utils.py:288-302) for n poses: pose6 [n,6] = [t, axis-angle] -> transforms [n,12]
 * (rows of [R|t]); and its backward d_transforms [n,12] -> d_pose6 [n,6] (mask nullable: 0 = fixed pose;
 * accumulate != 0 adds to d_pose6). */
int lnr_pose_forward(const float* pose6, int32_t n, float* transforms, void* stream);
int lnr_pose_backward(const float* pose6, const float* d_transforms, const uint8_t* mask, int32_t n, float* d_pose6,
                      int32_t accumulate, int32_t* poison_dev, int32_t poison_tag, void* stream);

/* Order-preserving compaction of candidate rays by `keep` (the boolean indexing at
 * ray_utils.py:322 and the vstack at optimizer.py:333-338 for a whole window).
 * seg_start [n_seg+1] (host) delimits the keyframes inside the candidate arrays;
 * out_seg_start [n_seg+1] (device) receives the compacted segment starts; n_out_dev the total. */
int lnr_compact_rays(const float* rays_in, const float* depths_in, const uint8_t* keep, const int64_t* src_index,
                     int32_t n_in, const int32_t* seg_start /*host*/, int32_t n_seg,
                     float* rays_out, float* depths_out, int64_t* src_index_out,
                     int32_t* out_seg_start, int32_t* n_out_dev, void* stream);

/* lnr_compact_rays followed, in the same launch, by what the loss needs next from the compacted batch: counts_dev [2] = the normalisers
 * lnr_count_opaque computes ({#rays, #opaque rays}, optimizer.py:460-463,488-489; far[0] = the batch's own first ray), OR record = the
 * rank's front record as lnr_shard_front_pack writes it (seg_order [n_seg] host, cap depth slots >= n_in).  Exactly one of the two. */
int lnr_compact_rays_front(const float* rays_in, const float* depths_in, const uint8_t* keep, const int64_t* src_index,
                           int32_t n_in, const int32_t* seg_start /*host*/, int32_t n_seg,
                           float* rays_out, float* depths_out, int64_t* src_index_out,
                           int32_t* out_seg_start, int32_t* n_out_dev,
                           int32_t* counts_dev /*[2] or NULL*/, const int32_t* seg_order /*[n_seg] host or NULL*/, int32_t cap,
                           float* record /*[LNR_FRONT_HEADER + cap] or NULL*/, void* stream);

/* Sharded windows (one process per GPU, keyframes round-robin): the reference's `depth > far[0]` test (optimizer.py:460-461) uses the
 * FIRST ray of the whole batch = the first kept ray of the first keyframe, in window order, that kept any.  Each rank reports its
 * candidate as one 64-bit key = (seg_order of its first segment with a kept ray) << 32 | bits of that ray's far; INT64_MAX when it kept
 * none.  A MIN all-reduce over the ranks then leaves the batch's first ray's key everywhere (low word = far[0] as float bits).
 * rays / out_seg_start: the outputs of lnr_compact_rays; seg_order [n_seg] (host): ascending position of each segment in the window. */
int lnr_first_ray_key(const float* rays, const int32_t* out_seg_start /*[n_seg+1] device*/, const int32_t* seg_order /*[n_seg] host*/,
                      int32_t n_seg, int64_t* key_out /*[1] device*/, void* stream);

/* The sharded loop's one small collective per iteration (loner_amd/mapping/sharding.py; SURVEY 8e collectives (2) + the far[0] quirk).
 * far[0] of the whole batch and the global normalisers #rays / #opaque rays (optimizer.py:460-463,488-489,569-578) - the second of
 * which depends on far[0] - come out of ONE all-gather of per-rank "front records" instead of a key exchange followed by a count
 * exchange.  A record is LNR_FRONT_HEADER + cap float32 words: [0..1] the rank's first-ray key as lnr_first_ray_key defines it (int64
 * bits), [2] its live-ray count (int32 bits), [3] 0, [4..] the ground-truth depths of its kept rays (zeros beyond the count).
 * lnr_shard_front_pack writes a rank's record from the outputs of lnr_compact_rays (n_seg = 0: a rank without keyframes - the key is
 * INT64_MAX, the count 0, every pointer but `record` may be NULL).  lnr_shard_front_reduce reads the `world` gathered records
 * (consecutive, `stride` = LNR_FRONT_HEADER + cap words each) and writes counts_dev = {sum of the live counts, number of depths d over
 * all ranks with d > 0 and not d > far[0]} - the values lnr_count_opaque computes for an unsharded batch - and far0_dev = far[0] (NaN
 * bits when no rank kept a ray), for lnr_los_loss_fused / lnr_occ_grid_step. */
#define LNR_FRONT_HEADER 4
int lnr_shard_front_pack(const float* rays, const int32_t* out_seg_start /*[n_seg+1] device*/, const int32_t* seg_order /*[n_seg] host*/,
                         int32_t n_seg, const float* depths, int32_t n_rays, const int32_t* n_rays_dev, int32_t cap,
                         float* record /*[LNR_FRONT_HEADER + cap] device*/, void* stream);
int lnr_shard_front_reduce(const float* records /*[world][stride] device*/, int32_t world, int32_t stride,
                           int32_t* counts_dev /*[2]*/, float* far0_dev /*[1]*/, void* stream);

/* ---- the sharded loop's collectives: RCCL on the caller's stream (csrc/lnr_comm.hip) -------------------------------------------
 * The reference has no multi-GPU mapping (its only fan-out is independent trials, examples/run_loner.py:339-424); SURVEY.md 8e
 * defines the keyframe-sharded window these calls serve.  One communicator per rank and process: rank 0 makes an id
 * (lnr_comm_unique_id), hands its LNR_COMM_ID_BYTES bytes to the other ranks by any means (the Python layer: the torch.distributed
 * store), every rank calls lnr_comm_init on its HIP device (collective: returns when all have).  A collective is ONE enqueue on
 * `stream` - ordered with the kernels before and after it on that stream, no host synchronisation, capturable into a hipGraph.
 * librccl.so.1 is loaded on first use (dlopen); lnr_comm_available() = 0 when it is not there. */
#define LNR_COMM_ID_BYTES 128
typedef enum LnrCommDtype { LNR_COMM_F32 = 0, LNR_COMM_BF16 = 1, LNR_COMM_I64 = 2, LNR_COMM_I32 = 3, LNR_COMM_U8 = 4 } LnrCommDtype;
typedef enum LnrCommOp { LNR_COMM_SUM = 0, LNR_COMM_MIN = 1, LNR_COMM_MAX = 2 } LnrCommOp;
int lnr_comm_available(void);
int lnr_comm_unique_id(void* id /*[LNR_COMM_ID_BYTES] host*/, size_t id_bytes);
int lnr_comm_init(const void* id, size_t id_bytes, int32_t rank, int32_t world, void** comm_out);
int lnr_comm_destroy(void* comm);
/* in place; the gradient all-reduce (optimizer.py:366's loss.backward() summed over the shards), the occupancy pseudo-gradient (I64), the failure word (MIN) */
int lnr_comm_all_reduce(void* comm, void* buf, size_t count, int32_t dtype /*LnrCommDtype*/, int32_t op /*LnrCommOp*/, void* stream);
/* recv [recv_count] = sum over ranks of send[rank * recv_count ...]: a rank's chunk of the flat gradient */
int lnr_comm_reduce_scatter(void* comm, const void* send, void* recv, size_t recv_count, int32_t dtype, void* stream);
/* recv [world * bytes_per_rank]; send may be the rank's own slot of recv (in place): front records, stepped parameter chunks */
int lnr_comm_all_gather(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
int lnr_comm_broadcast(void* comm, void* buf, size_t bytes, int32_t root, void* stream);

/* Backward of lnr_build_lidar_rays for a window: dL/drays -> dL/d[R|t] per keyframe
 * (the autograd tail ray_utils.py:281-305 <- keyframe.py:80-88).  d_transform [n_seg,12]. */
int lnr_lidar_rays_backward(const float* d_rays /*[n,13]*/, const float* rays /*[n,13]*/,
                            const int64_t* src_index /*[n]*/, const int32_t* seg_start /*device [n_seg+1]*/,
                            int32_t n_seg, const float* const* directions /*host array of n_seg device ptrs*/,
                            const int64_t* n_points /*host [n_seg]*/, const float* transforms /*[n_seg,12]*/,
                            float scale, float* d_transform /*[n_seg,12]*/, void* stream);

/* ---- samplers ---------------------------------------------------------------------------------- */
/* OccupancyGridModel.interpolate (model_tcnn.py:122-131): trilinear lookup, zero padding. */
int lnr_occ_interpolate(const float* grid /*[V,V,V] z,y,x*/, int32_t V, const float* pts /*[n,3]*/,
                        int64_t n, float* out /*[n]*/, void* stream);

/* OccGridRaySampler.get_samples (ray_sampling.py:53-92) incl. sample_pdf (rendering_tcnn.py:18-67).
 * steps: torch.linspace(0,1,n_samples/2) as a device table.  u_jitter/u_pdf [n_rays,n_samples/2]:
 * the two torch.rand draws; either may be NULL, then a counter-based generator keyed by
 * (seed, ray, sample) is used.  dbg_inds/dbg_probs/dbg_cdf (nullable) expose the searchsorted
 * indices, point_probs and cdf for stage-wise parity tests. */
int lnr_sample_rays_occ(const float* rays, int32_t n_rays, const int32_t* n_rays_dev,
                        const float* grid, int32_t V, int32_t n_samples, float perturb,
                        const float* steps, const float* u_jitter, const float* u_pdf, uint64_t seed,
                        float* z_out /*[n_rays,n_samples] sorted*/,
                        int64_t* dbg_inds, float* dbg_probs, float* dbg_cdf, void* stream);

/* UniformRaySampler.get_samples (ray_sampling.py:22-43). steps: linspace(0,1,n_samples). */
int lnr_sample_rays_uniform(const float* rays, int32_t n_rays, const int32_t* n_rays_dev,
                            int32_t n_samples, float perturb, const float* steps,
                            const float* u_jitter, uint64_t seed, float* z_out, void* stream);

/* The random numbers the kernels draw when the caller passes no random tensors, as tensors (diagnostics: distribution tests, and
 * the proof that a seeded call equals the same call with these draws handed in).  The reference draws, per forward,
 * torch.rand [n_rays, n_samples/2] twice (ray_sampling.py:72, rendering_tcnn.py:48) and torch.randn [n_rays, n_samples] once
 * (rendering_tcnn.py:104); per keyframe torch.randint (optimizer.py:288).
 *   which = LNR_DRAW_JITTER / LNR_DRAW_PDF: the uniforms in [0,1) that lnr_sample_rays_* use for element [ray][j] under `seed`;
 *   which = LNR_DRAW_NOISE: the N(0,1) values that lnr_render_* / lnr_los_loss_fused add (x noise_std) to sigma [ray][i] under `seed`;
 *   which = LNR_DRAW_RAY_INDEX + segment: the uniforms behind the ray indices lnr_build_window_rays draws for that segment
 *           (out [n_rays * n_per_ray] in draw order: index = min(floor(u * n_points), n_points - 1)). */
#define LNR_DRAW_JITTER 0
#define LNR_DRAW_PDF 1
#define LNR_DRAW_NOISE 2
#define LNR_DRAW_RAY_INDEX 16
int lnr_rng_draws(int32_t which, uint64_t seed, int32_t n_rays, int32_t n_per_ray, float* out /*[n_rays,n_per_ray]*/, void* stream);

/* ---- volume rendering ---------------------------------------------------------------------------- */
/* raw2outputs(sigma_only=True, far, ret_var=True) (rendering_tcnn.py:71-147).
 * noise [n_rays,n_samples] = randn*raw_noise_std (rendering_tcnn.py:104) or NULL with
 * noise_std>0 for the in-kernel generator (noise_std==0: no noise).  Outputs nullable. */
int lnr_render_forward(const float* sigma, const float* z, const float* rays, int32_t n_rays,
                       const int32_t* n_rays_dev, int32_t n_samples,
                       const float* noise, float noise_std, uint64_t seed,
                       float* depth, float* weights, float* opacity, float* variance, void* stream);

/* Backward of lnr_render_forward for arbitrary upstream gradients (nullable each):
 * g_depth[n], g_weights[n,S], g_opacity[n], g_variance[n]  ->  d_sigma [n,S] and the direct
 * ray-record contribution d_rays [n,13] (cols 3:6 through |dir|, col 12 = far); d_rays is
 * OVERWRITTEN.  Points' contribution to cols 0:6 is added by lnr_points_grad_to_rays. */
int lnr_render_backward(const float* sigma, const float* z, const float* rays, int32_t n_rays,
                        const int32_t* n_rays_dev, int32_t n_samples,
                        const float* noise, float noise_std, uint64_t seed,
                        const float* g_depth, const float* g_weights, const float* g_opacity,
                        const float* g_variance, float* d_sigma, float* d_rays, void* stream);

/* xyz = o + d*z  =>  d_rays[:,0:3] += sum_s d_pts ; d_rays[:,3:6] += sum_s z*d_pts
 * (rendering_tcnn.py:241 backward). */
int lnr_points_grad_to_rays(const float* d_pts /*[n,S,3]*/, const float* z, int32_t n_rays,
                            const int32_t* n_rays_dev, int32_t n_samples, float* d_rays, void* stream);

/* ---- loss ------------------------------------------------------------------------------------------ */
/* get_weights_gt (losses.py:29-51); eps_ray [n] per-ray or NULL -> eps_scalar. */
int lnr_weights_gt(const float* s /*[n,S] metres*/, const float* g /*[n] metres*/, const float* eps_ray,
                   float eps_scalar, int32_t normalise, int32_t n_rays, int32_t n_samples,
                   float* out /*[n,S]*/, void* stream);

/* get_logits_grad (losses.py:54-62), defaults eps=2, l_free=0.25, l_occ=2.5. */
int lnr_logits_grad(const float* s /*[n,S]*/, const float* g /*[n]*/, int32_t n_rays, int32_t n_samples,
                    float margin, float l_free, float l_occ, float* out, void* stream);

/* Front-to-back inference, an opt-in route of Model.render_depth (the reference composites all N_samples_test samples of every ray:
 * model_tcnn.py:73-105 -> rendering_tcnn.py:71-147).  The ray's sorted depths z [n_rays, n_samples] are drawn in full; the network is
 * evaluated in blocks of 256 samples along the ray on the rays whose transmittance is still >= 2^-24 (what lies behind contributes less
 * than fp32 resolution to the depth).  Per block b0: lnr_render_ftb_gather copies the alive rays' records and block depths into
 * compact arrays for lnr_density_forward (rays form, n_rays_dev = n_alive_dev) and zeroes next_count_dev; lnr_render_ftb_composite
 * (one wave per alive ray, the arithmetic and noise draw of lnr_render_forward) adds the block's contributions to depth_acc /
 * opacity_acc [n_rays], updates transmittance [n_rays] (caller: ones / zeros before block 0) and appends surviving rays to next_idx.
 * idx [cap] int32: alive ray indices, n_alive_dev their count; last != 0: nothing is appended.  depth = depth_acc + (1 - opacity_acc) far. */
int lnr_render_ftb_gather(const float* rays, const float* z, int32_t n_samples, const int32_t* idx, const int32_t* n_alive_dev,
                          int32_t cap, int32_t b0, int32_t block_samples, float* rays_c /*[cap,13]*/, float* z_c /*[cap,block]*/,
                          int32_t* next_count_dev, void* stream);
int lnr_render_ftb_composite(const float* sigma_c /*[cap,256]*/, const float* z, const float* rays, int32_t n_samples, const int32_t* idx,
                             const int32_t* n_alive_dev, int32_t cap, int32_t b0, int32_t block_samples,
                             const float* noise /*[n_rays,n_samples] explicit draws or NULL*/, float noise_std, uint64_t seed, float* transmittance, float* depth_acc, float* opacity_acc, int32_t* next_idx, int32_t* next_count_dev,
                             int32_t last, void* stream);

/* Fused Optimizer.compute_loss (optimizer.py:437-595, lidar branch) forward + backward:
 * render (as lnr_render_forward), weighted mean/var, JS divergence (:476-482,:614-626), dynamic
 * margin (:495-503), target weights (:504-506), depth MSE + LOS L1/L2 + opacity terms, then the
 * analytic backward to d_sigma [n,S] and the direct part of d_rays [n,13] (overwritten).
 * Reproduces the reference's `depth > far[0]` broadcast quirk (:460-461): every ground-truth depth is compared with the
 * `far` of the FIRST ray of the batch.  far0_dev (nullable, device float[1]): that value when this call sees only a shard
 * of the batch (keyframe-sharded window: rank 0's first ray, broadcast by the caller); NULL = this call's own first ray.
 * counts_dev [2] int32: {number of rays, number of opaque rays} over the WHOLE batch the loss is
 * normalised by (all GPUs) -- from lnr_count_opaque, all-reduced by the caller when sharded.
 * loss_out [8] float (accumulated; caller zeroes): {total, depth, los, opacity, sum of eps, -,-,-};
 * block_partials (nullable) [ceil(n_rays / LNR_LOSS_RAYS_PER_BLOCK) * 8] scratch: with it the terms are summed
 * without atomics (deterministic, and ~0.25 ms faster at 4096 rays than 20 k same-address atomics);
 * ray_stats (nullable) [n,8]: {depth, opacity, variance, mean_m, std_m, js, eps, opaque}.
 * weights_out (nullable) [n,S]. */
int lnr_count_opaque(const float* rays, const float* depth_gt, int32_t n_rays, const int32_t* n_rays_dev,
                     const float* far0_dev, int32_t* counts_dev /*[2], overwritten*/, void* stream);
int lnr_los_loss_fused(const float* sigma, const float* z, const float* rays, const float* depth_gt,
                       int32_t n_rays, const int32_t* n_rays_dev, int32_t n_samples,
                       const float* noise, float noise_std, uint64_t seed,
                       float scale, const LnrLossConfig* cfg /*host*/, const int32_t* counts_dev, const float* far0_dev,
                       float* loss_out, float* d_sigma, float* d_rays, float* ray_stats,
                       float* weights_out, float* block_partials, int32_t* poison_dev, int32_t poison_tag, void* stream);

/* ---- optimisers -------------------------------------------------------------------------------------- */
/* torch.optim.Adam step (optimizer.py:257-269,376-380): betas (b1,b2), eps, no weight decay,
 * `step` = 1-based step count.  If zero_grad != 0 the gradient is cleared after use
 * (zero_grad(set_to_none=True), :380).  grad_scale multiplies the gradient first (1.0 normally). */
int lnr_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t count,
                  float lr, float beta1, float beta2, float eps, int32_t step, float grad_scale,
                  int32_t zero_grad, const int32_t* poison_dev, void* stream);

/* Optimizer._step_occupancy_grid (optimizer.py:598-609): pseudo-gradient per sample
 * (losses.py:54-62) scattered trilinearly into the logit grid, SGD step grid -= lr*grad.
 * grad_acc (nullable int64 [V^3]): if given, the pseudo-gradient is accumulated there in 64-bit fixed point (2^-42;
 * caller zeroes; integer atomics: exact, order-independent, and summable over ranks with an integer all-reduce) and
 * applied by lnr_occ_grid_apply, which re-zeroes what it applied; if NULL the update is applied in place with float
 * atomics (summation order not fixed). */
int lnr_occ_grid_step(float* grid, int32_t V, const float* rays, const float* z, const float* depth_gt,
                      int32_t n_rays, const int32_t* n_rays_dev, int32_t n_samples, float scale,
                      float lr, float margin, float l_free, float l_occ, int64_t* grad_acc, void* stream);
int lnr_occ_grid_apply(float* grid, int64_t* grad_acc, int64_t count, float lr, int32_t zero_grad, const int32_t* poison_dev,
                       void* stream);

/* ---- self test ------------------------------------------------------------------------------------------ */
/* Checks the MFMA fragment layouts the density kernels rely on (v_mfma_f32_16x16x4_f32, v_mfma_f32_16x16x32_f16 and the split-bf16
 * product on v_mfma_f32_16x16x32_bf16; asymmetric operands); out[0] = max abs error of the fp32 form, out[1] = of the fp16 form,
 * out[2] = of the three-term bf16 split (operands chosen so that the kept partial products are the exact product: must be 0, and
 * the split itself must give back its input). */
int lnr_selftest_mfma(float* out /*[3]*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LONER_HIP_H */
