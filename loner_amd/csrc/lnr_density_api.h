// Internal interface between the density host dispatcher (lnr_density.hip) and the per-width kernel
// translation units (lnr_density_ht.hip compiled with -DLNR_HT=1|2|4|8|16).
#pragma once
#include "lnr_common.h"

#define LNR_DENSITY_BLOCK 256
#define LNR_DENSITY_MAX_BLOCKS 1024
#define LNR_LDS_LIMIT (160 * 1024)

struct PointSrc {
    const float* pts;
    const float* rays;
    const float* z;
    int32_t n_samples;
    int64_t n_points;   // used when pts != null
    int32_t n_rays;
    const int32_t* n_rays_dev;
};

#define LNR_BWD_MAX_BLOCKS 512
#define LNR_SLICE_SHIFT 13            // 8192 floats per owner (64 KB of LDS: 64-bit fixed-point accumulators)
#define LNR_REGION_BUDGET (24ull << 30)
#define LNR_COMBINE_SCALE_MAX 3000.0f
#define LNR_COMBINE_FILL 1.0        // expected records of a run-length combined level, as a fraction of its uncombined count
#ifndef LNR_REGION_HEADROOM
#define LNR_REGION_HEADROOM 2.0     // region capacity = expectation x this + LNR_REGION_SLACK records
#endif
#ifndef LNR_REGION_SLACK
#define LNR_REGION_SLACK 64.0
#endif
#ifndef LNR_REGION_HEADROOM_XP
#define LNR_REGION_HEADROOM_XP LNR_REGION_HEADROOM     // the same for the x-pair levels (A/B knob)
#endif
#ifndef LNR_ENC_BWD_BLOCK
#define LNR_ENC_BWD_BLOCK 512       // threads of an encode-backward workgroup = samples of one partition batch
#endif
#ifndef LNR_BATCHES_PER_WG
#define LNR_BATCHES_PER_WG 4        // partition batches per encode-backward workgroup: a region collects the records of this many batches
#endif
#define LNR_REDUCE_SPLIT 32         // reduce workgroups per owner of a dense-indexed record level
#define LNR_FIX_SCALE 4398046511104.0f /* 2^42: LDS gradient accumulators are 64-bit fixed point */
#define LNR_BIN_BYTES 1024            /* LDS bin of one owner in the binned partition (lnr_encode.hip) */
#define LNR_BIN_MAX_OWNERS 64         /* 64 bins = 64 KB of LDS: two workgroups per CU */
#ifndef LNR_XPAIR_SCALE_MIN
#define LNR_XPAIR_SCALE_MIN 3000.0f   /* hashed power-of-two levels at least this fine take x-pair records (below: run-length combined 8-byte records) */
#endif

#ifndef LNR_SPLIT_DX
#define LNR_SPLIT_DX 0      /* 1: hash grids take the input gradient as a kernel of its own (encode_dx_pair_kernel); measured SLOWER */
#endif
#define LNR_ENC_PART_DX 1        /* lnr_encode_backward: the input gradient (d_pts / d_rays) */
#define LNR_ENC_PART_RECORDS 2   /* lnr_encode_backward: the table-gradient records */

// rn(v * 2^42) as a 64-bit integer.  There is no f32 -> i64 convert instruction (the compiler's expansion is ~20 VALU
// instructions, and these kernels are VALU-issue bound): for |v| < 2^8 the sum (double)v * 2^42 + 1.5 * 2^52 is exact up to
// its one rounding to an integer (nearest even, like __float2ll_rn) and leaves that integer in the mantissa - a convert, an
// fp64 fma (full rate on CDNA4) and a 32-bit subtract.  Same result as the generic path wherever both are defined.
__device__ __forceinline__ long long lnr_to_fix_small(float v) {
    const double d = __builtin_fma((double)v, 4398046511104.0, 6755399441055744.0);
    return __double_as_longlong(d) - 0x4338000000000000ll;
}
__device__ __forceinline__ long long lnr_to_fix(float v) {
    if (__builtin_expect(__builtin_fabsf(v) < 256.0f, 1)) return lnr_to_fix_small(v);
    return __float2ll_rn(v * LNR_FIX_SCALE);
}

// Table-gradient records are 8 bytes.  n_features == 1: {float index, fp32 value}.  n_features >= 2: two values of one
// entry pair, each rounded (to nearest even) to 26 bits = sign, 8 exponent, 17 mantissa bits (relative error <= 2^-18,
// below the reordering noise of an fp32 sum of the same records), and the pair's index inside its owner slice:
//   hi = [v0: 31..6][v1 high 6: 5..0]     lo = [v1 low 20: 31..12][pair index: 11..0]
static_assert(LNR_SLICE_SHIFT == 13, "pair records hold a 12-bit pair index: 8192-float owner slices");
__host__ __device__ __forceinline__ uint32_t lnr_pack26(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    u += 0x1Fu + ((u >> 6) & 1u);
    return u >> 6;
}
__device__ __forceinline__ uint2 lnr_pack_pair(uint32_t pair_idx, float v0, float v1) {
    const uint32_t a = lnr_pack26(v0), b = lnr_pack26(v1);
    return make_uint2((b << 12) | (pair_idx & 0xFFFu), (a << 6) | (b >> 20));     // .x = lo, .y = hi
}
__device__ __forceinline__ void lnr_unpack_pair(uint2 r, uint32_t& pair_idx, float& v0, float& v1) {
    pair_idx = r.x & 0xFFFu;
    v0 = __uint_as_float(r.y & 0xFFFFFFC0u);
    v1 = __uint_as_float(((r.y & 0x3Fu) << 26) | ((r.x >> 6) & 0x03FFFFC0u));
}

// x-pair records (12 bytes): on hashed power-of-two levels the table index is x ^ (y*P1 ^ z*P2), so the corners (x, y, z) and (x+1, y, z)
// of a cell differ only in the low bits of the index - e1 = e0 ^ (2^(t+1) - 1), t = trailing one bits of x - and fall into the same
// owner slice unless t >= 12.  Their four updates (two corners x two features) are a rank-1 product (1-fx, fx) x (a0, a1) with
// a = wy*wz*g, so ONE record carries both corners: half the records to rank, stage and copy on the fine levels (where nothing is
// run-length combined) and 12 instead of 16 bytes per corner pair.  Used when n_features == 2 (lnr_level_uses_xpairs).
//   a = [a0 bits 25..10 :16][t :4][pair index :12]   b = [a0 bits 9..0 :10][a1 bits 25..4 :22]   c = [a1 bits 3..0 :4][fx :28]
// a0, a1 rounded to 26 bits like the values of a pair record, fx (in [0,1)) to 28 bits; the reduce - and the overflow path, with
// the same rounding - forms (1-fx)*a and fx*a in fp32.
struct LnrXRec { uint32_t a, b, c; };
__device__ __forceinline__ LnrXRec lnr_pack_xpair(uint32_t pair_idx, uint32_t t, float a0, float a1, float fx) {
    const uint32_t p0 = lnr_pack26(a0), p1 = lnr_pack26(a1);
    uint32_t fb;
    memcpy(&fb, &fx, 4);
    fb = (fb + 8u) >> 4;
    LnrXRec r;
    r.a = (pair_idx & 0xFFFu) | ((t & 0xFu) << 12) | ((p0 >> 10) << 16);
    r.b = ((p0 & 0x3FFu) << 22) | (p1 >> 4);
    r.c = ((p1 & 0xFu) << 28) | (fb & 0x0FFFFFFFu);
    return r;
}
__device__ __forceinline__ void lnr_unpack_xpair(const LnrXRec& r, uint32_t& pair_idx, uint32_t& t, float& a0, float& a1, float& fx) {
    pair_idx = r.a & 0xFFFu;
    t = (r.a >> 12) & 0xFu;
    a0 = __uint_as_float((((r.a >> 16) << 10) | (r.b >> 22)) << 6);
    a1 = __uint_as_float((((r.b & 0x3FFFFFu) << 4) | (r.c >> 28)) << 6);
    fx = __uint_as_float((r.c & 0x0FFFFFFFu) << 4);
}
// the four fixed-point contributions of an (unpacked) x-pair record: (1-fx)*a0, (1-fx)*a1 to the x corner, fx*a0, fx*a1 to the x+1 corner
// (products rounded to fp32 first; one definition for the reduce and the overflow path, which must agree to the bit)
__device__ __forceinline__ void lnr_xpair_fix(float a0, float a1, float fx, long long q[4]) {
    const float gx = 1.0f - fx;
    if (__builtin_expect(__builtin_fmaxf(__builtin_fabsf(a0), __builtin_fabsf(a1)) < 256.0f, 1)) {
        q[0] = lnr_to_fix_small(gx * a0); q[1] = lnr_to_fix_small(gx * a1); q[2] = lnr_to_fix_small(fx * a0); q[3] = lnr_to_fix_small(fx * a1);
    } else {
        q[0] = __float2ll_rn(gx * a0 * LNR_FIX_SCALE); q[1] = __float2ll_rn(gx * a1 * LNR_FIX_SCALE);
        q[2] = __float2ll_rn(fx * a0 * LNR_FIX_SCALE); q[3] = __float2ll_rn(fx * a1 * LNR_FIX_SCALE);
    }
}
// which levels can take x-pair records; the host decides once per launch (RegionPlan::xp) and both kernels read the plan
static inline bool lnr_level_can_use_xpairs(const LnrNetSpec& s, int l, float scale_min) {
    const uint32_t size = s.level_size[l];
    return s.n_features == 2 && s.level_hashed[l] != 0 && (size & (size - 1u)) == 0u && s.level_scale[l] >= scale_min &&
           ((s.level_offset[l] * 2u) & ((1u << LNR_SLICE_SHIFT) - 1u)) == 0u;      // owner slices aligned with the level: low index bits stay inside a slice
}

// launch plan chosen by the dispatcher
// padded LDS copies of the weight matrices (lnr_density_impl.h: lnr_fill_w_lds): row stride and total size in floats
__host__ __device__ inline int lnr_w_stride(int cols) { return cols + 4; }
__host__ __device__ inline int lnr_w_lds_floats(int H, int in_dim, int n_hidden, bool with_first) {
    return (with_first ? H * lnr_w_stride(in_dim) : 0) + (n_hidden - 1) * H * lnr_w_stride(H) + 16 * H;
}

struct DensityPlan {
    int grid;        // workgroups
    int waves;       // waves per workgroup (1, 2 or 4)
    int w_lds;       // 1: MLP matrices staged in LDS, 0: read from global memory (regs kernel also 2: all but the first layer's in LDS)
    size_t lds;      // dynamic LDS bytes
    int fast32;      // 1: the register-resident kernels for 32 features -> <= 64 ReLU neurons -> 1
    int dw64;        // general backward: weight gradients accumulate in LDS in 64-bit fixed point (1) or fp32 (0)
    int regs;        // backward: 1 = mlp_backward_regs_kernel (weight gradient in registers, owned by rows; lnr_density_regs.h)
    int n_slabs;     // weight-gradient slabs the backward writes
};

// point count description for the MLP kernels (features come from planes)
struct MlpPoints {
    int64_t n_points;            // explicit point count (pts mode)
    const int32_t* n_rays_dev;   // non-null: live ray count on the device (rays mode)
    int32_t n_rays, n_samples;
    int32_t* clip_flag;          // workspace status word: number of sigma outputs clipped by the forward kernels (nullable)
};

#define LNR_DECLARE_HT(HT)                                                                                              \
    int lnr_mlp_fwd_ht##HT(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, \
                           float* sigma, const DensityPlan* plan, hipStream_t st);                                      \
    int lnr_mlp_bwd_ht##HT(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, \
                           const float* d_sigma, float* dfeat, float* slabs, int want_dfeat, const DensityPlan* plan,   \
                           hipStream_t st);
#define LNR_DECLARE_REGS(HT, NH)                                                                                        \
    int lnr_mlp_bwd_regs_ht##HT##_nh##NH(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, \
                                         const MlpPoints* pt, const float* d_sigma, float* dfeat, float* slabs,         \
                                         int want_dfeat, const DensityPlan* plan, hipStream_t st);
LNR_DECLARE_REGS(4, 1) LNR_DECLARE_REGS(4, 2) LNR_DECLARE_REGS(4, 3)
LNR_DECLARE_REGS(8, 1) LNR_DECLARE_REGS(8, 2) LNR_DECLARE_REGS(8, 3)
LNR_DECLARE_REGS(16, 1)
LNR_DECLARE_HT(1)
LNR_DECLARE_HT(2)
LNR_DECLARE_HT(4)
LNR_DECLARE_HT(8)
LNR_DECLARE_HT(16)

// the record levels handled by one encode-backward launch
struct LevelList {
    int n;                          // levels handled by a launch
    int lv[LNR_MAX_LEVELS];
    int slab_off[LNR_MAX_LEVELS];   // offset of the level's 64-bit overflow accumulators (see below)
};
// Every record level has 64-bit fixed-point overflow accumulators in the workspace (one per table float of the level): a
// record that does not fit its staging bin or its region is added there (integer atomics: exact, order-independent) with
// the same 26-bit rounding as a packed record, and the reduce folds the accumulators into the same fixed-point sum - so the
// table gradient does not depend on which records happened to overflow.  Levels with dense (non-hashed) indexing overflow
// routinely (they are spatially coherent: an owner slice is a slab of cells and a ray pours into a few of them), hashed
// levels only by statistical accident.

// Record regions, sized per level: [owner of the level][chunk (= encode-backward workgroup)] x bytes[l], back to back from off[l].
// A region holds twice the level's expected records per (owner, chunk) plus 64 (LNR_REGION_HEADROOM / _SLACK; binned levels: plus
// one bin and a line, the room a region must have left to stay open), in that level's record format (8-byte pair records, 12-byte
// x-pair records), in 256-byte units.  The reduce reads only the records a region holds (its count), so capacity costs memory -
// several GB at the bench shape, lnr_density_workspace reports it - not bandwidth.
struct RegionPlan {
    uint64_t off[LNR_MAX_LEVELS];       // byte offset of the level's regions
    uint32_t bytes[LNR_MAX_LEVELS];     // bytes per region (multiple of 16); 0: all records overflow (LNR_BWD_TABLE_ATOMICS)
    uint8_t xp[LNR_MAX_LEVELS];         // 1: the level's records are 12-byte x-pair records, 0: 8-byte records
    uint8_t split[LNR_MAX_LEVELS];      // > 1: the level's owners are reduced by this many workgroups each (table_grad_reduce_split_kernel)
    uint8_t binned[LNR_MAX_LEVELS];     // 1: partitioned by encode_backward_binned_kernel (fixed LDS bin per owner, whole-line appends)
};

// level-major encoding (lnr_encode.hip)
int lnr_encode_forward(const LnrNetSpec* spec, const float* params, const PointSrc* src, int64_t cap_points, float* feat,
                       int64_t m_pad, bool half_planes, hipStream_t st);

// fp16-storage MLP kernels (lnr_density_f16.hip): features arrive as half2 planes [level][m_pad]
bool lnr_f16_supported(const LnrNetSpec* spec);
// fp16 mode + frequency encoding (up to 16 frequencies): the encoding is evaluated inside the MLP kernels (lnr_f16_freq.h) - no
// encode launches, no feature / d_feature planes; the backward writes d_pts itself
bool lnr_f16_fused_freq(const LnrNetSpec* spec);
int lnr_mlp_fwd_f16(const LnrNetSpec* spec, const float* params, const void* featp, int64_t m_pad, const MlpPoints* pt, float* sigma,
                    const PointSrc* src, hipStream_t st);
int lnr_mlp_bwd_f16(const LnrNetSpec* spec, const float* params, const void* featp, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                    float* dfeat, float* slabs, int want_dfeat, int* n_slabs, const PointSrc* src, float* d_pts, hipStream_t st);
int lnr_f16_bwd_slabs(const LnrNetSpec* spec, int64_t n_points);
int lnr_selftest_mfma_f16(float* out, hipStream_t st);
// fp32 mode, default shape class, on the bf16 matrix pipe with three-term operand splits (lnr_density_bf3.hip)
bool lnr_bf3_class(const LnrNetSpec* spec, int64_t n_points);
int lnr_mlp_fwd_bf3(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, float* sigma, hipStream_t st);
int lnr_mlp_bwd_bf3(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                    float* dfeat, float* slabs, int want_dfeat, int* n_slabs, hipStream_t st);
int lnr_bf3_bwd_slabs(const LnrNetSpec* spec, int64_t n_points);
int lnr_selftest_mfma_bf3(float* out, hipStream_t st);
// 256 neurons x 2..3 hidden layers, both precisions: layer by layer through chunk planes in the workspace (lnr_density_wide.hip)
bool lnr_wide_class(const LnrNetSpec* spec);
size_t lnr_wide_workspace(const LnrNetSpec* spec, int64_t n_points);
int lnr_wide_slabs(void);
int lnr_mlp_fwd_wide(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, float* sigma,
                     void* planes, hipStream_t st);
int lnr_mlp_bwd_wide(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                     float* dfeat, float* slabs, int want_dfeat, int want_dw, int* n_slabs, void* planes, hipStream_t st);
int lnr_encode_backward(const LnrNetSpec* spec, const float* params, const PointSrc* src, int64_t cap_points, const float* dfeat,
                        float* dxl, int64_t m_pad, float* grad_table, void* regions, const RegionPlan* plan, int* counts, int bpg,
                        int maxo, int shift, long long* ovf, int* ovf_flag, int epoch, float* d_pts, float* d_rays_acc, long long* ray_acc, bool bins_w8, int parts,
                        hipStream_t st);
