// Pieces shared by the fp16-mode density kernels (lnr_density_f16.hip: the reference network's class and the general backward;
// lnr_density_f16_fwd.hip: the general forward).  Lane / fragment conventions: see the head of lnr_density_f16.hip.
#pragma once
#include "lnr_density_impl.h"
#include "lnr_encoding.h"

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));


#define F16_KB_MAX 4                 // 32-wide K blocks of a layer's input: in_dim <= 128, H <= 128
#define F16_NH_MAX 3

__device__ __forceinline__ f16x8 frag_from_dwords(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return __builtin_bit_cast(f16x8, u32x4{a, b, c, d});
}
// (as ONE vector conversion: v_cvt_pk_f16_f32, round to nearest even; written as two scalar casts the pair is two v_cvt_f16_f32 and a
// v_perm_b32 whenever the values come out of an MFMA)
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2_{lo, hi}, h2));
}
__device__ __forceinline__ float round_f16(float v) { return (float)(f16)v; }
// A wave-uniform GLOBAL base address the compiler cannot prove uniform (a select between kernel arguments inside divergent code), as
// scalar registers, and a load of base + 32-bit lane offset from it.  The pointer keeps its address space: rebuilt from integers as a
// generic pointer the loads became FLAT loads, which count against vmcnt AND lgkmcnt - every wait for them was a wait for everything.
__device__ __forceinline__ uint64_t uniform_base(const void* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <typename T> __device__ __forceinline__ T ld32_at(uint64_t base, uint32_t byte_off) {
    typedef const T __attribute__((address_space(1)))* GP;
    return *(GP)(base + (uint64_t)byte_off);
}
// ReLU of two values rounded to fp16, on the packed pair: one v_pk_max_i16 on the bit patterns (negative halves, -0 included, are
// negative integers; rounding and max(., 0) commute: rounding is monotone and keeps 0) instead of one integer max per fp32 value in
// front of the conversion.  A NaN with a clear sign bit stays NaN, as with the fp32 form (fwd_act).
__device__ __forceinline__ uint32_t relu_pack_h2(float lo, float hi) {
    typedef short s16x2_ __attribute__((ext_vector_type(2)));
    const s16x2_ v = __builtin_bit_cast(s16x2_, pack_h2(lo, hi));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(v, s16x2_{0, 0}));
}

__device__ __forceinline__ int64_t live_samples(int64_t n_points, const int32_t* n_rays_dev, int n_rays, int n_samples) {
    return n_rays_dev ? (int64_t)lnr_live_rays(n_rays, n_rays_dev) * n_samples : n_points;
}

// activations; ACT >= 0 fixes the kind at compile time.  Sine (SIREN) uses the hardware sine/cosine (v_sin_f32 / v_cos_f32, ~1e-6
// absolute): its result is rounded to fp16 (5e-4) right after, and the range-reduced libm sinf was what bounded these kernels.
template <int ACT> __device__ __forceinline__ float gact(float v, int kind) {
    if (ACT == LNR_ACT_SINE) return __sinf(v);
    return ACT >= 0 ? act_fwd(v, ACT) : act_fwd(v, kind);
}
template <int ACT> __device__ __forceinline__ float gact_d(float v, int kind) {
    if (ACT == LNR_ACT_SINE) return __cosf(v);
    return ACT >= 0 ? act_bwd(v, ACT) : act_bwd(v, kind);
}

template <typename K>
static inline int f16_set_lds(K kernel, size_t lds, const char* who) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        lnr_set_error("%s: hipFuncSetAttribute(%zu) failed", who, lds);
        return LNR_ERR_LAUNCH;
    }
    return LNR_OK;
}


// the general forward (any supported width / depth / activation), lnr_density_f16_fwd.hip: features from half2 pair planes (_gen) or the
// frequency encoding of `src`'s points evaluated inside the kernel (_freq: lnr_f16_freq.h; featp / m_pad unused)
int lnr_mlp_fwd_f16_gen(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, float* sigma,
                        int64_t blocks, const PointSrc* src, hipStream_t st);
int lnr_mlp_fwd_f16_freq(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, float* sigma,
                         int64_t blocks, const PointSrc* src, hipStream_t st);
// the general backward, lnr_density_f16_bwd.hip: one launch, `blocks` workgroups = weight-gradient slabs; its LDS need (0: no kernel).
// _freq: no feature planes in, no d_feature planes out - the input gradient goes to d_pts [n][3] (nullable) directly
int lnr_mlp_bwd_f16_gen(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                        float* dfeat, float* slabs, int want_dfeat, int blocks, const PointSrc* src, float* d_pts, hipStream_t st);
int lnr_mlp_bwd_f16_freq(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                         float* dfeat, float* slabs, int want_dfeat, int blocks, const PointSrc* src, float* d_pts, hipStream_t st);
size_t lnr_f16_gen_bwd_lds(const LnrNetSpec* spec);
size_t lnr_f16_freq_bwd_lds(const LnrNetSpec* spec);
