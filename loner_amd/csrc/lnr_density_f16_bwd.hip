// General fp16-mode backward of the density MLP: host dispatch over the kernel's compile-time shape (lnr_f16_bwd_kernel.h).
// Compiled as six objects: LNR_BWD_PART 0 = ReLU and Sine kernels + the entry points, 1 / 2 = the run-time-activation kernels up to
// 64 neurons / from 128 (the slow ones to compile); LNR_BWD_FQ 0 = features from half2 pair planes, d_feature planes out; 1 = the
// frequency encoding and its input gradient evaluated inside the kernel (lnr_f16_freq.h).
#include "lnr_f16_bwd_kernel.h"

#if !defined(LNR_BWD_PART) || !defined(LNR_BWD_FQ)
#error "compile with -DLNR_BWD_PART=0|1|2 -DLNR_BWD_FQ=0|1 (loner_amd/build.py)"
#endif

#if LNR_BWD_FQ
#define LNR_BWD_ENTRY lnr_mlp_bwd_f16_freq
#define LNR_BWD_OTHER lnr_mlp_bwd_f16_freq_other
#define LNR_BWD_OTHER_WIDE lnr_mlp_bwd_f16_freq_other_wide
#define LNR_BWD_LDS lnr_f16_freq_bwd_lds
#define LNR_BWD_KT_LO 2
#define LNR_BWD_KT_HI 3
static int f16_gen_kt(const LnrNetSpec* spec) { return lnr_freq_kt(spec->n_frequencies); }
#else
#define LNR_BWD_ENTRY lnr_mlp_bwd_f16_gen
#define LNR_BWD_OTHER lnr_mlp_bwd_f16_gen_other
#define LNR_BWD_OTHER_WIDE lnr_mlp_bwd_f16_gen_other_wide
#define LNR_BWD_LDS lnr_f16_gen_bwd_lds
#define LNR_BWD_KT_LO 2
#define LNR_BWD_KT_HI 4
static int f16_gen_kt(const LnrNetSpec* spec) { return (spec->in_dim + 31) / 32 <= 2 ? 2 : 4; }   // first-layer K blocks (as the forward)
#endif

int LNR_BWD_OTHER(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                  float* dfeat, float* slabs, int want_dfeat, int blocks, const PointSrc* src, float* d_pts, hipStream_t st);
int LNR_BWD_OTHER_WIDE(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                       float* dfeat, float* slabs, int want_dfeat, int blocks, const PointSrc* src, float* d_pts, hipStream_t st);

#define LNR_F16_GEN_BWD_K(HT, ACT, NH, KT)                                                                                        \
    do {                                                                                                                         \
        const size_t lds = BwdLds<HT, NH, KT>::BYTES;                                                                            \
        int rc_ = f16_set_lds(mlp_backward_f16_gen_kernel<HT, ACT, NH, KT, LNR_BWD_FQ != 0>, lds, "lnr_density_backward");       \
        if (rc_) return rc_;                                                                                                     \
        hipLaunchKernelGGL((mlp_backward_f16_gen_kernel<HT, ACT, NH, KT, LNR_BWD_FQ != 0>), grid, block, lds, st, *spec, params, featp, m_pad, pt->n_points, \
                           pt->n_rays_dev, pt->n_rays, pt->n_samples, d_sigma, dfeat, slabs, want_dfeat, *src, d_pts);           \
    } while (0)
#define LNR_F16_GEN_BWD(HT, ACT, NH) do { if (kt == LNR_BWD_KT_LO) LNR_F16_GEN_BWD_K(HT, ACT, NH, LNR_BWD_KT_LO); else LNR_F16_GEN_BWD_K(HT, ACT, NH, LNR_BWD_KT_HI); } while (0)
#define LNR_F16_GEN_BWD_N(HT, ACT) do { if (spec->n_hidden == 1) LNR_F16_GEN_BWD(HT, ACT, 1); else if (spec->n_hidden == 2) LNR_F16_GEN_BWD(HT, ACT, 2); else LNR_F16_GEN_BWD(HT, ACT, 3); } while (0)

#if LNR_BWD_PART == 0
// LDS bytes of the general backward for this network, 0 if the shape has no kernel
size_t LNR_BWD_LDS(const LnrNetSpec* spec) {
    const int kt = f16_gen_kt(spec);
    if (kt == 0) return 0;
#define LNR_LDS_K(HT, NH) (kt == LNR_BWD_KT_LO ? BwdLds<HT, NH, LNR_BWD_KT_LO>::BYTES : BwdLds<HT, NH, LNR_BWD_KT_HI>::BYTES)
#define LNR_LDS_N(HT) (spec->n_hidden == 1 ? LNR_LDS_K(HT, 1) : spec->n_hidden == 2 ? LNR_LDS_K(HT, 2) : spec->n_hidden == 3 ? LNR_LDS_K(HT, 3) : 0)
    switch (spec->n_neurons) {
        case 16: return LNR_LDS_N(1);
        case 32: return LNR_LDS_N(2);
        case 64: return LNR_LDS_N(4);
        case 128: return LNR_LDS_N(8);
        case 256: return spec->n_hidden == 1 ? LNR_LDS_K(16, 1) : 0;
        default: return 0;
    }
#undef LNR_LDS_N
#undef LNR_LDS_K
}

// planes form: src / d_pts unused; fused form (LNR_BWD_FQ): featp / m_pad / dfeat unused, want_dfeat = (d_pts != nullptr)
int LNR_BWD_ENTRY(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                  float* dfeat, float* slabs, int want_dfeat, int blocks, const PointSrc* src, float* d_pts, hipStream_t st) {
#if LNR_BWD_FQ
    want_dfeat = d_pts != nullptr ? 1 : 0;
#else
    // a K block's 16 feature planes are one buffer descriptor (32-bit record count, 32-bit lane offsets); the feature gradient is
    // stored with 32-bit byte offsets over all its planes
    if (m_pad * 4 * 16 > (int64_t)0x7FFFFFFF || (want_dfeat && (int64_t)spec->enc_dim * m_pad * 4 > (int64_t)0xFFFFFFFFll)) {
        lnr_set_error("lnr_density_backward: too many points per call for the general fp16 kernels (a plane of %lld samples: 64 x plane bytes and enc_dim x plane bytes must fit 32 bits)", (long long)m_pad);
        return LNR_ERR_UNSUPPORTED;
    }
#endif
    const int akind = spec->activation;
    if (akind != LNR_ACT_RELU && akind != LNR_ACT_SINE) return LNR_BWD_OTHER(spec, params, featp, m_pad, pt, d_sigma, dfeat, slabs, want_dfeat, blocks, src, d_pts, st);
    const int kt = f16_gen_kt(spec);
    const dim3 grid((unsigned)blocks), block(LNR_DENSITY_BLOCK);          // one workgroup per CU (LDS), persistent over the steps
#define LNR_F16_GEN_BWD_A(HT) do { if (akind == LNR_ACT_RELU) LNR_F16_GEN_BWD_N(HT, LNR_ACT_RELU); else LNR_F16_GEN_BWD_N(HT, LNR_ACT_SINE); } while (0)
#define LNR_F16_GEN_BWD_W(ACT) LNR_F16_GEN_BWD(16, ACT, 1)           /* 256 neurons: one hidden layer (lnr_f16_supported) */
    switch (spec->n_neurons / 16) {
        case 1: LNR_F16_GEN_BWD_A(1); break;
        case 2: LNR_F16_GEN_BWD_A(2); break;
        case 4: LNR_F16_GEN_BWD_A(4); break;
        case 8: LNR_F16_GEN_BWD_A(8); break;
        default: if (akind == LNR_ACT_RELU) LNR_F16_GEN_BWD_W(LNR_ACT_RELU); else LNR_F16_GEN_BWD_W(LNR_ACT_SINE); break;
    }
#undef LNR_F16_GEN_BWD_W
#undef LNR_F16_GEN_BWD_A
#ifdef LNR_PHASE_TIMING
    if (getenv("LNR_PHASE_TIMING")) {
        static const char* names[LNR_N_PHASES] = {"fill + first load", "step head (unit)", "forward layer 1", "forward layers 2..", "write images (hidden)",
                                                  "dA = W^T dZ (hidden)", "barrier a", "dW hidden", "barrier b", "next step's features + d_sigma",
                                                  "first layer dX + chain rule", "write x image", "barrier c", "dW first", "barrier d", "slab write"};
        unsigned long long h[LNR_N_PHASES];
        if (lnr_phase_fetch(HIP_SYMBOL(lnr_f16_bwd_phase_cycles), h, LNR_N_PHASES, st)) lnr_phase_print(LNR_BWD_FQ ? "mlp_backward_f16 fused" : "mlp_backward_f16 planes", names, h);
    }
#endif
    return LNR_OK;
}
#elif LNR_BWD_PART == 1
int LNR_BWD_OTHER(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                  float* dfeat, float* slabs, int want_dfeat, int blocks, const PointSrc* src, float* d_pts, hipStream_t st) {
    const int kt = f16_gen_kt(spec);
    const dim3 grid((unsigned)blocks), block(LNR_DENSITY_BLOCK);          // one workgroup per CU (LDS), persistent over the steps
    switch (spec->n_neurons / 16) {
        case 1: LNR_F16_GEN_BWD_N(1, -1); break;
        case 2: LNR_F16_GEN_BWD_N(2, -1); break;
        case 4: LNR_F16_GEN_BWD_N(4, -1); break;
        default: return LNR_BWD_OTHER_WIDE(spec, params, featp, m_pad, pt, d_sigma, dfeat, slabs, want_dfeat, blocks, src, d_pts, st);
    }
    return LNR_OK;
}
#else
int LNR_BWD_OTHER_WIDE(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                       float* dfeat, float* slabs, int want_dfeat, int blocks, const PointSrc* src, float* d_pts, hipStream_t st) {
    const int kt = f16_gen_kt(spec);
    const dim3 grid((unsigned)blocks), block(LNR_DENSITY_BLOCK);          // one workgroup per CU (LDS), persistent over the steps
    switch (spec->n_neurons / 16) {
        case 8: LNR_F16_GEN_BWD_N(8, -1); break;
        default: LNR_F16_GEN_BWD(16, -1, 1); break;           /* 256 neurons: one hidden layer (lnr_f16_supported) */
    }
    return LNR_OK;
}
#endif
