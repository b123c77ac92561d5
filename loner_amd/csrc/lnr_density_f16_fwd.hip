// General fp16-mode forward of the density MLP: host dispatch over the kernel's compile-time shape (lnr_f16_fwd_kernel.h).
// Compiled as four objects: LNR_FWD_PART 0 = ReLU and Sine kernels + the entry point, 1 = the run-time-activation kernels (whose
// every activation carries the switch over the libm evaluations: they are the slow ones to compile); LNR_FWD_FQ 0 = features from
// half2 pair planes (hash grids), 1 = the frequency encoding evaluated inside the kernel (lnr_f16_freq.h).
#include "lnr_f16_fwd_kernel.h"

#if !defined(LNR_FWD_PART) || !defined(LNR_FWD_FQ)
#error "compile with -DLNR_FWD_PART=0|1 -DLNR_FWD_FQ=0|1 (loner_amd/build.py)"
#endif

#define LNR_F16_FWD_CT 2             // 16-sample column tiles per wave step

#ifdef LNR_DEV_PROBES
#define LNR_F16_FWD_PROBE_ONE_PER_CU (getenv("LNR_F16_FWD_ONE_PER_CU") != nullptr)
#else
#define LNR_F16_FWD_PROBE_ONE_PER_CU false
#endif

#if LNR_FWD_FQ
#define LNR_FWD_ENTRY lnr_mlp_fwd_f16_freq
#define LNR_FWD_OTHER lnr_mlp_fwd_f16_freq_other
#define LNR_FWD_KT_LO 2
#define LNR_FWD_KT_HI 3
#else
#define LNR_FWD_ENTRY lnr_mlp_fwd_f16_gen
#define LNR_FWD_OTHER lnr_mlp_fwd_f16_gen_other
#define LNR_FWD_KT_LO 2
#define LNR_FWD_KT_HI 4
#endif

int LNR_FWD_OTHER(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, float* sigma,
                  int64_t blocks, const PointSrc* src, int kt, hipStream_t st);

#define LNR_F16_GEN_FWD_K(HT, ACT, NH, KT)                                                                                        \
    do {                                                                                                                         \
        const size_t lds = FwdLds<HT, NH, KT>::BYTES;                                                                            \
        /* persistent: every workgroup converts the weights into its LDS once, so no more workgroups than the chip holds */      \
        int64_t resident = 256 * (int64_t)((size_t)LNR_LDS_LIMIT / lds >= 2 ? 2 : 1);                                            \
        if (LNR_F16_FWD_PROBE_ONE_PER_CU) resident = 256;  /* development probe (-DLNR_DEV_PROBES): one wave per SIMD */          \
        const dim3 grid((unsigned)(blocks < resident ? blocks : resident));                                                      \
        int rc_ = f16_set_lds(mlp_forward_f16_gen_kernel<HT, ACT, NH, KT, LNR_F16_FWD_CT, LNR_FWD_FQ != 0>, lds, "lnr_density_forward"); \
        if (rc_) return rc_;                                                                                                     \
        hipLaunchKernelGGL((mlp_forward_f16_gen_kernel<HT, ACT, NH, KT, LNR_F16_FWD_CT, LNR_FWD_FQ != 0>), grid, block, lds, st, *spec, params, featp, m_pad, \
                           pt->n_points, pt->n_rays_dev, pt->n_rays, pt->n_samples, sigma, pt->clip_flag, *src);                 \
    } while (0)
#define LNR_F16_GEN_FWD(HT, ACT, NH) do { if (kt == LNR_FWD_KT_LO) LNR_F16_GEN_FWD_K(HT, ACT, NH, LNR_FWD_KT_LO); else LNR_F16_GEN_FWD_K(HT, ACT, NH, LNR_FWD_KT_HI); } while (0)
#define LNR_F16_GEN_FWD_N(HT, ACT) do { if (spec->n_hidden == 1) LNR_F16_GEN_FWD(HT, ACT, 1); else if (spec->n_hidden == 2) LNR_F16_GEN_FWD(HT, ACT, 2); else LNR_F16_GEN_FWD(HT, ACT, 3); } while (0)

#if LNR_FWD_PART == 0
// planes form: src is the point source the planes were encoded from (unused by the kernel); fused form: featp / m_pad unused
int LNR_FWD_ENTRY(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, float* sigma,
                  int64_t blocks, const PointSrc* src, hipStream_t st) {
#if LNR_FWD_FQ
    const int kt = lnr_freq_kt(spec->n_frequencies);
    if (kt == 0) { lnr_set_error("lnr_density_forward: no fused frequency kernel for n_frequencies = %d", spec->n_frequencies); return LNR_ERR_UNSUPPORTED; }
#else
    // a K block's 16 feature planes are one buffer descriptor (32-bit record count, 32-bit lane offsets)
    if (m_pad * 4 * 16 > (int64_t)0x7FFFFFFF) {
        lnr_set_error("lnr_density_forward: at most 2^25 points per call for the general fp16 kernels (got a plane of %lld samples)", (long long)m_pad);
        return LNR_ERR_UNSUPPORTED;
    }
    const int kt = (spec->in_dim + 31) / 32 <= 2 ? 2 : 4;              // first-layer K blocks: at most one block of zero padding
#endif
    const int akind = spec->activation;
    if (akind != LNR_ACT_RELU && akind != LNR_ACT_SINE) return LNR_FWD_OTHER(spec, params, featp, m_pad, pt, sigma, blocks, src, kt, st);
    const dim3 block(LNR_DENSITY_BLOCK);
#define LNR_F16_GEN_FWD_A(HT) do { if (akind == LNR_ACT_RELU) LNR_F16_GEN_FWD_N(HT, LNR_ACT_RELU); else LNR_F16_GEN_FWD_N(HT, LNR_ACT_SINE); } while (0)
    switch (spec->n_neurons / 16) {
        case 1: LNR_F16_GEN_FWD_A(1); break;
        case 2: LNR_F16_GEN_FWD_A(2); break;
        case 4: LNR_F16_GEN_FWD_A(4); break;
        case 8: LNR_F16_GEN_FWD_A(8); break;
        default: if (akind == LNR_ACT_RELU) LNR_F16_GEN_FWD(16, LNR_ACT_RELU, 1); else LNR_F16_GEN_FWD(16, LNR_ACT_SINE, 1); break;   /* 256 neurons: one hidden layer */
    }
#undef LNR_F16_GEN_FWD_A
#ifdef LNR_PHASE_TIMING
    if (getenv("LNR_PHASE_TIMING")) {
        static const char* names[LNR_N_PHASES] = {"fill + first features", "next points (issue)", "layer 1 (+ side slots)", "layer 2 (+ side slots)", "output store", "left-over slots"};
        unsigned long long h[LNR_N_PHASES];
        if (lnr_phase_fetch(HIP_SYMBOL(lnr_f16_fwd_phase_cycles), h, LNR_N_PHASES, st)) lnr_phase_print(LNR_FWD_FQ ? "mlp_forward_f16 fused" : "mlp_forward_f16 planes", names, h);
    }
#endif
    return LNR_OK;
}
#else
int LNR_FWD_OTHER(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, float* sigma,
                  int64_t blocks, const PointSrc* src, int kt, hipStream_t st) {
    const dim3 block(LNR_DENSITY_BLOCK);
    switch (spec->n_neurons / 16) {
        case 1: LNR_F16_GEN_FWD_N(1, -1); break;
        case 2: LNR_F16_GEN_FWD_N(2, -1); break;
        case 4: LNR_F16_GEN_FWD_N(4, -1); break;
        case 8: LNR_F16_GEN_FWD_N(8, -1); break;
        default: LNR_F16_GEN_FWD(16, -1, 1); break;                     /* 256 neurons: one hidden layer */
    }
    return LNR_OK;
}
#endif
