// General fp16-mode forward of the density MLP: host dispatch over the kernel's compile-time shape (lnr_f16_fwd_kernel.h).
#include "lnr_f16_fwd_kernel.h"

#define LNR_F16_FWD_CT 2             // 16-sample column tiles per wave step

int lnr_mlp_fwd_f16_gen(const LnrNetSpec* spec, const float* params, const uint32_t* featp, int64_t m_pad, const MlpPoints* pt, float* sigma,
                        int64_t blocks, hipStream_t st) {
    const int akind = spec->activation;
    const int kt = (spec->in_dim + 31) / 32 <= 2 ? 2 : 4;              // first-layer K blocks: at most one block of zero padding
    const dim3 block(LNR_DENSITY_BLOCK);
#define LNR_F16_GEN_FWD_K(HT, ACT, NH, KT)                                                                                        \
    do {                                                                                                                         \
        const size_t lds = FwdLds<HT, NH, KT>::BYTES;                                                                            \
        /* persistent: every workgroup converts the weights into its LDS once, so no more workgroups than the chip holds */      \
        const int64_t resident = 256 * (int64_t)((size_t)LNR_LDS_LIMIT / lds >= 2 ? 2 : 1);                                      \
        const dim3 grid((unsigned)(blocks < resident ? blocks : resident));                                                      \
        int rc_ = f16_set_lds(mlp_forward_f16_gen_kernel<HT, ACT, NH, KT, LNR_F16_FWD_CT>, lds, "lnr_density_forward");          \
        if (rc_) return rc_;                                                                                                     \
        hipLaunchKernelGGL((mlp_forward_f16_gen_kernel<HT, ACT, NH, KT, LNR_F16_FWD_CT>), grid, block, lds, st, *spec, params, featp, m_pad, \
                           pt->n_points, pt->n_rays_dev, pt->n_rays, pt->n_samples, sigma, pt->clip_flag);                       \
    } while (0)
#define LNR_F16_GEN_FWD(HT, ACT, NH) do { if (kt == 2) LNR_F16_GEN_FWD_K(HT, ACT, NH, 2); else LNR_F16_GEN_FWD_K(HT, ACT, NH, 4); } while (0)
#define LNR_F16_GEN_FWD_N(HT, ACT) do { if (spec->n_hidden == 1) LNR_F16_GEN_FWD(HT, ACT, 1); else if (spec->n_hidden == 2) LNR_F16_GEN_FWD(HT, ACT, 2); else LNR_F16_GEN_FWD(HT, ACT, 3); } while (0)
#define LNR_F16_GEN_FWD_A(HT) do { if (akind == LNR_ACT_RELU) LNR_F16_GEN_FWD_N(HT, LNR_ACT_RELU); else if (akind == LNR_ACT_SINE) LNR_F16_GEN_FWD_N(HT, LNR_ACT_SINE); else LNR_F16_GEN_FWD_N(HT, -1); } while (0)
#define LNR_F16_GEN_FWD_W(ACT) LNR_F16_GEN_FWD(16, ACT, 1)           /* 256 neurons: one hidden layer (lnr_f16_supported) */
    switch (spec->n_neurons / 16) {
        case 1: LNR_F16_GEN_FWD_A(1); break;
        case 2: LNR_F16_GEN_FWD_A(2); break;
        case 4: LNR_F16_GEN_FWD_A(4); break;
        case 8: LNR_F16_GEN_FWD_A(8); break;
        default: if (akind == LNR_ACT_RELU) LNR_F16_GEN_FWD_W(LNR_ACT_RELU); else if (akind == LNR_ACT_SINE) LNR_F16_GEN_FWD_W(LNR_ACT_SINE); else LNR_F16_GEN_FWD_W(-1); break;
    }
#undef LNR_F16_GEN_FWD_W
#undef LNR_F16_GEN_FWD_N
#undef LNR_F16_GEN_FWD_K
#undef LNR_F16_GEN_FWD_A
#undef LNR_F16_GEN_FWD
    return LNR_OK;
}
