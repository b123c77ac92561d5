// One translation unit per hidden width: compile with -DLNR_HT=<n_neurons/16>.  MLP kernels on feature planes.
#include "lnr_density_impl.h"

#ifndef LNR_HT
#error "compile with -DLNR_HT=1|2|4|8|16"
#endif
#define LNR_CAT2(a, b) a##b
#define LNR_CAT(a, b) LNR_CAT2(a, b)

template <typename K>
static int set_lds(K kernel, size_t bytes, const char* who) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        lnr_set_error("%s: hipFuncSetAttribute(%zu) failed: %s", who, bytes, hipGetErrorString(e));
        return LNR_ERR_LAUNCH;
    }
    return LNR_OK;
}

#define LNR_LAUNCH_MF(WL, ACT)                                                                                        \
    do {                                                                                                              \
        rc = set_lds(mlp_forward_kernel<LNR_HT, WL, ACT>, plan->lds, "lnr_density_forward");                          \
        if (rc) return rc;                                                                                            \
        hipLaunchKernelGGL((mlp_forward_kernel<LNR_HT, WL, ACT>), grid, block, plan->lds, st, *spec, params, feat, m_pad, \
                           pt->n_points, pt->n_rays_dev, pt->n_rays, pt->n_samples, sigma, pt->clip_flag);            \
    } while (0)

int LNR_CAT(lnr_mlp_fwd_ht, LNR_HT)(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt,
                                     float* sigma, const DensityPlan* plan, hipStream_t st) {
    int rc;
    const dim3 grid(plan->grid), block(64 * plan->waves);
    const bool relu = spec->activation == LNR_ACT_RELU;
#if LNR_HT <= 4
    if (plan->fast32) {
        rc = set_lds(mlp_forward_relu32_kernel<LNR_HT>, plan->lds, "lnr_density_forward");
        if (rc) return rc;
        hipLaunchKernelGGL((mlp_forward_relu32_kernel<LNR_HT>), grid, block, plan->lds, st, *spec, params, feat, m_pad, pt->n_points,
                           pt->n_rays_dev, pt->n_rays, pt->n_samples, sigma, pt->clip_flag);
        return LNR_OK;
    }
#endif
    const bool sine = spec->activation == LNR_ACT_SINE;        // (compile-time activations: ReLU and Sine, the north star's two; the others by a run-time switch)
    if (plan->w_lds) { if (relu) LNR_LAUNCH_MF(true, LNR_ACT_RELU); else if (sine) LNR_LAUNCH_MF(true, LNR_ACT_SINE); else LNR_LAUNCH_MF(true, -1); }
    else { if (relu) LNR_LAUNCH_MF(false, LNR_ACT_RELU); else if (sine) LNR_LAUNCH_MF(false, LNR_ACT_SINE); else LNR_LAUNCH_MF(false, -1); }
    return LNR_OK;
}

#define LNR_LAUNCH_MB2(WL, ACT, D64)                                                                                  \
    do {                                                                                                              \
        rc = set_lds(mlp_backward_kernel<LNR_HT, WL, ACT, D64>, plan->lds, "lnr_density_backward");         \
        if (rc) return rc;                                                                                            \
        hipLaunchKernelGGL((mlp_backward_kernel<LNR_HT, WL, ACT, D64>), grid, block, plan->lds, st, *spec, params, feat, \
                           m_pad, pt->n_points, pt->n_rays_dev, pt->n_rays, pt->n_samples, d_sigma, dfeat, slabs, want_dfeat); \
    } while (0)

int LNR_CAT(lnr_mlp_bwd_ht, LNR_HT)(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt,
                                     const float* d_sigma, float* dfeat, float* slabs, int want_dfeat, const DensityPlan* plan,
                                     hipStream_t st) {
    int rc;
    const dim3 grid(plan->grid), block(64 * plan->waves);
    const bool relu = spec->activation == LNR_ACT_RELU;
#if LNR_HT <= 4
    if (plan->fast32) {
        rc = set_lds(mlp_backward_relu32_kernel<LNR_HT>, plan->lds, "lnr_density_backward");
        if (rc) return rc;
        hipLaunchKernelGGL((mlp_backward_relu32_kernel<LNR_HT>), grid, block, plan->lds, st, *spec, params, feat, m_pad, pt->n_points,
                           pt->n_rays_dev, pt->n_rays, pt->n_samples, d_sigma, dfeat, slabs, want_dfeat);
        return LNR_OK;
    }
#endif
#define LNR_LAUNCH_MB(WL, ACT) do { if (plan->dw64) LNR_LAUNCH_MB2(WL, ACT, true); else LNR_LAUNCH_MB2(WL, ACT, false); } while (0)
    if (plan->w_lds) { if (relu) LNR_LAUNCH_MB(true, LNR_ACT_RELU); else LNR_LAUNCH_MB(true, -1); }
    else { if (relu) LNR_LAUNCH_MB(false, LNR_ACT_RELU); else LNR_LAUNCH_MB(false, -1); }
    return LNR_OK;
}
