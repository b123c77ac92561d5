// One translation unit per hidden width: compile with -DLNR_HT=<n_neurons/16>.
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

int LNR_CAT(lnr_density_fwd_ht, LNR_HT)(const LnrNetSpec* spec, const float* params, const PointSrc* src, float* sigma,
                                         const DensityPlan* plan, hipStream_t st) {
    int rc;
    const dim3 grid(plan->grid), block(64 * plan->waves);
    if (plan->w_lds) {
        rc = set_lds(density_forward_kernel<LNR_HT, true>, plan->lds, "lnr_density_forward");
        if (rc) return rc;
        hipLaunchKernelGGL((density_forward_kernel<LNR_HT, true>), grid, block, plan->lds, st, *spec, params, *src, sigma);
    } else {
        rc = set_lds(density_forward_kernel<LNR_HT, false>, plan->lds, "lnr_density_forward");
        if (rc) return rc;
        hipLaunchKernelGGL((density_forward_kernel<LNR_HT, false>), grid, block, plan->lds, st, *spec, params, *src, sigma);
    }
    return LNR_OK;
}

#define LNR_LAUNCH_BWD(DX, WL, DWK)                                                                                  \
    do {                                                                                                             \
        rc = set_lds(density_backward_kernel<LNR_HT, DX, WL, DWK>, plan->lds, "lnr_density_backward");               \
        if (rc) return rc;                                                                                           \
        hipLaunchKernelGGL((density_backward_kernel<LNR_HT, DX, WL, DWK>), grid, block, plan->lds, st, *spec, params, *src, \
                           d_sigma, grad_table, d_pts, slabs, *sink);                                                \
    } while (0)

int LNR_CAT(lnr_density_bwd_ht, LNR_HT)(const LnrNetSpec* spec, const float* params, const PointSrc* src, const float* d_sigma,
                                         float* grad_table, float* d_pts, float* slabs, const BwdSinkArgs* sink,
                                         const DensityPlan* plan, hipStream_t st) {
    int rc;
    const dim3 grid(plan->grid), block(64 * plan->waves);
#if LNR_HT <= 4
    // the reference's default shape class (one hidden layer, 32 encoded features): register-resident dW1
    if (plan->w_lds && spec->n_hidden == 1 && spec->in_dim == 32) {
        if (d_pts) LNR_LAUNCH_BWD(true, true, 2); else LNR_LAUNCH_BWD(false, true, 2);
        return LNR_OK;
    }
#endif
    if (d_pts) {
        if (plan->w_lds) LNR_LAUNCH_BWD(true, true, 0); else LNR_LAUNCH_BWD(true, false, 0);
    } else {
        if (plan->w_lds) LNR_LAUNCH_BWD(false, true, 0); else LNR_LAUNCH_BWD(false, false, 0);
    }
    return LNR_OK;
}
