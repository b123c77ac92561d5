// One translation unit per (hidden width, depth): compile with -DLNR_HT=<n_neurons/16> -DLNR_NH=<n_hidden_layers>.
// Host dispatch of mlp_backward_regs_kernel (lnr_density_regs.h) over the first layer's K blocks, the weights' home and the activation.
#include "lnr_density_regs.h"

#if !defined(LNR_HT) || !defined(LNR_NH)
#error "compile with -DLNR_HT=4|8|16 -DLNR_NH=1|2|3"
#endif
#define LNR_CAT4_(a, b, c, d) a##b##c##d
#define LNR_CAT4(a, b, c, d) LNR_CAT4_(a, b, c, d)

template <int KT1M, int WM, int ACT>
static int launch_regs(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                       float* dfeat, float* slabs, int want_dfeat, const DensityPlan* plan, hipStream_t st) {
    auto kernel = mlp_backward_regs_kernel<LNR_HT, LNR_NH, KT1M, WM, ACT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan->lds);
    if (e != hipSuccess) {
        lnr_set_error("lnr_density_backward: hipFuncSetAttribute(%zu) failed: %s", plan->lds, hipGetErrorString(e));
        return LNR_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kernel, dim3(plan->grid), dim3(LNR_DENSITY_BLOCK), plan->lds, st, *spec, params, feat, m_pad, pt->n_points, pt->n_rays_dev,
                       pt->n_rays, pt->n_samples, d_sigma, dfeat, slabs, want_dfeat);
    return LNR_OK;
}

template <int KT1M>
static int launch_regs_k(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                         float* dfeat, float* slabs, int want_dfeat, const DensityPlan* plan, hipStream_t st) {
    const bool relu = spec->activation == LNR_ACT_RELU, sine = spec->activation == LNR_ACT_SINE;
#define LNR_REGS_GO(WM) (relu ? launch_regs<KT1M, WM, LNR_ACT_RELU>(spec, params, feat, m_pad, pt, d_sigma, dfeat, slabs, want_dfeat, plan, st) \
                         : sine ? launch_regs<KT1M, WM, LNR_ACT_SINE>(spec, params, feat, m_pad, pt, d_sigma, dfeat, slabs, want_dfeat, plan, st) \
                                : launch_regs<KT1M, WM, -1>(spec, params, feat, m_pad, pt, d_sigma, dfeat, slabs, want_dfeat, plan, st))
    if (plan->w_lds == 1) return LNR_REGS_GO(1);
#if LNR_NH > 1
    if (plan->w_lds == 2) return LNR_REGS_GO(2);
#endif
    return LNR_REGS_GO(0);
#undef LNR_REGS_GO
}

int LNR_CAT4(lnr_mlp_bwd_regs_ht, LNR_HT, _nh, LNR_NH)(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad,
                                                       const MlpPoints* pt, const float* d_sigma, float* dfeat, float* slabs, int want_dfeat,
                                                       const DensityPlan* plan, hipStream_t st) {
    const int kt1 = spec->in_dim / 16;
    if (kt1 <= 2) return launch_regs_k<2>(spec, params, feat, m_pad, pt, d_sigma, dfeat, slabs, want_dfeat, plan, st);
    if (kt1 <= 5) return launch_regs_k<5>(spec, params, feat, m_pad, pt, d_sigma, dfeat, slabs, want_dfeat, plan, st);
    return launch_regs_k<8>(spec, params, feat, m_pad, pt, d_sigma, dfeat, slabs, want_dfeat, plan, st);
}
