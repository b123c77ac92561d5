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
#define LNR_SLICE_SHIFT 13            // 8192 floats (32 KB of LDS) per owner
#define LNR_REGION_BUDGET (24ull << 30)
#define LNR_COMBINE_SCALE_MAX 3000.0f

// launch plan chosen by the dispatcher
struct DensityPlan {
    int grid;        // workgroups
    int waves;       // waves per workgroup (1, 2 or 4)
    int w_lds;       // 1: MLP matrices staged in LDS, 0: read from global memory
    size_t lds;      // dynamic LDS bytes
    int fast32;      // 1: the register-resident kernels for 32 features -> <= 64 ReLU neurons -> 1
};

// point count description for the MLP kernels (features come from planes)
struct MlpPoints {
    int64_t n_points;            // explicit point count (pts mode)
    const int32_t* n_rays_dev;   // non-null: live ray count on the device (rays mode)
    int32_t n_rays, n_samples;
};

#define LNR_DECLARE_HT(HT)                                                                                              \
    int lnr_mlp_fwd_ht##HT(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, \
                           float* sigma, const DensityPlan* plan, hipStream_t st);                                      \
    int lnr_mlp_bwd_ht##HT(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, \
                           const float* d_sigma, float* dfeat, float* slabs, int want_dfeat, const DensityPlan* plan,   \
                           hipStream_t st);
LNR_DECLARE_HT(1)
LNR_DECLARE_HT(2)
LNR_DECLARE_HT(4)
LNR_DECLARE_HT(8)
LNR_DECLARE_HT(16)

// level-major encoding (lnr_encode.hip)
int lnr_encode_forward(const LnrNetSpec* spec, const float* params, const PointSrc* src, int64_t cap_points, float* feat,
                       int64_t m_pad, hipStream_t st);
int lnr_encode_backward(const LnrNetSpec* spec, const float* params, const PointSrc* src, int64_t cap_points, const float* dfeat,
                        float* dxl, int64_t m_pad, float* grad_table, void* regions, int* counts, int bpg, int maxo, int cap,
                        int shift, int debug, float* d_pts, hipStream_t st);
