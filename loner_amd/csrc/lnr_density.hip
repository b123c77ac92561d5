// Density network: host-side dispatch (C ABI), slab reduction and the MFMA layout self-test.
// The kernels live in lnr_density_impl.h and are instantiated per hidden width in lnr_density_ht.hip.
#include "lnr_density_api.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int n_slabs, int n_mlp, float* __restrict__ grad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_mlp) return;
    float s = 0.0f;
    for (int b = 0; b < n_slabs; ++b) s += slabs[(size_t)b * n_mlp + i];
    grad[i] += s;
}

#define LNR_LV_WORDS_HOST (5 * LNR_MAX_LEVELS)

static int check_spec(const LnrNetSpec* spec, const char* who) {
    LNR_REQUIRE(spec != nullptr, "%s: null spec", who);
    LNR_REQUIRE(spec->in_dim > 0 && spec->n_mlp_params > 0, "%s: spec not finalized (call lnr_net_spec_finalize)", who);
    const int ht = spec->n_neurons / 16;
    if (!(ht == 1 || ht == 2 || ht == 4 || ht == 8 || ht == 16) || spec->n_neurons % 16) {
        lnr_set_error("%s: n_neurons=%d not supported (need 16, 32, 64, 128 or 256)", who, spec->n_neurons);
        return LNR_ERR_UNSUPPORTED;
    }
    return LNR_OK;
}

static size_t fwd_lds(const LnrNetSpec* s, int w_lds) {
    return (LNR_LV_WORDS_HOST + (w_lds ? (size_t)s->n_mlp_params : 0)) * sizeof(float);
}
static size_t bwd_lds(const LnrNetSpec* s, int w_lds, int waves) {
    const size_t H = s->n_neurons;
    const size_t scratch = H * 16 + (size_t)s->in_dim * 16 + (s->n_hidden > 1 ? (size_t)(s->n_hidden + 1) * H * 16 : 0);
    return (LNR_LV_WORDS_HOST + (w_lds ? 2 : 1) * (size_t)s->n_mlp_params + (size_t)waves * scratch) * sizeof(float);
}

// Pick the launch shape: prefer weights in LDS and 4 waves per workgroup; fall back to fewer waves, then to
// weights read from global memory, until the workgroup fits the 160 KB LDS of a CDNA4 CU.
static int plan_launch(const LnrNetSpec* spec, int64_t n_points, bool backward, DensityPlan* plan, const char* who) {
    static const int opts[6][2] = {{1, 4}, {1, 2}, {1, 1}, {0, 4}, {0, 2}, {0, 1}};
    for (int o = 0; o < 6; ++o) {
        const int w_lds = opts[o][0], waves = opts[o][1];
        if (!backward && waves != 4) continue;           // forward scratch does not depend on the wave count
        const size_t lds = backward ? bwd_lds(spec, w_lds, waves) : fwd_lds(spec, w_lds);
        // keep two workgroups per CU resident when the weights are LDS-staged copies (latency hiding for the gathers)
        if (lds > (size_t)LNR_LDS_LIMIT) continue;
        plan->w_lds = w_lds; plan->waves = waves; plan->lds = lds;
        const int64_t tiles = (n_points + 15) / 16;
        int64_t blocks = (tiles + waves - 1) / waves;
        if (blocks > LNR_DENSITY_MAX_BLOCKS) blocks = LNR_DENSITY_MAX_BLOCKS;
        if (blocks < 1) blocks = 1;
        plan->grid = (int)blocks;
        return LNR_OK;
    }
    lnr_set_error("%s: network (n_neurons=%d, n_hidden_layers=%d, in_dim=%d: %d MLP weights) does not fit the 160 KB LDS of a CU "
                  "even with one wave per workgroup; not supported by the fp32 kernels", who, spec->n_neurons, spec->n_hidden,
                  spec->in_dim, spec->n_mlp_params);
    return LNR_ERR_UNSUPPORTED;
}

static int make_src(PointSrc* s, const float* pts, int64_t n_points, const float* rays, const float* z,
                    int32_t n_rays, int32_t n_samples, const int32_t* n_rays_dev, const char* who) {
    if (pts != nullptr) {
        LNR_REQUIRE(n_points >= 0, "%s: negative n_points", who);
        *s = PointSrc{pts, nullptr, nullptr, 1, n_points, 0, nullptr};
    } else {
        LNR_REQUIRE(rays != nullptr && z != nullptr, "%s: need either pts or (rays, z)", who);
        LNR_REQUIRE(n_rays >= 0 && n_samples > 0, "%s: bad n_rays/n_samples", who);
        *s = PointSrc{nullptr, rays, z, n_samples, 0, n_rays, n_rays_dev};
    }
    return LNR_OK;
}

extern "C" int lnr_density_forward(const LnrNetSpec* spec, const float* params, const float* pts, int64_t n_points,
                                   const float* rays, const float* z, int32_t n_rays, int32_t n_samples,
                                   const int32_t* n_rays_dev, float* sigma, void* stream) {
    int rc = check_spec(spec, "lnr_density_forward");
    if (rc) return rc;
    LNR_REQUIRE(params && sigma, "lnr_density_forward: null params/sigma");
    PointSrc src;
    rc = make_src(&src, pts, n_points, rays, z, n_rays, n_samples, n_rays_dev, "lnr_density_forward");
    if (rc) return rc;
    const int64_t cap = pts ? n_points : (int64_t)n_rays * n_samples;
    if (cap == 0) return LNR_OK;
    DensityPlan plan;
    rc = plan_launch(spec, cap, false, &plan, "lnr_density_forward");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    switch (spec->n_neurons / 16) {
        case 1: rc = lnr_density_fwd_ht1(spec, params, &src, sigma, &plan, st); break;
        case 2: rc = lnr_density_fwd_ht2(spec, params, &src, sigma, &plan, st); break;
        case 4: rc = lnr_density_fwd_ht4(spec, params, &src, sigma, &plan, st); break;
        case 8: rc = lnr_density_fwd_ht8(spec, params, &src, sigma, &plan, st); break;
        default: rc = lnr_density_fwd_ht16(spec, params, &src, sigma, &plan, st); break;
    }
    if (rc) return rc;
    LNR_CHECK_LAUNCH("lnr_density_forward");
    return LNR_OK;
}

extern "C" size_t lnr_density_backward_workspace(const LnrNetSpec* spec) {
    if (!spec) return 0;
    return (size_t)LNR_DENSITY_MAX_BLOCKS * (size_t)spec->n_mlp_params * sizeof(float);
}

extern "C" int lnr_density_backward(const LnrNetSpec* spec, const float* params, const float* pts, int64_t n_points,
                                    const float* rays, const float* z, int32_t n_rays, int32_t n_samples,
                                    const int32_t* n_rays_dev, const float* d_sigma, float* grad_params, float* d_pts,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_spec(spec, "lnr_density_backward");
    if (rc) return rc;
    LNR_REQUIRE(params && d_sigma && grad_params && workspace, "lnr_density_backward: null argument");
    if (workspace_bytes < lnr_density_backward_workspace(spec)) {
        lnr_set_error("lnr_density_backward: workspace %zu < %zu", workspace_bytes, lnr_density_backward_workspace(spec));
        return LNR_ERR_WORKSPACE;
    }
    PointSrc src;
    rc = make_src(&src, pts, n_points, rays, z, n_rays, n_samples, n_rays_dev, "lnr_density_backward");
    if (rc) return rc;
    const int64_t cap = pts ? n_points : (int64_t)n_rays * n_samples;
    if (cap == 0) return LNR_OK;
    DensityPlan plan;
    rc = plan_launch(spec, cap, true, &plan, "lnr_density_backward");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* slabs = (float*)workspace;
    float* grad_table = grad_params + spec->n_mlp_params;
    switch (spec->n_neurons / 16) {
        case 1: rc = lnr_density_bwd_ht1(spec, params, &src, d_sigma, grad_table, d_pts, slabs, &plan, st); break;
        case 2: rc = lnr_density_bwd_ht2(spec, params, &src, d_sigma, grad_table, d_pts, slabs, &plan, st); break;
        case 4: rc = lnr_density_bwd_ht4(spec, params, &src, d_sigma, grad_table, d_pts, slabs, &plan, st); break;
        case 8: rc = lnr_density_bwd_ht8(spec, params, &src, d_sigma, grad_table, d_pts, slabs, &plan, st); break;
        default: rc = lnr_density_bwd_ht16(spec, params, &src, d_sigma, grad_table, d_pts, slabs, &plan, st); break;
    }
    if (rc) return rc;
    LNR_CHECK_LAUNCH("lnr_density_backward");
    const int n_mlp = spec->n_mlp_params;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(lnr_div_up(n_mlp, 256)), dim3(256), 0, st, slabs, plan.grid, n_mlp, grad_params);
    LNR_CHECK_LAUNCH("lnr_density_backward(reduce)");
    return LNR_OK;
}

// ------------------------------------------------------------------------------------------------
// MFMA layout self-test: D = A(16x4) * B(4x16) with asymmetric operands, checked against the
// layout the kernels assume (A: lane->A[l&15][l>>4]; B: lane->B[l>>4][l&15]; D: reg r ->
// D[4*(l>>4)+r][l&15]).
// ------------------------------------------------------------------------------------------------
__global__ void selftest_mfma_kernel(float* out) {
    const int lane = threadIdx.x;
    const int i = lane & 15, k = lane >> 4;
    const float a = (float)(i * 7 + k * 3 + 1) * 0.25f;          // A[i][k]
    const float b = (float)((lane & 15) * 5 - k * 11 + 2) * 0.5f; // B[k][j], j = lane&15
    f32x4 d = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
    float err = 0.0f;
    const int j = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float ref = 0.0f;
        for (int kk = 0; kk < 4; ++kk) ref += ((float)(row * 7 + kk * 3 + 1) * 0.25f) * ((float)(j * 5 - kk * 11 + 2) * 0.5f);
        err = fmaxf(err, fabsf(ref - d[r]));
    }
    err = wave_max(err);
    if (lane == 0) out[0] = err;
}

extern "C" int lnr_selftest_mfma(float* out, void* stream) {
    LNR_REQUIRE(out != nullptr, "lnr_selftest_mfma: null out");
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    LNR_CHECK_LAUNCH("lnr_selftest_mfma");
    return LNR_OK;
}
