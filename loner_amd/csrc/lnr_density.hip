// Density network: host-side dispatch (C ABI), slab reduction and the MFMA layout self-test.
// The kernels live in lnr_density_impl.h and are instantiated per hidden width in lnr_density_ht.hip.
#include <stdlib.h>

#include "lnr_density_api.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int n_slabs, int n_mlp, float* __restrict__ grad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_mlp) return;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int b = 0;
    for (; b + 3 < n_slabs; b += 4) {
        s0 += slabs[(size_t)b * n_mlp + i]; s1 += slabs[(size_t)(b + 1) * n_mlp + i];
        s2 += slabs[(size_t)(b + 2) * n_mlp + i]; s3 += slabs[(size_t)(b + 3) * n_mlp + i];
    }
    for (; b < n_slabs; ++b) s0 += slabs[(size_t)b * n_mlp + i];
    grad[i] += (s0 + s1) + (s2 + s3);
}

#define LNR_LV_WORDS_HOST (5 * LNR_MAX_LEVELS)
#define LNR_FIX_SCALE 4398046511104.0f   /* 2^42 */

// Second half of the table gradient: workgroup `o` owns floats [o << shift, (o+1) << shift) of the table
// gradient, sums every record addressed to it in LDS and adds the slice to grad_table with plain,
// coalesced read-modify-writes (it is the only writer of that slice).  PAIR: 16-byte {idx, v0, v1, -}
// records (n_features >= 2) or 8-byte {idx, v} records.  Four independent loads per lane are kept in
// flight so that the HBM stream is bandwidth- not latency-bound.
template <int PAIR>
__global__ void __launch_bounds__(512)
table_grad_reduce_kernel(const void* __restrict__ regions_v, const int* __restrict__ counts, int n_src, int nown, int cap,
                         int shift, float* __restrict__ grad_table, int64_t n_table_floats, int debug) {
    // LDS float atomics (ds_add_f32) run at < 1 lane/clk/CU on CDNA4, integer ones ~16x faster: the slice is
    // accumulated in 64-bit fixed point (2^-42 resolution, +-2e6 range; exact and order-independent), converted once.
    extern __shared__ long long acc[];
    __shared__ int cnt[LNR_BWD_MAX_BLOCKS];
    const int o = blockIdx.x;
    const int slice = 1 << shift;
    const uint32_t base = (uint32_t)o << shift;
    for (int i = threadIdx.x; i < slice; i += blockDim.x) acc[i] = 0ll;
    for (int b = threadIdx.x; b < n_src; b += blockDim.x) cnt[b] = counts[(size_t)b * nown + o];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int b = wave; b < n_src; b += nwaves) {
        const int n = cnt[b];
        const size_t off = ((size_t)b * nown + o) * cap;
        if (PAIR) {
            const uint4* r = reinterpret_cast<const uint4*>(regions_v) + off;
            for (int i0 = 0; i0 < n; i0 += 256) {
                uint4 rec[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = i0 + u * 64 + lane; rec[u] = i < n ? r[i] : make_uint4(base, 0u, 0u, 0u); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float v0 = __uint_as_float(rec[u].y), v1 = __uint_as_float(rec[u].z);
                    if (debug & 4) { if (v0 == 1e30f) acc[0] = (long long)v1; continue; }
                    if (v0 != 0.0f) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[rec[u].x - base]), (unsigned long long)__float2ll_rn(v0 * LNR_FIX_SCALE));
                    if (v1 != 0.0f) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[rec[u].x - base + 1]), (unsigned long long)__float2ll_rn(v1 * LNR_FIX_SCALE));
                }
            }
        } else {
            const uint2* r = reinterpret_cast<const uint2*>(regions_v) + off;
            for (int i0 = 0; i0 < n; i0 += 256) {
                uint2 rec[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = i0 + u * 64 + lane; rec[u] = i < n ? r[i] : make_uint2(base, 0u); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float v = __uint_as_float(rec[u].y);
                    if (v != 0.0f) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[rec[u].x - base]), (unsigned long long)__float2ll_rn(v * LNR_FIX_SCALE));
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < slice; i += blockDim.x) {
        const int64_t gi = (int64_t)base + i;
        const long long q = acc[i];
        if (gi < n_table_floats && q != 0ll) grad_table[gi] += (float)((double)q * (1.0 / (double)LNR_FIX_SCALE));
    }
}

struct SinkLayout {
    int nown, nown_padded, cap, shift, rec_bytes;
    size_t slabs_bytes, counts_bytes, regions_bytes;
};

// Region capacity: an uncombined level of `level_size` entries spreads n_points*8 entry updates over
// level_size*F/slice owners, i.e. n_points*8*slice/level_size float records per owner (F cancels), divided
// over the source workgroups; the busiest (smallest uncombined) level sets the capacity, +25 % + 256 slack.
// Anything beyond that (skewed data) falls back to global atomics, so this is a performance knob only.
static SinkLayout sink_layout(const LnrNetSpec* spec, int64_t n_points) {
    SinkLayout L;
    const int64_t n_table = spec->n_params - spec->n_mlp_params;
    L.shift = LNR_SLICE_SHIFT;
    L.nown = (int)((n_table + (1 << L.shift) - 1) >> L.shift);
    L.nown_padded = (L.nown + 3) & ~3;
    L.rec_bytes = spec->n_features >= 2 ? 16 : 8;
    const double rec_per_touch = spec->n_features >= 4 ? (spec->n_features == 8 ? 2.0 : 2.0) : 1.0;   // F=4/8: two pair-records per 4 features
    double per_owner = 0.0;
    if (spec->encoding == LNR_ENC_HASHGRID) {
        for (int l = 0; l < spec->n_levels; ++l) {
            double owners = (double)spec->level_size[l] * spec->n_features / (double)(1 << L.shift);
            if (owners < 1.0) owners = 1.0;
            double r = (double)n_points * 8.0 * rec_per_touch * (spec->n_features == 8 ? 2.0 : 1.0) / owners;
            if (spec->level_scale[l] < LNR_COMBINE_SCALE_MAX) r *= 0.25;      // run-length combined levels emit far fewer records
            if (r > per_owner) per_owner = r;
        }
    }
    int64_t cap = (int64_t)(per_owner / LNR_BWD_MAX_BLOCKS * 1.25) + 256;
    const int64_t budget_cap = L.nown > 0 ? (int64_t)(LNR_REGION_BUDGET / ((uint64_t)L.rec_bytes * LNR_BWD_MAX_BLOCKS * (uint64_t)L.nown)) : 0;
    if (cap > budget_cap) cap = budget_cap;
    if (cap < 64) cap = 64;
    L.cap = (int)cap;
    L.slabs_bytes = (size_t)LNR_BWD_MAX_BLOCKS * (size_t)spec->n_mlp_params * sizeof(float);
    L.counts_bytes = ((size_t)LNR_BWD_MAX_BLOCKS * (size_t)L.nown * sizeof(int) + 255) & ~(size_t)255;
    L.regions_bytes = (size_t)LNR_BWD_MAX_BLOCKS * (size_t)L.nown * (size_t)L.cap * (size_t)L.rec_bytes;
    return L;
}

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
    return (LNR_LV_WORDS_HOST + (size_t)sink_layout(s, 0).nown_padded + (w_lds ? 2 : 1) * (size_t)s->n_mlp_params +
            (size_t)waves * scratch) * sizeof(float);
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
        const int64_t max_blocks = backward ? LNR_BWD_MAX_BLOCKS : LNR_DENSITY_MAX_BLOCKS;
        if (blocks > max_blocks) blocks = max_blocks;
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

extern "C" size_t lnr_density_backward_workspace(const LnrNetSpec* spec, int64_t n_points) {
    if (!spec) return 0;
    const SinkLayout L = sink_layout(spec, n_points);
    return L.slabs_bytes + L.counts_bytes + L.regions_bytes;
}

extern "C" int lnr_density_backward(const LnrNetSpec* spec, const float* params, const float* pts, int64_t n_points,
                                    const float* rays, const float* z, int32_t n_rays, int32_t n_samples,
                                    const int32_t* n_rays_dev, const float* d_sigma, float* grad_params, float* d_pts,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_spec(spec, "lnr_density_backward");
    if (rc) return rc;
    LNR_REQUIRE(params && d_sigma && grad_params && workspace, "lnr_density_backward: null argument");
    const int64_t cap_points = pts ? n_points : (int64_t)n_rays * n_samples;
    if (workspace_bytes < lnr_density_backward_workspace(spec, cap_points)) {
        lnr_set_error("lnr_density_backward: workspace %zu < %zu", workspace_bytes, lnr_density_backward_workspace(spec, cap_points));
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
    const SinkLayout L = sink_layout(spec, cap_points);
    float* slabs = (float*)workspace;
    BwdSinkArgs sink;
    sink.counts = (int*)((char*)workspace + L.slabs_bytes);
    sink.regions = (void*)((char*)workspace + L.slabs_bytes + L.counts_bytes);
    sink.nown = L.nown; sink.nown_padded = L.nown_padded; sink.cap = L.cap; sink.shift = L.shift;
    { const char* e = getenv("LNR_DEBUG"); sink.debug = e ? atoi(e) : 0; }
    if (sink.debug & 32) sink.cap = 0;      // test hook: every record takes the global-atomic fallback path
    sink.combine_scale_max = LNR_COMBINE_SCALE_MAX;   // levels up to ~2^11 cells per axis: consecutive samples of a ray share cells
    float* grad_table = grad_params + spec->n_mlp_params;
    switch (spec->n_neurons / 16) {
        case 1: rc = lnr_density_bwd_ht1(spec, params, &src, d_sigma, grad_table, d_pts, slabs, &sink, &plan, st); break;
        case 2: rc = lnr_density_bwd_ht2(spec, params, &src, d_sigma, grad_table, d_pts, slabs, &sink, &plan, st); break;
        case 4: rc = lnr_density_bwd_ht4(spec, params, &src, d_sigma, grad_table, d_pts, slabs, &sink, &plan, st); break;
        case 8: rc = lnr_density_bwd_ht8(spec, params, &src, d_sigma, grad_table, d_pts, slabs, &sink, &plan, st); break;
        default: rc = lnr_density_bwd_ht16(spec, params, &src, d_sigma, grad_table, d_pts, slabs, &sink, &plan, st); break;
    }
    if (rc) return rc;
    LNR_CHECK_LAUNCH("lnr_density_backward");
    if (L.nown > 0) {
        const size_t lds = ((size_t)1 << L.shift) * sizeof(long long);
        const int64_t n_table = spec->n_params - spec->n_mlp_params;
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(table_grad_reduce_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(table_grad_reduce_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e0 != hipSuccess || e1 != hipSuccess) { lnr_set_error("lnr_density_backward: hipFuncSetAttribute failed"); return LNR_ERR_LAUNCH; }
        if (L.rec_bytes == 16)
            hipLaunchKernelGGL(table_grad_reduce_kernel<1>, dim3(L.nown), dim3(512), lds, st, sink.regions, sink.counts, plan.grid, L.nown,
                               L.cap, L.shift, grad_table, n_table, sink.debug);
        else
            hipLaunchKernelGGL(table_grad_reduce_kernel<0>, dim3(L.nown), dim3(512), lds, st, sink.regions, sink.counts, plan.grid, L.nown,
                               L.cap, L.shift, grad_table, n_table, sink.debug);
        LNR_CHECK_LAUNCH("lnr_density_backward(table reduce)");
    }
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
