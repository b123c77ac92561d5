// Level-major input encoding (forward and backward) for the density network (gfx950).
//
// Why level-major: the default hash grid is 29.7 MB, the L2 of an XCD 4 MB.  A kernel that walks all 16 levels per
// sample thrashes L2 (PMC: 3.5 GB of memory-side fetches per forward launch for 0.13 GB of algorithmic bytes).
// Here the grid is ordered level by level (blockIdx / blocks_per_group = level), so at any moment the whole chip
// works on ONE level whose table (<= 2 MB) stays resident in every XCD's L2, and each gather is an L2 hit.
// Features travel to the MLP kernels as [feature plane][sample] arrays in HBM (coalesced both ways).
//
// Backward: one thread per (sample, level) re-derives the cell, turns d_feature into table-gradient records
// (run-length combined over the 64 consecutive samples of a wave on coarse levels), and writes its contribution to
// d/dx as one plane per level; records are reduced by table_grad_reduce2_kernel (lnr_density.hip).
#include "lnr_encoding.h"

#define ENC_BLOCK 256

// ------------------------------------------------------------------------------------------------ forward
template <int F>
__device__ __forceinline__ void encode_level(const float* lvt, const float* __restrict__ table, int lv, const float x[3], float out[F]) {
    LevelCell c = level_cell(lvt, lv, x);
#pragma unroll
    for (int f = 0; f < F; ++f) out[f] = 0.0f;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const float w = corner_weight(c, corner);
        const float* e = table + (size_t)cell_entry(c, corner) * F;
        if constexpr (F == 1) {
            out[0] += w * e[0];
        } else if constexpr (F == 2) {
            const float2 v = *reinterpret_cast<const float2*>(e);
            out[0] += w * v.x; out[1] += w * v.y;
        } else {
#pragma unroll
            for (int q = 0; q < F / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(e + 4 * q);
                out[4 * q] += w * v.x; out[4 * q + 1] += w * v.y; out[4 * q + 2] += w * v.z; out[4 * q + 3] += w * v.w;
            }
        }
    }
}

template <int F>
__global__ void __launch_bounds__(ENC_BLOCK)
encode_forward_kernel(const LnrNetSpec spec, const float* __restrict__ table, const PointSrc src, float* __restrict__ feat,
                      int64_t m_pad, int bpg) {
    __shared__ float lvt[LNR_LV_WORDS];
    stage_level_tables(spec, lvt);
    __syncthreads();
    const int group = blockIdx.x / bpg, chunk = blockIdx.x % bpg;
    const int64_t M = live_points(src);
    const int64_t M16 = (M + 15) / 16 * 16;      // the MLP kernels read whole 16-sample tiles: zero the ragged tail
    const int nf = spec.encoding == LNR_ENC_HASHGRID ? F : 4;
    for (int64_t m = (int64_t)chunk * ENC_BLOCK + threadIdx.x; m < M16; m += (int64_t)bpg * ENC_BLOCK) {
        if (m >= M) {
            for (int f = 0; f < nf; ++f)
                if (group * nf + f < spec.enc_dim) feat[(size_t)(group * nf + f) * m_pad + m] = 0.0f;
            continue;
        }
        float x[3];
        load_unit_point(src, m, x);
        if (spec.encoding == LNR_ENC_HASHGRID) {
            float out[F];
            encode_level<F>(lvt, table, group, x, out);
#pragma unroll
            for (int f = 0; f < F; ++f) feat[(size_t)(group * F + f) * m_pad + m] = out[f];
        } else {
            float out[4];
            freq_features4(spec, x, 4 * group, out);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * group + r < spec.enc_dim) feat[(size_t)(4 * group + r) * m_pad + m] = out[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
struct EncSink {
    float* grad_table;
    void* regions;          // [block][maxo][cap] records
    int* counts;            // [block][maxo]
    int maxo, cap, shift;
    float combine_scale_max;
    int debug;
};

__device__ __forceinline__ void enc_emit(const EncSink& s, int* cursors, int first_owner, uint32_t fidx, float v0, float v1, bool pair) {
    if (v0 == 0.0f && (!pair || v1 == 0.0f)) return;
    const int local = (int)(fidx >> s.shift) - first_owner;
    int slot = s.cap;
    if (local >= 0 && local < s.maxo && s.cap > 0) slot = atomicAdd(&cursors[local], 1);
    if (slot < s.cap) {
        const size_t at = ((size_t)blockIdx.x * s.maxo + local) * s.cap + slot;
        if (pair) reinterpret_cast<uint4*>(s.regions)[at] = make_uint4(fidx, __float_as_uint(v0), __float_as_uint(v1), 0u);
        else reinterpret_cast<uint2*>(s.regions)[at] = make_uint2(fidx, __float_as_uint(v0));
    } else {
        atomicAdd(s.grad_table + fidx, v0);
        if (pair) atomicAdd(s.grad_table + fidx + 1, v1);
    }
}

// segmented sum over the 64 lanes of a wave: lanes with equal consecutive `seg` ids are summed into the run head
__device__ __forceinline__ float wave_run_sum(float v, int seg, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int s2 = __shfl_down(seg, o, 64);
        const float t = __shfl_down(v, o, 64);
        if (lane + o < 64 && s2 == seg) v += t;
    }
    return v;
}

template <int F, bool WANT_DX>
__global__ void __launch_bounds__(ENC_BLOCK)
encode_backward_kernel(const LnrNetSpec spec, const float* __restrict__ table, const PointSrc src, const float* __restrict__ dfeat,
                       float* __restrict__ dxl, int64_t m_pad, int bpg, const EncSink sink) {
    extern __shared__ int cursors[];
    __shared__ float lvt[LNR_LV_WORDS];
    stage_level_tables(spec, lvt);
    for (int i = threadIdx.x; i < sink.maxo; i += ENC_BLOCK) cursors[i] = 0;
    __syncthreads();
    const int lv = blockIdx.x / bpg, chunk = blockIdx.x % bpg;
    const int lane = threadIdx.x & 63;
    const int64_t M = live_points(src);
    const int64_t m_round = (M + 63) / 64 * 64;
    const int first_owner = (int)(((uint64_t)spec.level_offset[lv] * F) >> sink.shift);
    const bool combine = spec.level_scale[lv] < sink.combine_scale_max;
    constexpr bool PAIR = F >= 2;
    for (int64_t m = (int64_t)chunk * ENC_BLOCK + threadIdx.x; m < m_round; m += (int64_t)bpg * ENC_BLOCK) {
        const bool live = m < M;
        float g[F];
        bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) { g[f] = live ? dfeat[(size_t)(lv * F + f) * m_pad + m] : 0.0f; any |= (g[f] != 0.0f); }
        float dx[3] = {0.0f, 0.0f, 0.0f};
        if (__ballot(any) != 0ull) {                 // wave-uniform
            float x[3];
            load_unit_point(src, live ? m : M - 1, x);
            LevelCell c = level_cell(lvt, lv, x);
            int seg = 0;
            bool head = true;
            if (combine) {
                // runs = consecutive samples in the same CELL (not merely the same hashed entry)
                const uint32_t k1 = c.base[0] | (c.base[1] << 16), k2 = c.base[2];
                const uint32_t p1 = __shfl_up(k1, 1, 64), p2 = __shfl_up(k2, 1, 64);
                head = (lane == 0) || (p1 != k1) || (p2 != k2);
                seg = head ? 1 : 0;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(seg, o, 64); if (lane >= o) seg += t; }
            }
            float dfrac[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const float w = corner_weight(c, corner);
                const uint32_t e = cell_entry(c, corner) * F;
                float v[F];
#pragma unroll
                for (int f = 0; f < F; ++f) v[f] = w * g[f];
                if (combine) {
                    // runs are defined by corner 0's entry: within a run all samples share the cell, hence every corner
#pragma unroll
                    for (int f = 0; f < F; ++f) v[f] = wave_run_sum(v[f], seg, lane);
                }
                if (!combine || head) {
                    if (!(sink.debug & 2)) {
                        if constexpr (F == 1) enc_emit(sink, cursors, first_owner, e, v[0], 0.0f, false);
                        else {
#pragma unroll
                            for (int f = 0; f < F; f += 2) enc_emit(sink, cursors, first_owner, e + f, v[f], v[f + 1], true);
                        }
                    }
                }
                if constexpr (WANT_DX) {
                    if (any) {
                        float dot = 0.0f;
#pragma unroll
                        for (int f = 0; f < F; ++f) dot += g[f] * table[(size_t)e + f];
                        const float wx = (corner & 1) ? c.frac[0] : 1.0f - c.frac[0];
                        const float wy = (corner & 2) ? c.frac[1] : 1.0f - c.frac[1];
                        const float wz = (corner & 4) ? c.frac[2] : 1.0f - c.frac[2];
                        dfrac[0] += ((corner & 1) ? dot : -dot) * wy * wz;
                        dfrac[1] += ((corner & 2) ? dot : -dot) * wx * wz;
                        dfrac[2] += ((corner & 4) ? dot : -dot) * wx * wy;
                    }
                }
            }
            if constexpr (WANT_DX) {
#pragma unroll
                for (int d = 0; d < 3; ++d) dx[d] = dfrac[d] * c.scale;
            }
        }
        if constexpr (WANT_DX) {
            if (live) {
#pragma unroll
                for (int d = 0; d < 3; ++d) dxl[(size_t)(lv * 3 + d) * m_pad + m] = dx[d];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < sink.maxo; i += ENC_BLOCK)
        sink.counts[(size_t)blockIdx.x * sink.maxo + i] = cursors[i] < sink.cap ? cursors[i] : sink.cap;
}

// Frequency encoding has no table: backward is only the input gradient, one plane group for all features.
__global__ void __launch_bounds__(ENC_BLOCK)
freq_backward_kernel(const LnrNetSpec spec, const PointSrc src, const float* __restrict__ dfeat, float* __restrict__ dxl, int64_t m_pad) {
    const int64_t M = live_points(src);
    for (int64_t m = (int64_t)blockIdx.x * ENC_BLOCK + threadIdx.x; m < M; m += (int64_t)gridDim.x * ENC_BLOCK) {
        float x[3];
        load_unit_point(src, m, x);
        float dx[3] = {0.0f, 0.0f, 0.0f};
        for (int k0 = 0; k0 < spec.enc_dim; k0 += 4) {
            float d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = (k0 + r < spec.enc_dim) ? dfeat[(size_t)(k0 + r) * m_pad + m] : 0.0f;
            freq_features4_bwd(spec, x, k0, d, dx);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) dxl[(size_t)d * m_pad + m] = dx[d];
    }
}

// d_pts[m] = 0.5 * sum over groups of dxl[group][:, m]     (x = (xyz+1)/2)
__global__ void __launch_bounds__(ENC_BLOCK)
sum_dx_planes_kernel(const float* __restrict__ dxl, int n_groups, int64_t m_pad, const PointSrc src, float* __restrict__ d_pts) {
    const int64_t M = live_points(src);
    for (int64_t m = (int64_t)blockIdx.x * ENC_BLOCK + threadIdx.x; m < M; m += (int64_t)gridDim.x * ENC_BLOCK) {
        float a[3] = {0.0f, 0.0f, 0.0f};
        for (int gI = 0; gI < n_groups; ++gI) {
#pragma unroll
            for (int d = 0; d < 3; ++d) a[d] += dxl[(size_t)(gI * 3 + d) * m_pad + m];
        }
        d_pts[3 * m] = 0.5f * a[0]; d_pts[3 * m + 1] = 0.5f * a[1]; d_pts[3 * m + 2] = 0.5f * a[2];
    }
}

// ------------------------------------------------------------------------------------------------ host launchers
int lnr_encode_forward(const LnrNetSpec* spec, const float* params, const PointSrc* src, int64_t cap_points, float* feat,
                       int64_t m_pad, hipStream_t st) {
    const float* table = params + spec->n_mlp_params;
    const int n_groups = spec->encoding == LNR_ENC_HASHGRID ? spec->n_levels : (spec->enc_dim + 3) / 4;
    int64_t bpg = (cap_points + ENC_BLOCK * 4 - 1) / (ENC_BLOCK * 4);      // ~4 samples per thread
    if (bpg < 1) bpg = 1;
    if (bpg > 2048) bpg = 2048;
    const dim3 grid((unsigned)(n_groups * bpg)), block(ENC_BLOCK);
    const int F = spec->encoding == LNR_ENC_HASHGRID ? spec->n_features : 1;
    switch (F) {
        case 1: hipLaunchKernelGGL(encode_forward_kernel<1>, grid, block, 0, st, *spec, table, *src, feat, m_pad, (int)bpg); break;
        case 2: hipLaunchKernelGGL(encode_forward_kernel<2>, grid, block, 0, st, *spec, table, *src, feat, m_pad, (int)bpg); break;
        case 4: hipLaunchKernelGGL(encode_forward_kernel<4>, grid, block, 0, st, *spec, table, *src, feat, m_pad, (int)bpg); break;
        default: hipLaunchKernelGGL(encode_forward_kernel<8>, grid, block, 0, st, *spec, table, *src, feat, m_pad, (int)bpg); break;
    }
    return LNR_OK;
}

int lnr_encode_backward(const LnrNetSpec* spec, const float* params, const PointSrc* src, int64_t cap_points, const float* dfeat,
                        float* dxl, int64_t m_pad, float* grad_table, void* regions, int* counts, int bpg, int maxo, int cap,
                        int shift, int debug, float* d_pts, hipStream_t st) {
    const float* table = params + spec->n_mlp_params;
    int n_groups = 1;
    if (spec->encoding == LNR_ENC_HASHGRID) {
        n_groups = spec->n_levels;
        EncSink sink;
        sink.grad_table = grad_table; sink.regions = regions; sink.counts = counts; sink.maxo = maxo; sink.cap = cap; sink.shift = shift;
        sink.combine_scale_max = LNR_COMBINE_SCALE_MAX; sink.debug = debug;
        const dim3 grid((unsigned)(n_groups * bpg)), block(ENC_BLOCK);
        const size_t lds = (size_t)maxo * sizeof(int);
#define LNR_EB(F)                                                                                                             \
        do {                                                                                                                  \
            if (d_pts) hipLaunchKernelGGL((encode_backward_kernel<F, true>), grid, block, lds, st, *spec, table, *src, dfeat, dxl, m_pad, bpg, sink); \
            else hipLaunchKernelGGL((encode_backward_kernel<F, false>), grid, block, lds, st, *spec, table, *src, dfeat, dxl, m_pad, bpg, sink);     \
        } while (0)
        switch (spec->n_features) {
            case 1: LNR_EB(1); break;
            case 2: LNR_EB(2); break;
            case 4: LNR_EB(4); break;
            default: LNR_EB(8); break;
        }
#undef LNR_EB
    } else if (d_pts) {
        int64_t blocks = (cap_points + ENC_BLOCK - 1) / ENC_BLOCK;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(freq_backward_kernel, dim3((unsigned)blocks), dim3(ENC_BLOCK), 0, st, *spec, *src, dfeat, dxl, m_pad);
    }
    if (d_pts) {
        int64_t blocks = (cap_points + ENC_BLOCK - 1) / ENC_BLOCK;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(sum_dx_planes_kernel, dim3((unsigned)blocks), dim3(ENC_BLOCK), 0, st, dxl, n_groups, m_pad, *src, d_pts);
    }
    return LNR_OK;
}
