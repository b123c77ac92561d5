// Device code of the density network kernels; included by lnr_density_ht.hip (one translation unit per
// hidden width, -DLNR_HT=<n_neurons/16>) so that the five widths compile in parallel.
#pragma once
// Density network sigma = MLP(enc((xyz+1)/2)) forward / backward on MFMA (gfx950).
//
// Replaces the tinycudann NetworkWithInputEncoding the reference calls at
// src/models/nerf_tcnn.py:63-72 (forward) and through loss.backward()
// (src/mapping/optimizer.py:366).  Semantics = oracle/network.py.
//
// Kernel design (CDNA4, 64-wide waves, v_mfma_f32_16x16x4_f32 = exact fp32 fma chains):
//   * "transposed" MLP: a wave owns a tile of 16 samples as the N (column) dimension, neurons
//     are the M (row) dimension:  Z^T[j][c] = sum_k W[j][k] * X^T[k][c].
//     lane = (c = lane&15 -> sample, g = lane>>4 -> k-slot).
//   * the MFMA result layout (row = 4g+r, col = c) is exactly the B-operand layout of the next
//     layer if k-slot g of step (jt,r) is defined to be neuron 16jt+4g+r, so activations chain
//     through registers with no shuffles, no LDS.
//   * the same trick assigns input features: lane (c,g) produces features 16kt+4g+r (r=0..3) of
//     sample c, i.e. each lane interpolates 1/4 of the levels of one sample - every
//     (sample, level) pair is gathered exactly once per pass.
//   * backward: dX^T = W^T dZ^T needs no data movement either (dZ is already a B operand and the
//     result lands on the lane that owns those features, which then scatters into the table);
//     only the weight gradient contracts over samples (= lanes) and goes through a 16x16 LDS
//     transpose per tile; partial weight gradients are summed in LDS per block and written to a
//     per-block slab that a second kernel reduces (deterministic, no global atomics on weights).
//   * hash-table gradients use global float atomics (L2), skipped for samples whose upstream
//     gradient is exactly zero (ReLU-dead or fully occluded samples).
#include "lnr_common.h"
#include "lnr_density_api.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PRIME_Y 2654435761u
#define PRIME_Z 805459861u
#define LNR_PI_F 3.14159265358979323846f
#define LNR_PI_2_F 1.57079632679489661923f


__device__ __forceinline__ int64_t live_points(const PointSrc& s) {
    if (s.pts) return s.n_points;
    return (int64_t)lnr_live_rays(s.n_rays, s.n_rays_dev) * s.n_samples;
}

// unit-cube coordinates of point m
__device__ __forceinline__ void load_unit_point(const PointSrc& s, int64_t m, float x[3]) {
    float p[3];
    if (s.pts) {
        p[0] = s.pts[3 * m + 0]; p[1] = s.pts[3 * m + 1]; p[2] = s.pts[3 * m + 2];
    } else {
        int64_t ray = m / s.n_samples;
        float zv = s.z[m];
        const float* r = s.rays + ray * LNR_RAY_STRIDE;
#pragma unroll
        for (int d = 0; d < 3; ++d) p[d] = lnr_add_rn(r[d], lnr_mul_rn(r[3 + d], zv));   // o + d*z, as the reference rounds it
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = lnr_mul_rn(lnr_add_rn(p[d], 1.0f), 0.5f);   // (xyz+1)/2 rounded like the reference (no fma)
}

__device__ __forceinline__ float act_fwd(float v, int kind) {
    switch (kind) {
        case LNR_ACT_RELU: return fmaxf(v, 0.0f);
        case LNR_ACT_SINE: return sinf(v);
        case LNR_ACT_LEAKY_RELU: return v > 0.0f ? v : 0.01f * v;
        case LNR_ACT_EXPONENTIAL: return expf(v);
        case LNR_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case LNR_ACT_SQUAREPLUS: return 0.5f * (v + sqrtf(v * v + 4.0f));
        case LNR_ACT_SOFTPLUS: return v > 20.0f ? v : log1pf(expf(v));
        case LNR_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
// derivative of the activation with respect to its pre-activation v
__device__ __forceinline__ float act_bwd(float v, int kind) {
    switch (kind) {
        case LNR_ACT_RELU: return v > 0.0f ? 1.0f : 0.0f;
        case LNR_ACT_SINE: return cosf(v);
        case LNR_ACT_LEAKY_RELU: return v > 0.0f ? 1.0f : 0.01f;
        case LNR_ACT_EXPONENTIAL: return expf(v);
        case LNR_ACT_SIGMOID: { float s = 1.0f / (1.0f + expf(-v)); return s * (1.0f - s); }
        case LNR_ACT_SQUAREPLUS: return 0.5f * (1.0f + v / sqrtf(v * v + 4.0f));
        case LNR_ACT_SOFTPLUS: return 1.0f / (1.0f + expf(-v));
        case LNR_ACT_TANH: { float t = tanhf(v); return 1.0f - t * t; }
        default: return 1.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// multiresolution hash grid: one level, one point
// ------------------------------------------------------------------------------------------------
struct LevelCell {
    uint32_t base[3];   // integer cell
    float frac[3];
    float scale;
    uint32_t res, size, offset, hashed;
};

// Per-level geometry lives in LDS (5 x 32 words at the start of the dynamic LDS block): the level a
// lane works on depends on its lane group, and indexing kernel arguments by a VGPR would force the
// whole spec struct into scratch memory.
#define LNR_LV_WORDS (5 * LNR_MAX_LEVELS)
__device__ __forceinline__ void stage_level_tables(const LnrNetSpec& spec, float* lds) {
    uint32_t* u = reinterpret_cast<uint32_t*>(lds);
    for (int i = threadIdx.x; i < LNR_MAX_LEVELS; i += blockDim.x) {
        lds[i] = spec.level_scale[i];
        u[LNR_MAX_LEVELS + i] = spec.level_res[i];
        u[2 * LNR_MAX_LEVELS + i] = spec.level_size[i];
        u[3 * LNR_MAX_LEVELS + i] = spec.level_offset[i];
        const uint32_t sz = spec.level_size[i];
        u[4 * LNR_MAX_LEVELS + i] = (spec.level_hashed[i] & 1u) | ((sz != 0u && (sz & (sz - 1u)) == 0u) ? 2u : 0u);   // bit0 hashed, bit1 size is 2^k
    }
}

__device__ __forceinline__ LevelCell level_cell(const float* lvt, int lv, const float x[3]) {
    LevelCell c;
    const uint32_t* u = reinterpret_cast<const uint32_t*>(lvt);
    c.scale = lvt[lv];
    c.res = u[LNR_MAX_LEVELS + lv];
    c.size = u[2 * LNR_MAX_LEVELS + lv];
    c.offset = u[3 * LNR_MAX_LEVELS + lv];
    c.hashed = u[4 * LNR_MAX_LEVELS + lv];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float pos = lnr_add_rn(lnr_mul_rn(x[d], c.scale), 0.5f);
        float fl = floorf(pos);
        c.frac[d] = pos - fl;
        c.base[d] = (uint32_t)(int32_t)fl;
    }
    return c;
}

// kept out of line so that the common power-of-two case really skips the division sequence
__device__ __noinline__ uint32_t lnr_slow_mod(uint32_t a, uint32_t b) { return a % b; }

__device__ __forceinline__ uint32_t cell_entry(const LevelCell& c, int corner) {
    uint32_t cx = c.base[0] + (corner & 1), cy = c.base[1] + ((corner >> 1) & 1), cz = c.base[2] + ((corner >> 2) & 1);
    uint32_t idx = (c.hashed & 1u) ? (cx ^ (cy * PRIME_Y) ^ (cz * PRIME_Z)) : (cx + cy * c.res + cz * c.res * c.res);
    // table sizes are powers of two for every level of the usual configurations: mask instead of a ~40-instruction modulo
    if (c.hashed & 2u) idx &= (c.size - 1u); else idx = lnr_slow_mod(idx, c.size);
    return c.offset + idx;
}

__device__ __forceinline__ float corner_weight(const LevelCell& c, int corner) {
    float wx = (corner & 1) ? c.frac[0] : 1.0f - c.frac[0];
    float wy = (corner & 2) ? c.frac[1] : 1.0f - c.frac[1];
    float wz = (corner & 4) ? c.frac[2] : 1.0f - c.frac[2];
    return wx * wy * wz;
}

// The 4 input features k0..k0+3 of one point.  F = features per level.
//   F=1: 4 levels x 1 feature   F=2: 2 levels x 2   F=4: 1 level x 4   F=8: half a level
template <int F>
__device__ __forceinline__ void hash_features4(const LnrNetSpec& spec, const float* lvt, const float* __restrict__ table,
                                               const float x[3], int k0, float out[4]) {
    constexpr int NLV = F >= 4 ? 1 : 4 / F;
    constexpr int FPL = F >= 4 ? 4 : F;
#pragma unroll
    for (int li = 0; li < NLV; ++li) {
        const int lv = k0 / F + li;
        const int f0 = (F == 8) ? (k0 & 7) : 0;
        float acc[FPL];
#pragma unroll
        for (int f = 0; f < FPL; ++f) acc[f] = 0.0f;
        if (lv < spec.n_levels) {
            LevelCell c = level_cell(lvt, lv, x);
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const float w = corner_weight(c, corner);
                const float* e = table + (size_t)cell_entry(c, corner) * F + f0;
                if constexpr (FPL == 1) {
                    acc[0] += w * e[0];
                } else if constexpr (FPL == 2) {
                    float2 v = *reinterpret_cast<const float2*>(e);
                    acc[0] += w * v.x; acc[1] += w * v.y;
                } else {
                    float4 v = *reinterpret_cast<const float4*>(e);
                    acc[0] += w * v.x; acc[1] += w * v.y; acc[2] += w * v.z; acc[3] += w * v.w;
                }
            }
        } else {
#pragma unroll
            for (int f = 0; f < FPL; ++f) acc[f] = 1.0f;   // padding inputs are the constant 1
        }
#pragma unroll
        for (int f = 0; f < FPL; ++f) out[li * FPL + f] = acc[f];
    }
}

// ------------------------------------------------------------------------------------------------
// Table-gradient sink.  Float atomics to global memory top out at ~21 G atomics/s on MI355X whatever
// the scope, locality or table size (profiles/r01_atomic_throughput.txt) - 25 ms for the 537 M updates
// of one default iteration.  Instead every workgroup appends {float index, value} records to
// per-(workgroup, owner) regions in HBM (slot allocation = LDS atomic on a per-owner cursor) and a second
// kernel gives each slice of the table to one workgroup that sums its records in LDS
// (table_grad_reduce_kernel).  A record that does not fit its region falls back to a global atomic.
// ------------------------------------------------------------------------------------------------
struct GradSink {
    float* grad_table;     // fallback target
    void* regions;         // [n_workgroups][nown][cap] records: uint2 {idx, v} (pair=0) or uint4 {idx, v0, v1, -} (pair=1)
    int* cursors;          // LDS, [nown]
    int nown, cap, shift;
    float combine_scale_max;   // levels coarser than this are run-length combined over the 16 samples of a tile
    int debug;                 // LNR_DEBUG bits (profiling experiments only): 1 skip record store, 2 skip cursor atomic
};

// one float
__device__ __forceinline__ void sink_emit1(const GradSink& s, uint32_t fidx, float v) {
    if (v == 0.0f) return;
    const int owner = (int)(fidx >> s.shift);
    const int slot = atomicAdd(&s.cursors[owner], 1);
    if (slot < s.cap) reinterpret_cast<uint2*>(s.regions)[((size_t)blockIdx.x * s.nown + owner) * s.cap + slot] = make_uint2(fidx, __float_as_uint(v));
    else atomicAdd(s.grad_table + fidx, v);
}
// two consecutive floats (fidx even: both belong to the same owner slice); one 16-byte store = one line transaction
__device__ __forceinline__ void sink_emit2(const GradSink& s, uint32_t fidx, float v0, float v1) {
    if (v0 == 0.0f && v1 == 0.0f) return;
    const int owner = (int)(fidx >> s.shift);
    if (s.debug & 2) { if (v0 == 1e30f) s.cursors[owner] = 1; return; }
    const int slot = atomicAdd(&s.cursors[owner], 1);
    if (s.debug & 1) return;
    if (slot < s.cap)
        reinterpret_cast<uint4*>(s.regions)[((size_t)blockIdx.x * s.nown + owner) * s.cap + slot] =
            make_uint4(fidx, __float_as_uint(v0), __float_as_uint(v1), 0u);
    else { atomicAdd(s.grad_table + fidx, v0); atomicAdd(s.grad_table + fidx + 1, v1); }
}
template <int FPL>
__device__ __forceinline__ void sink_emit(const GradSink& s, uint32_t fidx, const float v[FPL]) {
    if constexpr (FPL == 1) sink_emit1(s, fidx, v[0]);
    else if constexpr (FPL == 2) sink_emit2(s, fidx, v[0], v[1]);
    else { sink_emit2(s, fidx, v[0], v[1]); sink_emit2(s, fidx + 2, v[2], v[3]); }
}

// Backward of hash_features4 for the 16 lanes of one lane group (= 16 consecutive samples, normally of one
// ray): emit table-gradient records, optionally accumulate d/dx.  Must be called by whole lane groups
// (row-uniform control flow): coarse levels are combined across the row with segmented shuffles because
// neighbouring samples of a ray fall into the same cell there.
template <int F, bool WANT_DX>
__device__ __forceinline__ void hash_features4_bwd(const LnrNetSpec& spec, const float* lvt, const float* __restrict__ table,
                                                   const GradSink& sink, const float x[3], int k0, int lane,
                                                   const float d_out[4], float dx[3]) {
    constexpr int NLV = F >= 4 ? 1 : 4 / F;
    constexpr int FPL = F >= 4 ? 4 : F;
    const int c16 = lane & 15, row = lane >> 4;
#pragma unroll
    for (int li = 0; li < NLV; ++li) {
        const int lv = k0 / F + li;                       // row-uniform
        const int f0 = (F == 8) ? (k0 & 7) : 0;
        if (lv >= spec.n_levels) continue;
        float g[FPL];
        bool any = false;
#pragma unroll
        for (int f = 0; f < FPL; ++f) { g[f] = d_out[li * FPL + f]; any |= (g[f] != 0.0f); }
        const unsigned row_any = (unsigned)((__ballot(any) >> (16 * row)) & 0xFFFFull);
        if (row_any == 0u) continue;                      // row-uniform
        LevelCell c = level_cell(lvt, lv, x);
        const bool combine = c.scale < sink.combine_scale_max;     // row-uniform (depends on the level only)
        float dfrac[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const float w = corner_weight(c, corner);
            const uint32_t entry = cell_entry(c, corner);
            const uint32_t e = entry * F + f0;
            float v[FPL];
#pragma unroll
            for (int f = 0; f < FPL; ++f) v[f] = w * g[f];
            if (combine) {
                // segmented sum over runs of equal entry along the row (DPP row shifts)
                const uint32_t prev = (uint32_t)row_up_i<1>((int)entry);
                const bool head = (c16 == 0) || (prev != entry);
                int seg = head ? 1 : 0;
                { int t;
                  t = row_up_i<1>(seg); if (c16 >= 1) seg += t;
                  t = row_up_i<2>(seg); if (c16 >= 2) seg += t;
                  t = row_up_i<4>(seg); if (c16 >= 4) seg += t;
                  t = row_up_i<8>(seg); if (c16 >= 8) seg += t; }
#define LNR_SEG_STEP(O)                                                                          \
                { const int s2 = row_down_i<O>(seg);                                             \
                  const bool take = (c16 + O < 16) && (s2 == seg);                               \
                  _Pragma("unroll") for (int f = 0; f < FPL; ++f) { const float t = row_down_f<O>(v[f]); if (take) v[f] += t; } }
                LNR_SEG_STEP(1) LNR_SEG_STEP(2) LNR_SEG_STEP(4) LNR_SEG_STEP(8)
#undef LNR_SEG_STEP
                if (head) sink_emit<FPL>(sink, e, v);
            } else {
                sink_emit<FPL>(sink, e, v);
            }
            if constexpr (WANT_DX) {
                if (any) {
                    float dot = 0.0f;
#pragma unroll
                    for (int f = 0; f < FPL; ++f) dot += g[f] * table[(size_t)e + f];
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
            for (int d = 0; d < 3; ++d) dx[d] += dfrac[d] * c.scale;
        }
    }
}

// Frequency encoding: feature k = sin(x[dim]*2^freq*pi + (k&1)*pi/2), order [dim][freq][sin,cos]
__device__ __forceinline__ float freq_phase(const LnrNetSpec& spec, const float x[3], int k, float* dphase_dx, int* dim_out) {
    const int per_dim = 2 * spec.n_frequencies;
    const int dim = k / per_dim;
    const int rem = k - dim * per_dim;
    const float mult = (float)(1u << (rem >> 1));     // exact power of two (exp2f is not exact on the GPU)
    float xv = dim == 0 ? x[0] : (dim == 1 ? x[1] : x[2]);
    float ph = lnr_mul_rn(lnr_mul_rn(xv, mult), LNR_PI_F);
    if (rem & 1) ph = lnr_add_rn(ph, LNR_PI_2_F);
    *dphase_dx = mult * LNR_PI_F;
    *dim_out = dim;
    return ph;
}

__device__ __forceinline__ void freq_features4(const LnrNetSpec& spec, const float x[3], int k0, float out[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + r;
        if (k < spec.enc_dim) {
            float d; int dim;
            out[r] = sinf(freq_phase(spec, x, k, &d, &dim));
        } else {
            out[r] = 1.0f;
        }
    }
}

__device__ __forceinline__ void freq_features4_bwd(const LnrNetSpec& spec, const float x[3], int k0,
                                                   const float d_out[4], float dx[3]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + r;
        if (k < spec.enc_dim && d_out[r] != 0.0f) {
            float d; int dim;
            float ph = freq_phase(spec, x, k, &d, &dim);
            float v = d_out[r] * cosf(ph) * d;
            if (dim == 0) dx[0] += v; else if (dim == 1) dx[1] += v; else dx[2] += v;
        }
    }
}

__device__ __forceinline__ void features4(const LnrNetSpec& spec, const float* lvt, const float* __restrict__ table,
                                          const float x[3], int k0, float out[4]) {
    if (spec.encoding == LNR_ENC_FREQUENCY) { freq_features4(spec, x, k0, out); return; }
    switch (spec.n_features) {
        case 1: hash_features4<1>(spec, lvt, table, x, k0, out); break;
        case 2: hash_features4<2>(spec, lvt, table, x, k0, out); break;
        case 4: hash_features4<4>(spec, lvt, table, x, k0, out); break;
        default: hash_features4<8>(spec, lvt, table, x, k0, out); break;
    }
}

template <bool WANT_DX>
__device__ __forceinline__ void features4_bwd(const LnrNetSpec& spec, const float* lvt, const float* __restrict__ table,
                                              const GradSink& sink, const float x[3], int k0, int lane,
                                              const float d_out[4], float dx[3]) {
    if (spec.encoding == LNR_ENC_FREQUENCY) {
        if constexpr (WANT_DX) freq_features4_bwd(spec, x, k0, d_out, dx);
        return;
    }
    switch (spec.n_features) {
        case 1: hash_features4_bwd<1, WANT_DX>(spec, lvt, table, sink, x, k0, lane, d_out, dx); break;
        case 2: hash_features4_bwd<2, WANT_DX>(spec, lvt, table, sink, x, k0, lane, d_out, dx); break;
        case 4: hash_features4_bwd<4, WANT_DX>(spec, lvt, table, sink, x, k0, lane, d_out, dx); break;
        default: hash_features4_bwd<8, WANT_DX>(spec, lvt, table, sink, x, k0, lane, d_out, dx); break;
    }
}

#define MFMA4(acc, a4, b0, b1, b2, b3)                                      \
    do {                                                                    \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).x, b0, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).y, b1, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).z, b2, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).w, b3, acc, 0, 0, 0); \
    } while (0)

// layer 1 for one tile: Z[jt] (+)= W1 * X, features generated on the fly.
// If xt != nullptr the features are also stored transposed for the weight-gradient GEMM: xt[k*16 + c].
template <int HT>
__device__ __forceinline__ void layer1_forward(const LnrNetSpec& spec, const float* lvt, const float* W1, const float* __restrict__ table,
                                               const float x[3], int c, int g, f32x4 Z[HT], float* xt) {
    const int in_dim = spec.in_dim;
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) Z[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int kt = 0; kt < in_dim / 16; ++kt) {
        const int k0 = 16 * kt + 4 * g;
        float xf[4];
        features4(spec, lvt, table, x, k0, xf);
        if (xt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) xt[(k0 + r) * 16 + c] = xf[r];
        }
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) {
            const float4 wa = *reinterpret_cast<const float4*>(W1 + (16 * jt + c) * in_dim + k0);
            MFMA4(Z[jt], wa, xf[0], xf[1], xf[2], xf[3]);
        }
    }
}

// hidden layer: Zn = Wl * act(Z)
template <int HT>
__device__ __forceinline__ void hidden_forward(const float* Wl, int H, int act, int c, int g, const f32x4 Z[HT], f32x4 Zn[HT]) {
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) Zn[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kt = 0; kt < HT; ++kt) {
        const float a0 = act_fwd(Z[kt].x, act), a1 = act_fwd(Z[kt].y, act), a2 = act_fwd(Z[kt].z, act), a3 = act_fwd(Z[kt].w, act);
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) {
            const float4 wa = *reinterpret_cast<const float4*>(Wl + (16 * jt + c) * H + 16 * kt + 4 * g);
            MFMA4(Zn[jt], wa, a0, a1, a2, a3);
        }
    }
}

// W_LDS: MLP matrices staged in LDS (small networks) or read straight from global memory / L2.
template <int HT, bool W_LDS>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK)
density_forward_kernel(const LnrNetSpec spec, const float* __restrict__ params, const PointSrc src, float* __restrict__ sigma) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const float* lvt = smem_all;
    float* smem = smem_all + LNR_LV_WORDS;
    const int H = 16 * HT;
    const int n_mlp = spec.n_mlp_params;
    const int nw = blockDim.x >> 6;
    stage_level_tables(spec, smem_all);
    if (W_LDS) for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) smem[i] = params[i];
    __syncthreads();
    const float* W1 = W_LDS ? smem : params;
    const float* Wh = smem + H * spec.in_dim;
    const float* Wo = Wh + (spec.n_hidden - 1) * H * H;
    const float* table = params + n_mlp;
    const int act = spec.activation;

    const int64_t M = live_points(src);
    if (M <= 0) return;
    const int64_t n_tiles = (M + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < n_tiles; tile += (int64_t)gridDim.x * nw) {
        int64_t m = tile * 16 + c;
        const bool valid = m < M;
        if (!valid) m = M - 1;
        float x[3];
        load_unit_point(src, m, x);
        f32x4 Z[HT];
        layer1_forward<HT>(spec, lvt, W1, table, x, c, g, Z, nullptr);
        for (int l = 1; l < spec.n_hidden; ++l) {
            f32x4 Zn[HT];
            hidden_forward<HT>(Wh + (l - 1) * H * H, H, act, c, g, Z, Zn);
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) Z[jt] = Zn[jt];
        }
        float part = 0.0f;
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) {
            const float4 wo = *reinterpret_cast<const float4*>(Wo + 16 * jt + 4 * g);
            part += wo.x * act_fwd(Z[jt].x, act) + wo.y * act_fwd(Z[jt].y, act) + wo.z * act_fwd(Z[jt].z, act) + wo.w * act_fwd(Z[jt].w, act);
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (g == 0 && valid) sigma[m] = part;
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// LDS map (floats): [level tables][W : n_mlp if W_LDS][dW : n_mlp][per-wave scratch x nw]
//   scratch = T_dz [H*16] | T_x [in_dim*16] | if n_hidden>1: T_a [H*16] | zsave [n_hidden*H*16]
// DWK > 0: single-hidden-layer networks with in_dim == 16*DWK keep the layer-1 weight gradient in MFMA
// accumulators for the whole kernel (no per-tile LDS float atomics, which run at < 1 lane/clk/CU on CDNA4);
// DWK == 0: general path, per-tile accumulation into the workgroup's LDS copy.
#ifndef LNR_BWD_WAVES_PER_SIMD
#define LNR_BWD_WAVES_PER_SIMD (LNR_HT <= 4 ? 2 : 1)
#endif
template <int HT, bool WANT_DX, bool W_LDS, int DWK>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK, LNR_BWD_WAVES_PER_SIMD)
density_backward_kernel(const LnrNetSpec spec, const float* __restrict__ params, const PointSrc src,
                        const float* __restrict__ d_sigma, float* __restrict__ grad_table,
                        float* __restrict__ d_pts, float* __restrict__ slabs, const BwdSinkArgs sa) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const float* lvt = smem_all;
    float* smem = smem_all + LNR_LV_WORDS;
    stage_level_tables(spec, smem_all);
    const int H = 16 * HT;
    const int NH = spec.n_hidden;
    const int in_dim = spec.in_dim;
    const int n_mlp = spec.n_mlp_params;
    const int nw = blockDim.x >> 6;
    int* cursors = reinterpret_cast<int*>(smem);               // [sa.nown] record cursors
    smem += sa.nown_padded;
    float* dW = smem + (W_LDS ? n_mlp : 0);
    for (int i = threadIdx.x; i < sa.nown; i += blockDim.x) cursors[i] = 0;
    for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) { if (W_LDS) smem[i] = params[i]; dW[i] = 0.0f; }
    __syncthreads();
    GradSink sink;
    sink.grad_table = grad_table; sink.regions = sa.regions; sink.cursors = cursors;
    sink.nown = sa.nown; sink.cap = sa.cap; sink.shift = sa.shift; sink.combine_scale_max = sa.combine_scale_max; sink.debug = sa.debug;
    const float* W = W_LDS ? smem : params;
    const float* W1 = W;
    const float* Wh = W + H * in_dim;
    const float* Wo = Wh + (NH - 1) * H * H;
    float* dW1 = dW;
    float* dWh = dW + H * in_dim;
    float* dWo = dWh + (NH - 1) * H * H;
    const float* table = params + n_mlp;
    const int act = spec.activation;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int scratch_per_wave = H * 16 + in_dim * 16 + (NH > 1 ? (NH + 1) * H * 16 : 0);
    float* T_dz = dW + n_mlp + wave * scratch_per_wave;
    float* T_x = T_dz + H * 16;
    float* T_a = T_x + in_dim * 16;
    float* zsave = T_a + H * 16;

    f32x4 dWo_acc[HT];
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) dWo_acc[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int DWK_N = DWK > 0 ? DWK : 1;
    f32x4 dW1_acc[HT][DWK_N];
#pragma unroll
    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
        for (int kt = 0; kt < DWK_N; ++kt) dW1_acc[jt][kt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int64_t M = live_points(src);
    const int64_t n_tiles = M > 0 ? (M + 15) / 16 : 0;
    for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < n_tiles; tile += (int64_t)gridDim.x * nw) {
        int64_t m = tile * 16 + c;
        const bool valid = m < M;
        if (!valid) m = M - 1;
        const float ds = valid ? d_sigma[m] : 0.0f;
        if (__ballot(ds != 0.0f) == 0ull) {          // nothing flows back into this tile
            if (WANT_DX && g == 0 && valid) { d_pts[3 * m] = 0.0f; d_pts[3 * m + 1] = 0.0f; d_pts[3 * m + 2] = 0.0f; }
            continue;
        }
        float x[3];
        load_unit_point(src, m, x);

        // ---- recompute the forward pass, keeping what backward needs ----------------------------
        f32x4 Z[HT];
        layer1_forward<HT>(spec, lvt, W1, table, x, c, g, Z, T_x);     // T_x = X^T
        if (NH > 1) {
#pragma unroll
            for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) zsave[((0 * HT + jt) * 4 + r) * 64 + lane] = Z[jt][r];
            for (int l = 1; l < NH; ++l) {
                f32x4 Zn[HT];
                hidden_forward<HT>(Wh + (l - 1) * H * H, H, act, c, g, Z, Zn);
#pragma unroll
                for (int jt = 0; jt < HT; ++jt) {
                    Z[jt] = Zn[jt];
#pragma unroll
                    for (int r = 0; r < 4; ++r) zsave[((l * HT + jt) * 4 + r) * 64 + lane] = Zn[jt][r];
                }
            }
        }
        // Z now holds the pre-activations of the LAST hidden layer.

        // ---- output layer ----------------------------------------------------------------------------
        f32x4 dA[HT];     // gradient w.r.t. the activations of the current layer
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) {
            const float4 wo = *reinterpret_cast<const float4*>(Wo + 16 * jt + 4 * g);
            dA[jt] = f32x4{ds * wo.x, ds * wo.y, ds * wo.z, ds * wo.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) dWo_acc[jt][r] += ds * act_fwd(Z[jt][r], act);
        }

        // ---- hidden layers, last to first ---------------------------------------------------------------
        for (int l = NH - 1; l >= 0; --l) {
            f32x4 dZ[HT];
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float zv = (NH > 1) ? zsave[((l * HT + jt) * 4 + r) * 64 + lane] : Z[jt][r];
                    dZ[jt][r] = dA[jt][r] * act_bwd(zv, act);
                    T_dz[(16 * jt + 4 * g + r) * 16 + c] = dZ[jt][r];
                }
            }
            if (l > 0) {   // inputs of this layer = activations of layer l-1, transposed into T_a
#pragma unroll
                for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        T_a[(16 * jt + 4 * g + r) * 16 + c] = act_fwd(zsave[(((l - 1) * HT + jt) * 4 + r) * 64 + lane], act);
            }
            const float* T_p = (l == 0) ? T_x : T_a;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // weight gradient  dW_l[j][k] += sum_c dZ[j][c] * P[k][c]
            const int K = (l == 0) ? in_dim : H;
            float* dWl = (l == 0) ? dW1 : dWh + (l - 1) * H * H;
            if constexpr (DWK > 0) {           // n_hidden == 1, in_dim == 16*DWK: accumulate in registers
                if (!(sa.debug & 16))
#pragma unroll
                for (int kt = 0; kt < DWK; ++kt) {
                    const float4 b4 = *reinterpret_cast<const float4*>(T_p + (16 * kt + c) * 16 + 4 * g);
#pragma unroll
                    for (int jt = 0; jt < HT; ++jt) {
                        const float4 a4 = *reinterpret_cast<const float4*>(T_dz + (16 * jt + c) * 16 + 4 * g);
                        MFMA4(dW1_acc[jt][kt], a4, b4.x, b4.y, b4.z, b4.w);
                    }
                }
            } else
            for (int kt = 0; kt < K / 16; ++kt) {
                const float4 b4 = *reinterpret_cast<const float4*>(T_p + (16 * kt + c) * 16 + 4 * g);
#pragma unroll
                for (int jt = 0; jt < HT; ++jt) {
                    const float4 a4 = *reinterpret_cast<const float4*>(T_dz + (16 * jt + c) * 16 + 4 * g);
                    f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    MFMA4(acc, a4, b4.x, b4.y, b4.z, b4.w);
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(dWl + (16 * jt + 4 * g + r) * K + 16 * kt + c, acc[r]);
                }
            }

            // input gradient  dP[k][c] = sum_j W_l[j][k] * dZ[j][c]
            if (l > 0) {
                const float* Wl = Wh + (l - 1) * H * H;
#pragma unroll
                for (int kt = 0; kt < HT; ++kt) {
                    f32x4 D = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            D = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[(16 * jt + 4 * g + r) * H + 16 * kt + c], dZ[jt][r], D, 0, 0, 0);
                    dA[kt] = D;
                }
            } else {
                const bool need = WANT_DX || spec.encoding == LNR_ENC_HASHGRID;
                float dx[3] = {0.0f, 0.0f, 0.0f};
                if (need) {
                    for (int kt = 0; kt < in_dim / 16; ++kt) {
                        f32x4 D = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                D = __builtin_amdgcn_mfma_f32_16x16x4f32(W1[(16 * jt + 4 * g + r) * in_dim + 16 * kt + c], dZ[jt][r], D, 0, 0, 0);
                        if (!(sa.debug & 8)) {   // all 16 lanes of the row take part (segmented shuffles); dead samples carry zeros
                            const float dfeat[4] = {D.x, D.y, D.z, D.w};
                            features4_bwd<WANT_DX>(spec, lvt, table, sink, x, 16 * kt + 4 * g, lane, dfeat, dx);
                        }
                    }
                }
                if constexpr (WANT_DX) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        dx[d] += __shfl_xor(dx[d], 16, 64);
                        dx[d] += __shfl_xor(dx[d], 32, 64);
                    }
                    if (g == 0 && valid) {      // x = (xyz+1)/2
                        d_pts[3 * m + 0] = 0.5f * dx[0];
                        d_pts[3 * m + 1] = 0.5f * dx[1];
                        d_pts[3 * m + 2] = 0.5f * dx[2];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }

    if constexpr (DWK > 0) {       // flush the register-resident layer-1 weight gradient once per wave
#pragma unroll
        for (int jt = 0; jt < HT; ++jt)
#pragma unroll
            for (int kt = 0; kt < DWK; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(dW1 + (16 * jt + 4 * g + r) * in_dim + 16 * kt + c, dW1_acc[jt][kt][r]);
    }
    // output-layer weight gradient: reduce the per-lane partial sums over the 16 sample lanes
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = dWo_acc[jt][r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (c == 0) atomicAdd(dWo + 16 * jt + 4 * g + r, v);
        }
    }
    __syncthreads();
    float* slab = slabs + (size_t)blockIdx.x * n_mlp;
    for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) slab[i] = dW[i];
    for (int i = threadIdx.x; i < sa.nown; i += blockDim.x)
        sa.counts[(size_t)blockIdx.x * sa.nown + i] = cursors[i] < sa.cap ? cursors[i] : sa.cap;
}



// ================================================================================================
// MLP kernels on feature planes (level-major pipeline: lnr_encode.hip produces / consumes the planes)
//   feat [enc_dim][m_pad]   dfeat [enc_dim][m_pad]     padded inputs (k >= enc_dim) are the constant 1
// ================================================================================================
template <int HT>
__device__ __forceinline__ void layer1_from_planes(const LnrNetSpec& spec, const float* W1, const float* __restrict__ feat,
                                                   int64_t m_pad, int64_t m, int c, int g, f32x4 Z[HT]) {
    const int in_dim = spec.in_dim;
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) Z[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int kt = 0; kt < in_dim / 16; ++kt) {
        const int k0 = 16 * kt + 4 * g;
        float xf[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xf[r] = (k0 + r < spec.enc_dim) ? feat[(size_t)(k0 + r) * m_pad + m] : 1.0f;
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) {
            const float4 wa = *reinterpret_cast<const float4*>(W1 + (16 * jt + c) * in_dim + k0);
            MFMA4(Z[jt], wa, xf[0], xf[1], xf[2], xf[3]);
        }
    }
}

template <int HT, bool W_LDS>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK)
mlp_forward_kernel(const LnrNetSpec spec, const float* __restrict__ params, const float* __restrict__ feat, int64_t m_pad,
                   int64_t n_points, const int32_t* __restrict__ n_rays_dev, int n_rays, int n_samples, float* __restrict__ sigma) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = 16 * HT;
    const int n_mlp = spec.n_mlp_params;
    const int nw = blockDim.x >> 6;
    if (W_LDS) for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) smem[i] = params[i];
    __syncthreads();
    const float* W1 = W_LDS ? smem : params;
    const float* Wh = W1 + H * spec.in_dim;
    const float* Wo = Wh + (spec.n_hidden - 1) * H * H;
    const int act = spec.activation;
    const int64_t M = n_rays_dev ? (int64_t)lnr_live_rays(n_rays, n_rays_dev) * n_samples : n_points;
    if (M <= 0) return;
    const int64_t n_tiles = (M + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < n_tiles; tile += (int64_t)gridDim.x * nw) {
        int64_t m = tile * 16 + c;
        const bool valid = m < M;
        if (!valid) m = M - 1;
        f32x4 Z[HT];
        layer1_from_planes<HT>(spec, W1, feat, m_pad, m, c, g, Z);
        for (int l = 1; l < spec.n_hidden; ++l) {
            f32x4 Zn[HT];
            hidden_forward<HT>(Wh + (l - 1) * H * H, H, act, c, g, Z, Zn);
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) Z[jt] = Zn[jt];
        }
        float part = 0.0f;
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) {
            const float4 wo = *reinterpret_cast<const float4*>(Wo + 16 * jt + 4 * g);
            part += wo.x * act_fwd(Z[jt].x, act) + wo.y * act_fwd(Z[jt].y, act) + wo.z * act_fwd(Z[jt].z, act) + wo.w * act_fwd(Z[jt].w, act);
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (g == 0 && valid) sigma[m] = part;
    }
}

// Backward of the MLP on feature planes: weight gradients (slabs) and dfeat planes.  No gathers, no scatters.
// LDS map (floats): [W if W_LDS][dW][per-wave: T_dz [H*16] | if n_hidden>1: T_a [H*16] | zsave [n_hidden*H*16]]
template <int HT, bool W_LDS, int DWK>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK)
mlp_backward_kernel(const LnrNetSpec spec, const float* __restrict__ params, const float* __restrict__ feat, int64_t m_pad,
                    int64_t n_points, const int32_t* __restrict__ n_rays_dev, int n_rays, int n_samples,
                    const float* __restrict__ d_sigma, float* __restrict__ dfeat, float* __restrict__ slabs, int want_dfeat) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = 16 * HT;
    const int NH = spec.n_hidden;
    const int in_dim = spec.in_dim;
    const int n_mlp = spec.n_mlp_params;
    const int nw = blockDim.x >> 6;
    float* dW = smem + (W_LDS ? n_mlp : 0);
    for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) { if (W_LDS) smem[i] = params[i]; dW[i] = 0.0f; }
    __syncthreads();
    const float* W = W_LDS ? smem : params;
    const float* W1 = W;
    const float* Wh = W + H * in_dim;
    const float* Wo = Wh + (NH - 1) * H * H;
    float* dW1 = dW;
    float* dWh = dW + H * in_dim;
    float* dWo = dWh + (NH - 1) * H * H;
    const int act = spec.activation;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int scratch_per_wave = H * 16 + (NH > 1 ? (NH + 1) * H * 16 : 0);
    float* T_dz = dW + n_mlp + wave * scratch_per_wave;
    float* T_a = T_dz + H * 16;
    float* zsave = T_a + H * 16;

    f32x4 dWo_acc[HT];
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) dWo_acc[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int DWK_N = DWK > 0 ? DWK : 1;
    f32x4 dW1_acc[HT][DWK_N];
#pragma unroll
    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
        for (int kt = 0; kt < DWK_N; ++kt) dW1_acc[jt][kt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int64_t M = n_rays_dev ? (int64_t)lnr_live_rays(n_rays, n_rays_dev) * n_samples : n_points;
    const int64_t n_tiles = M > 0 ? (M + 15) / 16 : 0;
    for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < n_tiles; tile += (int64_t)gridDim.x * nw) {
        int64_t m = tile * 16 + c;
        const bool valid = m < M;
        if (!valid) m = M - 1;
        const float ds = valid ? d_sigma[m] : 0.0f;
        if (__ballot(ds != 0.0f) == 0ull) {          // nothing flows back into this tile
            if (want_dfeat && valid) {
                for (int kt = 0; kt < in_dim / 16; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int k = 16 * kt + 4 * g + r; if (k < spec.enc_dim) dfeat[(size_t)k * m_pad + m] = 0.0f; }
            }
            continue;
        }
        f32x4 Z[HT];
        layer1_from_planes<HT>(spec, W1, feat, m_pad, m, c, g, Z);
        if (NH > 1) {
#pragma unroll
            for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) zsave[((0 * HT + jt) * 4 + r) * 64 + lane] = Z[jt][r];
            for (int l = 1; l < NH; ++l) {
                f32x4 Zn[HT];
                hidden_forward<HT>(Wh + (l - 1) * H * H, H, act, c, g, Z, Zn);
#pragma unroll
                for (int jt = 0; jt < HT; ++jt) {
                    Z[jt] = Zn[jt];
#pragma unroll
                    for (int r = 0; r < 4; ++r) zsave[((l * HT + jt) * 4 + r) * 64 + lane] = Zn[jt][r];
                }
            }
        }
        f32x4 dA[HT];
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) {
            const float4 wo = *reinterpret_cast<const float4*>(Wo + 16 * jt + 4 * g);
            dA[jt] = f32x4{ds * wo.x, ds * wo.y, ds * wo.z, ds * wo.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) dWo_acc[jt][r] += ds * act_fwd(Z[jt][r], act);
        }
        for (int l = NH - 1; l >= 0; --l) {
            f32x4 dZ[HT];
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float zv = (NH > 1) ? zsave[((l * HT + jt) * 4 + r) * 64 + lane] : Z[jt][r];
                    dZ[jt][r] = dA[jt][r] * act_bwd(zv, act);
                    T_dz[(16 * jt + 4 * g + r) * 16 + c] = dZ[jt][r];
                }
            }
            if (l > 0) {
#pragma unroll
                for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        T_a[(16 * jt + 4 * g + r) * 16 + c] = act_fwd(zsave[(((l - 1) * HT + jt) * 4 + r) * 64 + lane], act);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // weight gradient.  Layer 1: the B operand (X^T, 4 consecutive samples of one feature) is 16 contiguous
            // bytes of a feature plane - read straight from global memory, no LDS transpose.
            const int K = (l == 0) ? in_dim : H;
            float* dWl = (l == 0) ? dW1 : dWh + (l - 1) * H * H;
            const int64_t tile_base = tile * 16;
            if (l == 0) {
                auto load_b = [&](int kt) -> float4 {
                    const int k = 16 * kt + c;
                    if (k >= spec.enc_dim) return make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                    if (tile_base + 16 <= M) return *reinterpret_cast<const float4*>(feat + (size_t)k * m_pad + tile_base + 4 * g);
                    float4 b;   // ragged last tile: dZ of the padding samples is 0, any finite value will do
                    const float* p = feat + (size_t)k * m_pad;
                    b.x = p[min(tile_base + 4 * g + 0, M - 1)]; b.y = p[min(tile_base + 4 * g + 1, M - 1)];
                    b.z = p[min(tile_base + 4 * g + 2, M - 1)]; b.w = p[min(tile_base + 4 * g + 3, M - 1)];
                    return b;
                };
                if constexpr (DWK > 0) {
#pragma unroll
                    for (int kt = 0; kt < DWK; ++kt) {
                        const float4 b4 = load_b(kt);
#pragma unroll
                        for (int jt = 0; jt < HT; ++jt) {
                            const float4 a4 = *reinterpret_cast<const float4*>(T_dz + (16 * jt + c) * 16 + 4 * g);
                            MFMA4(dW1_acc[jt][kt], a4, b4.x, b4.y, b4.z, b4.w);
                        }
                    }
                } else {
                    for (int kt = 0; kt < K / 16; ++kt) {
                        const float4 b4 = load_b(kt);
#pragma unroll
                        for (int jt = 0; jt < HT; ++jt) {
                            const float4 a4 = *reinterpret_cast<const float4*>(T_dz + (16 * jt + c) * 16 + 4 * g);
                            f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                            MFMA4(acc, a4, b4.x, b4.y, b4.z, b4.w);
#pragma unroll
                            for (int r = 0; r < 4; ++r) atomicAdd(dWl + (16 * jt + 4 * g + r) * K + 16 * kt + c, acc[r]);
                        }
                    }
                }
            } else {
                for (int kt = 0; kt < K / 16; ++kt) {
                    const float4 b4 = *reinterpret_cast<const float4*>(T_a + (16 * kt + c) * 16 + 4 * g);
#pragma unroll
                    for (int jt = 0; jt < HT; ++jt) {
                        const float4 a4 = *reinterpret_cast<const float4*>(T_dz + (16 * jt + c) * 16 + 4 * g);
                        f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        MFMA4(acc, a4, b4.x, b4.y, b4.z, b4.w);
#pragma unroll
                        for (int r = 0; r < 4; ++r) atomicAdd(dWl + (16 * jt + 4 * g + r) * K + 16 * kt + c, acc[r]);
                    }
                }
            }
            // input gradient
            if (l > 0) {
                const float* Wl = Wh + (l - 1) * H * H;
#pragma unroll
                for (int kt = 0; kt < HT; ++kt) {
                    f32x4 D = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            D = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[(16 * jt + 4 * g + r) * H + 16 * kt + c], dZ[jt][r], D, 0, 0, 0);
                    dA[kt] = D;
                }
            } else if (want_dfeat) {
                for (int kt = 0; kt < in_dim / 16; ++kt) {
                    f32x4 D = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            D = __builtin_amdgcn_mfma_f32_16x16x4f32(W1[(16 * jt + 4 * g + r) * in_dim + 16 * kt + c], dZ[jt][r], D, 0, 0, 0);
                    if (valid) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const int k = 16 * kt + 4 * g + r; if (k < spec.enc_dim) dfeat[(size_t)k * m_pad + m] = D[r]; }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (DWK > 0) {
#pragma unroll
        for (int jt = 0; jt < HT; ++jt)
#pragma unroll
            for (int kt = 0; kt < DWK; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(dW1 + (16 * jt + 4 * g + r) * in_dim + 16 * kt + c, dW1_acc[jt][kt][r]);
    }
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = dWo_acc[jt][r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (c == 0) atomicAdd(dWo + 16 * jt + 4 * g + r, v);
        }
    }
    __syncthreads();
    float* slab = slabs + (size_t)blockIdx.x * n_mlp;
    for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) slab[i] = dW[i];
}
