// Density MLP in the reference's storage precision (LNR_PREC_F16): fp16 features and weights on
// v_mfma_f32_16x16x32_f16, fp32 accumulation (gfx950).
//
// The reference runs tinycudann's FullyFusedMLP in half precision (cfg/nerf_config/default_nerf_hash.yaml:20-31,
// src/models/nerf_tcnn.py:35-38: fp16 parameter copy, fp16 encoded features, fp16 tensor-core products with fp16
// accumulation, fp16 gradient atomics behind a loss scale of 128).  This mode keeps its storage types and replaces the
// weak parts: products accumulate in fp32, master parameters and all gradients stay fp32, and instead of a global loss
// scale every 32-sample tile is scaled by the exact power of two of its largest |d_sigma| before the fp16 conversion
// of dZ (undone in fp32 after the MFMA), so no gradient underflows whatever the loss magnitude.
// Semantics = oracle/network.py with precision="fp16".
//
// Shape class: 32 encoded features (16 levels x 2) -> 16*HT <= 64 ReLU neurons -> 1 (the reference's sigma network).
// Feature planes arrive as half2 pairs [level][sample] (lnr_encode.hip, F16 output): one dword load per lane yields two
// consecutive K entries of an MFMA B operand, no packing instructions.
//
// Tiling: a wave owns 32 samples per step = two 16-column tiles (the N dimension) for the products whose K is
// features or neurons, and ONE K = 32 block for the weight gradient, whose contraction runs over samples.
//   lane = (c = lane & 15, g = lane >> 4);  A operand: row c, K entries 8g..8g+7;  B: column c, K entries 8g..8g+7;
//   C/D: column c, rows 4g..4g+3  (checked on the device by lnr_selftest_mfma).
// As in the fp32 kernels the C layout of one product is made the B layout of the next by permuting K: K slot 8g+i of a
// 32-neuron block stands for neuron 4g+i of its first 16-neuron tile (i < 4) or 4g+i-4 of its second (i >= 4).
#include "lnr_f16_common.h"
#include "lnr_f16_freq.h"

#define F16_TS 40            // halves per neuron row of the dZ transpose buffer (32 samples + pad: 80-byte rows)

// layer-1 A fragments (W1 [H][32] row-major fp32 in `params`, rounded to fp16) and the output row, for the lane's neurons
template <int HT>
__device__ __forceinline__ void load_w1_frags(const float* __restrict__ params, int c, int g, f16x8 wa[HT], float wo[HT][4]) {
    constexpr int H = 16 * HT;
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) {
        const float* row = params + (16 * jt + c) * 32 + 8 * g;
#pragma unroll
        for (int i = 0; i < 8; ++i) wa[jt][i] = (f16)row[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) wo[jt][r] = round_f16(params[H * 32 + 16 * jt + 4 * g + r]);
    }
}

// B operand of layer 1 for column tile t of a 32-sample step: features 8g..8g+7 = half2 planes 4g..4g+3 of sample m
__device__ __forceinline__ u32x4 load_xb(const uint32_t* __restrict__ featp, uint32_t plane_bytes, uint32_t m, int g) {
    u32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = ld32<uint32_t>(featp, (uint32_t)(4 * g + q) * plane_bytes + m * 4u);
    return v;
}

template <int HT>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK, 2)
mlp_forward_f16_kernel(const float* __restrict__ params, const uint32_t* __restrict__ featp, int64_t m_pad, int64_t n_points,
                       const int32_t* __restrict__ n_rays_dev, int n_rays, int n_samples, float* __restrict__ sigma, int32_t* __restrict__ clip_flag) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    f16x8 wa[HT]; float wo[HT][4];
    load_w1_frags<HT>(params, c, g, wa, wo);
    const int64_t M = live_samples(n_points, n_rays_dev, n_rays, n_samples);
    const int64_t n_tiles = M > 0 ? (M + 31) / 32 : 0;
    const int64_t stride = (int64_t)gridDim.x * nw;
    const uint32_t plane_bytes = (uint32_t)m_pad * 4u;
    int64_t tile = (int64_t)blockIdx.x * nw + wave;
    if (tile >= n_tiles) return;
    auto sample_of = [&](int64_t tl, int t) -> uint32_t { const int64_t m = tl * 32 + 16 * t + c; return (uint32_t)(m < M ? m : M - 1); };
    u32x4 cur[2], nxt[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) cur[t] = load_xb(featp, plane_bytes, sample_of(tile, t), g);
    while (tile < n_tiles) {
        const int64_t nt = tile + stride;
#pragma unroll
        for (int t = 0; t < 2; ++t) nxt[t] = load_xb(featp, plane_bytes, sample_of(nt < n_tiles ? nt : tile, t), g);   // unconditional prefetch
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float part = 0.0f;
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) {
                const f32x4 Z = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[jt], __builtin_bit_cast(f16x8, cur[t]), f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) part += wo[jt][r] * fmaxf(Z[r], 0.0f);
            }
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            const int64_t m = tile * 32 + 16 * t + c;
            if (g == 0 && m < M) sigma[m] = finite_or_clipped<true>(part, clip_flag);
        }
        cur[0] = nxt[0]; cur[1] = nxt[1];
        tile = nt;
    }
}

// what one 32-sample step of the backward needs from memory
struct Step16 {
    float ds[2];
    u32x4 xb[2];            // layer-1 B operands of the two column tiles
    u32x4 xraw[2][2];       // weight-gradient B operand, before the half select: plane (8it + c/2), samples 8g..8g+7
};

__device__ __forceinline__ void load_step(const uint32_t* __restrict__ featp, const float* __restrict__ d_sigma, uint32_t plane_bytes,
                                          int64_t M, int64_t tile, int c, int g, Step16& s) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t m = tile * 32 + 16 * t + c;
        const uint32_t mc = (uint32_t)(m < M ? m : M - 1);
        s.xb[t] = load_xb(featp, plane_bytes, mc, g);
        const float v = d_sigma[mc];
        s.ds[t] = m < M ? v : 0.0f;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        // samples tile*32 + 8g .. +7 of plane 8it + (c >> 1); the planes are zero-filled up to the next multiple of 32 samples
        const uint32_t off = (uint32_t)(8 * it + (c >> 1)) * plane_bytes + (uint32_t)(tile * 32 + 8 * g) * 4u;
        s.xraw[it][0] = ld32<u32x4>(featp, off);
        s.xraw[it][1] = ld32<u32x4>(featp, off + 16u);
    }
}

// LDS: [dW n_mlp floats][per wave: T_dz  H x F16_TS halves][WT: 2 x KB x 64 lanes x 16 bytes]
template <int HT>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK, 2)
mlp_backward_f16_kernel(const float* __restrict__ params, int n_mlp, const uint32_t* __restrict__ featp, int64_t m_pad, int64_t n_points,
                        const int32_t* __restrict__ n_rays_dev, int n_rays, int n_samples, const float* __restrict__ d_sigma,
                        float* __restrict__ dfeat, float* __restrict__ slabs, int want_dfeat) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 16 * HT;
    constexpr int KB = (HT + 1) / 2;            // 32-neuron K blocks of the input-gradient product
    const int nw = blockDim.x >> 6;
    float* dW = smem;
    for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) dW[i] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    f16* T = reinterpret_cast<f16*>(smem + n_mlp) + wave * (H * F16_TS);

    f16x8 wa[HT]; float wo[HT][4];
    load_w1_frags<HT>(params, c, g, wa, wo);
    // A fragments of dX = W1^T dZ: row = feature 16it + c, K slot 8g+i = neuron 32kb + (i < 4 ? 4g+i : 16 + 4g+i-4).  They are
    // the same for every wave: built once per workgroup in LDS ([it][kb][lane] x 16 bytes) and re-read every step - as
    // registers they would push the HT = 4 kernel past the 256 VGPRs of two waves per SIMD.
    f16x8* WT = reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(smem + n_mlp) + nw * (H * F16_TS));
    if (wave == 0) {
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                f16x8 v;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int n = 32 * kb + (i < 4 ? 4 * g + i : 16 + 4 * g + (i - 4));
                    v[i] = n < H ? (f16)params[n * 32 + 16 * it + c] : (f16)0.0f;
                }
                WT[(it * KB + kb) * 64 + lane] = v;
            }
    }
    __syncthreads();
    f32x4 dW1_acc[HT][2];
    float dWo_acc[HT][4];
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) {
        dW1_acc[jt][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        dW1_acc[jt][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < 4; ++r) dWo_acc[jt][r] = 0.0f;
    }
    const uint32_t half_sel = (c & 1) ? 0x07060302u : 0x05040100u;     // v_perm selector: high or low halves of two dwords

    const int64_t M = live_samples(n_points, n_rays_dev, n_rays, n_samples);
    const int64_t n_tiles = M > 0 ? (M + 31) / 32 : 0;
    const int64_t stride = (int64_t)gridDim.x * nw;
    const uint32_t plane_bytes = (uint32_t)m_pad * 4u;
    int64_t tile = (int64_t)blockIdx.x * nw + wave;
    Step16 cur;
    if (tile < n_tiles) load_step(featp, d_sigma, plane_bytes, M, tile, c, g, cur);
    while (tile < n_tiles) {
        const int64_t nt = tile + stride;
        Step16 nxt;
        load_step(featp, d_sigma, plane_bytes, M, nt < n_tiles ? nt : tile, c, g, nxt);        // unconditional prefetch
        const bool any = (cur.ds[0] != 0.0f) | (cur.ds[1] != 0.0f);
        if (__ballot(any) == 0ull) {                    // nothing flows back into this step
            if (want_dfeat) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int64_t m = tile * 32 + 16 * t + c;
                    if (m < M) {
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            st32<float>(dfeat, (uint32_t)(16 * (k >> 2) + 4 * g + (k & 3)) * plane_bytes + (uint32_t)m * 4u, 0.0f);
                    }
                }
            }
        } else {
            // exact power-of-two scale of the step: the largest |d_sigma| lands in [1, 2) before dZ is rounded to fp16
            const float mx = wave_max(fmaxf(fabsf(cur.ds[0]), fabsf(cur.ds[1])));
            uint32_t be = (__float_as_uint(mx) >> 23) & 0xFFu;
            be = be < 1u ? 127u : (be > 253u ? 253u : be);            // (denormal maximum: no scaling)
            const float sc_dn = __uint_as_float((254u - be) << 23), sc_up = __uint_as_float(be << 23);
            uint32_t dzp[HT][2][2];                 // dZ of the lane's neurons 16jt+4g+{0..3} and samples t, as packed half pairs
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float dsn = cur.ds[t] * sc_dn;
#pragma unroll
                for (int jt = 0; jt < HT; ++jt) {
                    const f32x4 Z = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[jt], __builtin_bit_cast(f16x8, cur.xb[t]), f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
                    float dz[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dWo_acc[jt][r] += cur.ds[t] * fmaxf(Z[r], 0.0f);
                        dz[r] = Z[r] > 0.0f ? dsn * wo[jt][r] : 0.0f;
                        T[(16 * jt + 4 * g + r) * F16_TS + 16 * t + c] = (f16)dz[r];
                    }
                    dzp[jt][t][0] = pack_h2(dz[0], dz[1]);
                    dzp[jt][t][1] = pack_h2(dz[2], dz[3]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // dW1[neuron][feature] += 2^e * sum over the 32 samples of dZ[neuron][s] X[feature][s]
            f16x8 xs[2];
#pragma unroll
            for (int it = 0; it < 2; ++it)
                xs[it] = frag_from_dwords(__builtin_amdgcn_perm(cur.xraw[it][0][1], cur.xraw[it][0][0], half_sel),
                                          __builtin_amdgcn_perm(cur.xraw[it][0][3], cur.xraw[it][0][2], half_sel),
                                          __builtin_amdgcn_perm(cur.xraw[it][1][1], cur.xraw[it][1][0], half_sel),
                                          __builtin_amdgcn_perm(cur.xraw[it][1][3], cur.xraw[it][1][2], half_sel));
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) {
                const f16x8 a = *reinterpret_cast<const f16x8*>(T + (16 * jt + c) * F16_TS + 8 * g);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const f32x4 D = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xs[it], f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dW1_acc[jt][it][r] += sc_up * D[r];
                }
            }
            if (want_dfeat) {
                int wlane = lane;
                asm volatile("" : "+v"(wlane));              // opaque: keeps the compiler from hoisting the LDS reads out of the loop
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int64_t m = tile * 32 + 16 * t + c;
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        f32x4 D = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb) {
                            const bool second = 2 * kb + 1 < HT;               // HT == 1: the upper half of the K block is padding
                            const int j1 = second ? 2 * kb + 1 : 2 * kb;
                            const f16x8 b = frag_from_dwords(dzp[2 * kb][t][0], dzp[2 * kb][t][1],
                                                             second ? dzp[j1][t][0] : 0u, second ? dzp[j1][t][1] : 0u);
                            D = __builtin_amdgcn_mfma_f32_16x16x32_f16(WT[(it * KB + kb) * 64 + wlane], b, D, 0, 0, 0);
                        }
                        if (m < M) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                st32<float>(dfeat, (uint32_t)(16 * it + 4 * g + r) * plane_bytes + (uint32_t)m * 4u, sc_up * D[r]);
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                // T is rewritten by the next step
        }
        cur = nxt;
        tile = nt;
    }
    // the waves add their register accumulators to the workgroup's LDS copy one after the other: fixed order, reproducible
    float* dW1 = dW;
    float* dWo = dW + H * 32;
    for (int turn = 0; turn < nw; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) {
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dW1[(16 * jt + 4 * g + r) * 32 + 16 * it + c] += dW1_acc[jt][it][r];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = dWo_acc[jt][r];
                    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                    if (c == 0) dWo[16 * jt + 4 * g + r] += v;
                }
            }
        }
        __syncthreads();
    }
    float* slab = slabs + (size_t)blockIdx.x * n_mlp;
    for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) slab[i] = dW[i];
}

// ================================================================================================
// Every other supported shape (any encoding on half2 pair planes, 16..256 neurons, 1..3 hidden layers, any activation; the shape
// class the north star names besides the default one: frequency encoding + <= 128-wide SIREN / ReLU MLP) runs the general kernels:
// forward lnr_f16_fwd_kernel.h / lnr_density_f16_fwd.hip, backward lnr_f16_bwd_kernel.h / lnr_density_f16_bwd.hip.
// ------------------------------------------------------------------------------------------------ host side
static bool f16_fast_class(const LnrNetSpec* spec) {
    return spec->encoding == LNR_ENC_HASHGRID && spec->n_features == 2 && spec->enc_dim == 32 && spec->in_dim == 32 &&
           spec->n_hidden == 1 && spec->activation == LNR_ACT_RELU && spec->n_neurons <= 64;
}

// frequency encodings the fused kernels evaluate themselves (lnr_f16_freq.h: at most 12 (sin, cos) pairs per lane = three K blocks)
static bool lnr_f16_fused_freq_shape(const LnrNetSpec* spec) {
    return spec->encoding == LNR_ENC_FREQUENCY && spec->n_frequencies >= 1 && lnr_freq_kt(spec->n_frequencies) != 0;
}

// what LNR_PREC_F16 covers: half2 pair planes need an even number of features per level; the general kernels hold dW in registers
// (<= 128 neurons, <= 3 hidden layers) and the weights plus the transposes of four waves in LDS
bool lnr_f16_supported(const LnrNetSpec* spec) {
    if (f16_fast_class(spec)) return true;
    if (spec->encoding == LNR_ENC_HASHGRID && (spec->n_features & 1)) return false;
    if (spec->enc_dim & 1) return false;
    const int H = spec->n_neurons;
    // 256 neurons: with ONE hidden layer (H x in_dim + the output row: the weight gradient is 128 registers per wave); a 256 x 256
    // hidden matrix neither fits the LDS beside the exchange buffers nor the registers as a gradient
    if (!(H == 16 || H == 32 || H == 64 || H == 128 || (H == 256 && spec->n_hidden == 1))) return false;
    if (spec->n_hidden < 1 || spec->n_hidden > F16_NH_MAX || spec->in_dim > 32 * F16_KB_MAX) return false;
    const size_t lds = lnr_f16_fused_freq_shape(spec) ? lnr_f16_freq_bwd_lds(spec) : lnr_f16_gen_bwd_lds(spec);
    return lds > 0 && lds <= (size_t)LNR_LDS_LIMIT;
}

bool lnr_f16_fused_freq(const LnrNetSpec* spec) {
    return spec->precision == LNR_PREC_F16 && lnr_f16_fused_freq_shape(spec) && !lnr_wide_class(spec) && lnr_f16_supported(spec);
}

static size_t f16_bwd_lds(const LnrNetSpec* spec) {
    const int ht = spec->n_neurons / 16, kb = (ht + 1) / 2;
    return (size_t)spec->n_mlp_params * sizeof(float) + (size_t)(LNR_DENSITY_BLOCK / 64) * spec->n_neurons * F16_TS * sizeof(f16) +
           (size_t)2 * kb * 64 * 16;
}

int lnr_mlp_fwd_f16(const LnrNetSpec* spec, const float* params, const void* featp, int64_t m_pad, const MlpPoints* pt, float* sigma,
                    const PointSrc* src, hipStream_t st) {
    const int64_t tiles = (pt->n_points + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > LNR_DENSITY_MAX_BLOCKS) blocks = LNR_DENSITY_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), block(LNR_DENSITY_BLOCK);
    const uint32_t* fp = reinterpret_cast<const uint32_t*>(featp);
    if (f16_fast_class(spec)) {
        switch (spec->n_neurons / 16) {
            case 1: hipLaunchKernelGGL(mlp_forward_f16_kernel<1>, grid, block, 0, st, params, fp, m_pad, pt->n_points, pt->n_rays_dev, pt->n_rays, pt->n_samples, sigma, pt->clip_flag); break;
            case 2: hipLaunchKernelGGL(mlp_forward_f16_kernel<2>, grid, block, 0, st, params, fp, m_pad, pt->n_points, pt->n_rays_dev, pt->n_rays, pt->n_samples, sigma, pt->clip_flag); break;
            default: hipLaunchKernelGGL(mlp_forward_f16_kernel<4>, grid, block, 0, st, params, fp, m_pad, pt->n_points, pt->n_rays_dev, pt->n_rays, pt->n_samples, sigma, pt->clip_flag); break;
        }
        return LNR_OK;
    }
    if (lnr_f16_fused_freq(spec)) return lnr_mlp_fwd_f16_freq(spec, params, fp, m_pad, pt, sigma, blocks, src, st);
    return lnr_mlp_fwd_f16_gen(spec, params, fp, m_pad, pt, sigma, blocks, src, st);
}

// weight-gradient slabs lnr_mlp_bwd_f16 writes for up to n_points points (one per workgroup)
int lnr_f16_bwd_slabs(const LnrNetSpec* spec, int64_t n_points) {
    const int64_t tiles = (n_points + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > LNR_BWD_MAX_BLOCKS) blocks = LNR_BWD_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    if (!f16_fast_class(spec) && blocks > 256) blocks = 256;
    return (int)blocks;
}

int lnr_mlp_bwd_f16(const LnrNetSpec* spec, const float* params, const void* featp, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                    float* dfeat, float* slabs, int want_dfeat, int* n_slabs, const PointSrc* src, float* d_pts, hipStream_t st) {
    const int64_t tiles = (pt->n_points + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > LNR_BWD_MAX_BLOCKS) blocks = LNR_BWD_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    const dim3 block(LNR_DENSITY_BLOCK);
    const uint32_t* fp = reinterpret_cast<const uint32_t*>(featp);
    if (f16_fast_class(spec)) {
        *n_slabs = (int)blocks;
        const dim3 grid((unsigned)blocks);
        const size_t lds = f16_bwd_lds(spec);
#define LNR_F16_BWD(HT)                                                                                                          \
    do {                                                                                                                         \
        int rc_ = f16_set_lds(mlp_backward_f16_kernel<HT>, lds, "lnr_density_backward");                                         \
        if (rc_) return rc_;                                                                                                     \
        hipLaunchKernelGGL(mlp_backward_f16_kernel<HT>, grid, block, lds, st, params, spec->n_mlp_params, fp, m_pad, pt->n_points, \
                           pt->n_rays_dev, pt->n_rays, pt->n_samples, d_sigma, dfeat, slabs, want_dfeat);                       \
    } while (0)
        switch (spec->n_neurons / 16) {
            case 1: LNR_F16_BWD(1); break;
            case 2: LNR_F16_BWD(2); break;
            default: LNR_F16_BWD(4); break;
        }
#undef LNR_F16_BWD
        return LNR_OK;
    }
    if (blocks > 256) blocks = 256;                                    // general kernels: one workgroup per CU (LDS), persistent over the steps
    *n_slabs = (int)blocks;
    if (lnr_f16_fused_freq(spec)) return lnr_mlp_bwd_f16_freq(spec, params, fp, m_pad, pt, d_sigma, dfeat, slabs, want_dfeat, (int)blocks, src, d_pts, st);
    return lnr_mlp_bwd_f16_gen(spec, params, fp, m_pad, pt, d_sigma, dfeat, slabs, want_dfeat, (int)blocks, src, d_pts, st);
}

// ------------------------------------------------------------------------------------------------ layout self-test
// D = A(16x32) * B(32x16) with asymmetric small-integer operands (exact in fp16), checked against the layout the kernels
// assume: A lane -> A[l&15][8*(l>>4) + i], B lane -> B[8*(l>>4) + i][l&15], D reg r -> D[4*(l>>4) + r][l&15].
__global__ void selftest_mfma_f16_kernel(float* out) {
    const int lane = threadIdx.x;
    const int c = lane & 15, g = lane >> 4;
    auto A = [](int i, int k) { return (float)((i * 3 + k * 5) % 17 - 8) * 0.25f; };
    auto B = [](int k, int j) { return (float)((k * 7 - j * 2) % 13 - 6) * 0.5f; };
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (f16)A(c, 8 * g + i); b[i] = (f16)B(8 * g + i, c); }
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
    float err = 0.0f;
    for (int r = 0; r < 4; ++r) {
        float ref = 0.0f;
        for (int k = 0; k < 32; ++k) ref += A(4 * g + r, k) * B(k, c);
        err = fmaxf(err, fabsf(ref - d[r]));
    }
    // the half select of the weight-gradient operand
    const uint32_t lo = __builtin_amdgcn_perm(0xBBBB2222u, 0xAAAA1111u, 0x05040100u), hi = __builtin_amdgcn_perm(0xBBBB2222u, 0xAAAA1111u, 0x07060302u);
    if (lo != 0x22221111u || hi != 0xBBBBAAAAu) err = 1e9f;
    err = wave_max(err);
    if (lane == 0) out[1] = err;
}

int lnr_selftest_mfma_f16(float* out, hipStream_t st) {
    hipLaunchKernelGGL(selftest_mfma_f16_kernel, dim3(1), dim3(64), 0, st, out);
    return LNR_OK;
}
