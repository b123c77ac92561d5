// General fp32 backward of the density MLP with the weight gradient in REGISTERS (gfx950, v_mfma_f32_16x16x4_f32).
// Included by lnr_density_regs.hip, one translation unit per (hidden width, depth): -DLNR_HT=<n_neurons/16> -DLNR_NH=<hidden layers>.
//
// Replaces, for networks of up to three hidden layers and up to 128 (padded) inputs, mlp_backward_kernel (lnr_density_impl.h), which
// adds every 16-sample tile's 16 x 16 weight-gradient products into an LDS copy of the whole gradient with LDS atomics: one
// read-modify-write of all H x K elements per layer and per 16 samples - 50.8 ms for frequency-12 -> 128 x 2 at 2.1 M samples, 4 % of
// the fp32 MFMA peak, with the weights read from L2 because gradient copy and weights did not both fit the LDS.
// Here (the structure of the fp16 general backward, lnr_f16_bwd_kernel.h):
//   * a workgroup of four waves takes 64 samples per step, one 16-sample tile per wave; forward, back-propagation and input
//     gradient of a tile are its wave's own work, as before;
//   * the WEIGHT GRADIENT is owned by rows: wave w keeps the gradient of row tiles w RT .. w RT + RT - 1 (RT = HT / 4) of every layer
//     in accumulator registers for the whole launch - (KT1M + (NH - 1) HT) x 4 RT registers - and runs them over the transposed dZ /
//     layer-input images of all four tiles, which the waves leave in LDS ([neuron][16 samples], 80-byte rows): two workgroup barriers
//     per layer and step, and the MFMAs accumulate straight into those registers;
//   * depth NH and the first layer's K blocks (KT1M: 2, 5 or 8 blocks of 16 inputs, the network's own count rounded up) are compile
//     time, so that every accumulator has a static register; pre-activations of all layers stay in registers;
//   * LDS holds the weights too when they fit beside the images - all of them, or the hidden matrices only (128 x 2 with 80 inputs:
//     80 KB of images + 64 KB); what does not fit is read through L1 / L2;
//   * one slab per workgroup, every element written once at the end by the wave that owns its row (no zero-fill, no atomics).
// Semantics = oracle/network.py (tinycudann FullyFusedMLP behind src/models/nerf_tcnn.py:35-38, backward through optimizer.py:366).
#pragma once
#include "lnr_density_impl.h"

// Hidden layer Zn = Wl act(Z), one K block at a time: the HT weight fragments of a block are requested together and only then
// multiplied (at one wave per SIMD the compiler's own order - fragment, wait, four MFMAs, next fragment - exposes an LDS / L2 round
// trip per fragment).
template <int HT>
__device__ __forceinline__ void hidden_forward_blocks(const float* Wl, int row_stride, int act, int c, int g, const f32x4 Z[HT], f32x4 Zn[HT]) {
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) Zn[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kt = 0; kt < HT; ++kt) {
        float4 wa[HT];
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) wa[jt] = *reinterpret_cast<const float4*>(Wl + (16 * jt + c) * row_stride + 16 * kt + 4 * g);
        const float a0 = act_fwd(Z[kt].x, act), a1 = act_fwd(Z[kt].y, act), a2 = act_fwd(Z[kt].z, act), a3 = act_fwd(Z[kt].w, act);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) MFMA4(Zn[jt], wa[jt], a0, a1, a2, a3);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int HT, int NH, int KT1M, int WM, int ACT>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK, 1)
mlp_backward_regs_kernel(const LnrNetSpec spec, const float* __restrict__ params, const float* __restrict__ feat, int64_t m_pad,
                         int64_t n_points, const int32_t* __restrict__ n_rays_dev, int n_rays, int n_samples,
                         const float* __restrict__ d_sigma, float* __restrict__ dfeat, float* __restrict__ slabs, int want_dfeat) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 16 * HT, TS = LNR_TDZ_STRIDE, NW = LNR_DENSITY_BLOCK / 64;
    constexpr int RT = HT >= NW ? HT / NW : 1;                 // row tiles a wave owns (HT < 4: waves >= HT own none)
    constexpr int NHH = NH > 1 ? NH - 1 : 1;
    constexpr int IMG = H * TS;                                // floats of one tile's image
    const int in_dim = spec.in_dim, kt1 = in_dim >> 4;
    const int n_mlp = spec.n_mlp_params;
    // WM: which matrices sit in LDS (padded rows: lnr_fill_w_lds) - 1: all of them; 2: the hidden ones and the output rows (read twice
    // per step: forward and W^T dZ), the first layer's through L1 / L2; 0: none
    const int n_w1 = H * in_dim;
    const int n_lds = WM == 0 ? 0 : lnr_w_lds_floats(H, in_dim, NH, WM == 1);
    if (WM != 0) lnr_fill_w_lds(smem, params, H, in_dim, NH, WM == 1);
    __syncthreads();
    const int s1 = WM == 1 ? lnr_w_stride(in_dim) : in_dim;          // row strides (floats)
    constexpr int sh = WM != 0 ? H + 4 : H;
    const float* W1 = WM == 1 ? smem : params;
    const float* Wh = WM == 1 ? smem + H * s1 : (WM == 2 ? smem : params + n_w1);
    const float* Wo = Wh + (NH - 1) * H * sh;
    const int act = ACT >= 0 ? ACT : spec.activation;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    float* img_dz = smem + n_lds;                              // [NW tiles][H][TS]
    float* img_a = img_dz + NW * IMG;                          // (NH > 1) the layer inputs, same shape
    float* T_dz = img_dz + wave * IMG;
    float* T_a = img_a + wave * IMG;
    const bool owner = wave * RT < HT;                         // this wave owns row tiles jt0 .. jt0 + RT - 1
    const int jt0 = owner ? wave * RT : 0;

    f32x4 dw1[RT][KT1M], dwh[NHH][RT][HT], dWo_acc[HT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int kt = 0; kt < KT1M; ++kt) dw1[i][kt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int l = 0; l < NHH; ++l)
#pragma unroll
            for (int kt = 0; kt < HT; ++kt) dwh[l][i][kt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) dWo_acc[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int64_t M = n_rays_dev ? (int64_t)lnr_live_rays(n_rays, n_rays_dev) * n_samples : n_points;
    const int64_t n_tiles = M > 0 ? (M + 15) / 16 : 0;
    const int64_t n_steps = (n_tiles + (int64_t)gridDim.x * NW - 1) / ((int64_t)gridDim.x * NW);        // workgroup-uniform (the body has barriers)
    // the inputs of a step (features of the lane's sample, its d_sigma) are requested one step ahead
    auto sample_of = [&](int64_t step, bool& valid) -> int64_t {
        const int64_t mm = ((step * gridDim.x + blockIdx.x) * NW + wave) * 16 + c;
        valid = mm < M;
        return valid ? mm : M - 1;
    };
    float xf_n[KT1M][4], ds_n = 0.0f;
    if (n_steps > 0) {
        bool v;
        const int64_t m0 = sample_of(0, v);
        load_tile_inputs<KT1M>(spec, feat, m_pad, m0, g, kt1, xf_n);
        ds_n = d_sigma[m0];
    }
#pragma unroll 1
    for (int64_t step = 0; step < n_steps; ++step) {
        const int64_t tile0 = (step * gridDim.x + blockIdx.x) * NW;          // the workgroup's four tiles: 64 consecutive samples
        bool valid;
        const int64_t m = sample_of(step, valid);
        const float ds = valid ? ds_n : 0.0f;
        float xf[KT1M][4];
#pragma unroll
        for (int kt = 0; kt < KT1M; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) xf[kt][r] = xf_n[kt][r];
        {
            bool v;
            const int64_t mn = sample_of(step + 1 < n_steps ? step + 1 : step, v);
            load_tile_inputs<KT1M>(spec, feat, m_pad, mn, g, kt1, xf_n);
            ds_n = d_sigma[mn];
        }
        const bool mine = __ballot(ds != 0.0f) != 0ull;                      // a gradient flows back into this wave's tile
        if (!__syncthreads_or(mine ? 1 : 0)) {                               // ... into none of the four
            if (want_dfeat && valid) {
                for (int kt = 0; kt < kt1; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int k = 16 * kt + 4 * g + r; if (k < spec.enc_dim) dfeat[(size_t)k * m_pad + m] = 0.0f; }
            }
            continue;
        }
        // forward, pre-activations of every layer kept (a tile without gradient still runs: its images must be written, as zeros by ds = 0)
        f32x4 Z[NH][HT];
        layer1_from_regs<HT, KT1M>(W1, c, g, kt1, s1, xf, Z[0]);
#pragma unroll
        for (int l = 1; l < NH; ++l) hidden_forward_blocks<HT>(Wh + (l - 1) * H * sh, sh, act, c, g, Z[l - 1], Z[l]);
        f32x4 dA[HT];
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) {
            const float4 wo = *reinterpret_cast<const float4*>(Wo + 16 * jt + 4 * g);
            dA[jt] = f32x4{ds * wo.x, ds * wo.y, ds * wo.z, ds * wo.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) dWo_acc[jt][r] += ds * act_fwd(Z[NH - 1][jt][r], act);
        }
#pragma unroll
        for (int l = NH - 1; l >= 0; --l) {
            f32x4 dZ[HT];
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dZ[jt][r] = dA[jt][r] * act_bwd(Z[l][jt][r], act);
                    T_dz[(16 * jt + 4 * g + r) * TS + c] = dZ[jt][r];
                }
            }
            if (l > 0) {
#pragma unroll
                for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T_a[(16 * jt + 4 * g + r) * TS + c] = act_fwd(Z[l - 1][jt][r], act);
            }
            __syncthreads();
            // weight gradient of the rows this wave owns over all four tiles: rows = neurons 16 (jt0 + i) .., columns = the layer's inputs,
            // contraction over a tile's 16 samples (A operand: the dZ image; B operand: the layer's input, 4 consecutive samples of one
            // input - from the feature planes for the first layer (16 contiguous bytes, no image), from the input images for the others)
            if (owner) {
                if (l == 0) {
                    // B operands straight from the feature planes, all K blocks of a tile requested together
                    auto load_b = [&](int s, float4 b4[KT1M]) {
                        int64_t tile_base = (tile0 + s) * 16;
                        if (tile_base >= M) tile_base = 0;                  // (a tile past the end: its dZ image is zero)
                        const bool whole = tile_base + 16 <= M;
#pragma unroll
                        for (int kt = 0; kt < KT1M; ++kt) {
                            const int k = 16 * kt + c;
                            const bool real = kt < kt1 && k < spec.enc_dim;
                            const float* p = feat + (size_t)(real ? k : 0) * m_pad;
                            if (whole) b4[kt] = *reinterpret_cast<const float4*>(p + tile_base + 4 * g);
                            else {                      // ragged last tile: dZ of the padding samples is 0, any finite value will do
                                b4[kt].x = p[min(tile_base + 4 * g + 0, M - 1)]; b4[kt].y = p[min(tile_base + 4 * g + 1, M - 1)];
                                b4[kt].z = p[min(tile_base + 4 * g + 2, M - 1)]; b4[kt].w = p[min(tile_base + 4 * g + 3, M - 1)];
                            }
                            if (!real) b4[kt] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                        }
                    };
#pragma unroll 1
                    for (int s = 0; s < NW; ++s) {
                        const float* Sdz = img_dz + s * IMG;
                        float4 b4[KT1M], a4[RT];
                        load_b(s, b4);                  // (the lines were read by this step's forward a moment ago: L2)
#pragma unroll
                        for (int i = 0; i < RT; ++i) a4[i] = *reinterpret_cast<const float4*>(Sdz + (16 * (jt0 + i) + c) * TS + 4 * g);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int kt = 0; kt < KT1M; ++kt) {
                            if (kt < kt1) {
#pragma unroll
                                for (int i = 0; i < RT; ++i) MFMA4(dw1[i][kt], a4[i], b4[kt].x, b4[kt].y, b4[kt].z, b4[kt].w);
                            }
                        }
                    }
                } else {
#pragma unroll 1
                    for (int s = 0; s < NW; ++s) {
                        const float* Sdz = img_dz + s * IMG;
                        const float* Sa = img_a + s * IMG;
                        float4 a4[RT], b4[HT];
#pragma unroll
                        for (int i = 0; i < RT; ++i) a4[i] = *reinterpret_cast<const float4*>(Sdz + (16 * (jt0 + i) + c) * TS + 4 * g);
#pragma unroll
                        for (int kt = 0; kt < HT; ++kt) b4[kt] = *reinterpret_cast<const float4*>(Sa + (16 * kt + c) * TS + 4 * g);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int kt = 0; kt < HT; ++kt)
#pragma unroll
                            for (int i = 0; i < RT; ++i) MFMA4(dwh[l > 0 ? l - 1 : 0][i][kt], a4[i], b4[kt].x, b4[kt].y, b4[kt].z, b4[kt].w);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // input gradient of the layer (this wave's tile, from its registers)
            if (l > 0) {
                const float* Wl = Wh + (l - 1) * H * sh;
#pragma unroll
                for (int kt = 0; kt < HT; ++kt) {
                    float wt[HT][4];                     // W^T operands of the block, requested together
#pragma unroll
                    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) wt[jt][r] = Wl[(16 * jt + 4 * g + r) * sh + 16 * kt + c];
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 D = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) D = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[jt][r], dZ[jt][r], D, 0, 0, 0);
                    dA[kt] = D;
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (want_dfeat) {
#pragma unroll 1
                for (int kt = 0; kt < kt1; ++kt) {
                    float wt[HT][4];                     // W1^T operands of the block, requested together
#pragma unroll
                    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) wt[jt][r] = W1[(16 * jt + 4 * g + r) * s1 + 16 * kt + c];
                    f32x4 D = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) D = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[jt][r], dZ[jt][r], D, 0, 0, 0);
                    if (valid) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const int k = 16 * kt + 4 * g + r; if (k < spec.enc_dim) dfeat[(size_t)k * m_pad + m] = D[r]; }
                    }
                }
            }
            __syncthreads();                  // the images are rewritten by the next layer / step
        }
    }

    // the workgroup's slab: accumulator (row 4g + r, column c) of tile (jt0 + i, kt), by the wave that owns the row
    float* slab = slabs + (size_t)blockIdx.x * n_mlp;
    float* slab_h = slab + H * in_dim;
    float* slab_o = slab_h + (NH - 1) * H * H;
    if (owner) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
#pragma unroll
            for (int kt = 0; kt < KT1M; ++kt) {
                if (kt < kt1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[(16 * (jt0 + i) + 4 * g + r) * in_dim + 16 * kt + c] = dw1[i][kt][r];
                }
            }
            if constexpr (NH > 1) {
#pragma unroll
                for (int l = 0; l < NH - 1; ++l)
#pragma unroll
                    for (int kt = 0; kt < HT; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) slab_h[(l * H + 16 * (jt0 + i) + 4 * g + r) * H + 16 * kt + c] = dwh[l][i][kt][r];
            }
        }
    }
    // output row: the four waves' sums over their own tiles, through LDS (the images are free now)
    float* part = img_dz;                       // [NW][H]
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = dWo_acc[jt][r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (c == 0) part[wave * H + 16 * jt + 4 * g + r] = v;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * H; i += blockDim.x) {        // (the output matrix is padded to 16 rows: rows 1..15 take no gradient)
        float v = 0.0f;
        if (i < H) { for (int w = 0; w < NW; ++w) v += part[w * H + i]; }
        slab_o[i] = v;
    }
}
