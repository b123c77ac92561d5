// fp32 MFMA MLP kernels of the density network on feature planes (gfx950); included by lnr_density_ht.hip, one
// translation unit per hidden width (-DLNR_HT=<n_neurons/16>) so that the five widths compile in parallel.
//
// Replaces the fully-fused MLP half of the tinycudann NetworkWithInputEncoding the reference calls at
// src/models/nerf_tcnn.py:63-72 (forward) and through loss.backward() (src/mapping/optimizer.py:366).
// Semantics = oracle/network.py.  The encoding half lives in lnr_encode.hip.
//
// Kernel design (CDNA4, 64-wide waves, v_mfma_f32_16x16x4_f32 = exact fp32 fma chains):
//   * "transposed" MLP: a wave owns a tile of 16 samples as the N (column) dimension, neurons are the M (row)
//     dimension:  Z^T[j][c] = sum_k W[j][k] * X^T[k][c].   lane = (c = lane&15 -> sample, g = lane>>4 -> k-slot).
//   * the MFMA result layout (row = 4g+r, col = c) is exactly the B-operand layout of the next layer if k-slot g of
//     step (jt,r) is defined to be neuron 16jt+4g+r, so activations chain through registers: no shuffles, no LDS.
//   * the same trick assigns input features: lane (c,g) reads features 16kt+4g+r (r=0..3) of sample c from the
//     feature planes ([feature][sample], 64-byte coalesced per lane group).
//   * backward: dX^T = W^T dZ^T needs no data movement either (dZ is already a B operand and the result lands on
//     the lane that owns those features, which stores them to the d_feature planes); only the weight gradient
//     contracts over samples (= lanes): dZ goes through a 16x16 LDS transpose per tile, X^T is read straight from
//     the feature planes; partial weight gradients are kept in registers (default shape) or summed in LDS per
//     workgroup and written to a per-workgroup slab that a second kernel reduces (no global atomics on weights).
#pragma once
#include "lnr_common.h"
#include "lnr_density_api.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// sin / cos for the Sine activation (SIREN networks: three evaluations per neuron, sample and layer of a backward - the libm pair, with
// its exact range reduction inlined at every use, was most of the fp32 SIREN kernels' time and code).  Cody-Waite reduction in three
// fma steps + minimax polynomials on [-pi/4, pi/4]: max absolute error 9.3e-8 for |v| < 65536 (libm sinf: 3.3e-8; checked against
// float64 on 20 M points per range), beyond that the libm call.
__device__ __forceinline__ void lnr_sincos_fast(float v, float* s_out, float* c_out) {
    const float k = __builtin_rintf(v * 0.636619772367581343f);
    float r = __builtin_fmaf(-k, 1.57079637050628662109375f, v);
    r = __builtin_fmaf(-k, -4.37113900018624283e-8f, r);
    r = __builtin_fmaf(-k, -1.7151245100059e-15f, r);
    const float z = r * r;
    float sp = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = __builtin_fmaf(z, sp, -1.6666654611e-1f);
    const float s = __builtin_fmaf(z * r, sp, r);
    float cp = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = __builtin_fmaf(z, cp, 4.166664568298827e-2f);
    const float c = __builtin_fmaf(z * z, cp, __builtin_fmaf(z, -0.5f, 1.0f));
    const int q = (int)k;
    const float a = (q & 1) ? c : s, b = (q & 1) ? s : c;                 // q = 0: (s, c)  1: (c, -s)  2: (-s, -c)  3: (-c, s)
    *s_out = (q & 2) ? -a : a;
    *c_out = ((q + 1) & 2) ? -b : b;
}
__device__ __forceinline__ float lnr_sin(float v) {
    if (__builtin_expect(__builtin_fabsf(v) < 65536.0f, 1)) { float s, c; lnr_sincos_fast(v, &s, &c); return s; }
    return sinf(v);
}
__device__ __forceinline__ float lnr_cos(float v) {
    if (__builtin_expect(__builtin_fabsf(v) < 65536.0f, 1)) { float s, c; lnr_sincos_fast(v, &s, &c); return c; }
    return cosf(v);
}

#define LNR_K_ACT 10.0f
__device__ __forceinline__ float act_fwd(float v, int kind) {
    switch (kind) {
        case LNR_ACT_RELU: return fmaxf(v, 0.0f);
        case LNR_ACT_SINE: return lnr_sin(v);
        case LNR_ACT_LEAKY_RELU: return v > 0.0f ? v : 0.01f * v;
        case LNR_ACT_EXPONENTIAL: return expf(v);
        case LNR_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        // (tiny-cuda-nn: both on K_ACT v, result / K_ACT, K_ACT = 10 - oracle/network.py)
        case LNR_ACT_SQUAREPLUS: { const float y = LNR_K_ACT * v; return 0.5f * (y + sqrtf(y * y + 4.0f)) * (1.0f / LNR_K_ACT); }
        case LNR_ACT_SOFTPLUS: { const float y = LNR_K_ACT * v; return y > 20.0f ? v : log1pf(expf(y)) * (1.0f / LNR_K_ACT); }
        case LNR_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
// derivative of the activation with respect to its pre-activation v
__device__ __forceinline__ float act_bwd(float v, int kind) {
    switch (kind) {
        case LNR_ACT_RELU: return v > 0.0f ? 1.0f : 0.0f;
        case LNR_ACT_SINE: return lnr_cos(v);
        case LNR_ACT_LEAKY_RELU: return v > 0.0f ? 1.0f : 0.01f;
        case LNR_ACT_EXPONENTIAL: return expf(v);
        case LNR_ACT_SIGMOID: { float s = 1.0f / (1.0f + expf(-v)); return s * (1.0f - s); }
        case LNR_ACT_SQUAREPLUS: { const float y = LNR_K_ACT * v; return 0.5f * (1.0f + y / sqrtf(y * y + 4.0f)); }
        case LNR_ACT_SOFTPLUS: return 1.0f / (1.0f + expf(-LNR_K_ACT * v));
        case LNR_ACT_TANH: { float t = tanhf(v); return 1.0f - t * t; }
        default: return 1.0f;
    }
}

// DecoupledNeRF.forward clips non-finite densities (nerf_tcnn.py:70-78: nan_to_num with the extremes of the NETWORK's dtype,
// NaN -> 0, and a warning the first time).  HALF = false: the fp32 network, limits +-FLT_MAX.  HALF = true (LNR_PREC_F16): the
// reference's network returns fp16, so whatever exceeds 65504 is +-inf there and comes back as +-65504.  `flag` (nullable) counts
// the clipped outputs; the host prints the reference's warning once when it finds it non-zero.
template <bool HALF>
__device__ __forceinline__ float finite_or_clipped(float v, int32_t* __restrict__ flag) {
    const float lim = HALF ? 65504.0f : 3.402823466e+38f;
    if (__builtin_expect(__builtin_fabsf(v) <= lim, 1)) return v;          // false for NaN
    if (flag != nullptr) atomicAdd(flag, 1);
    return v != v ? 0.0f : copysignf(lim, v);
}

#define MFMA4(acc, a4, b0, b1, b2, b3)                                      \
    do {                                                                    \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).x, b0, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).y, b1, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).z, b2, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).w, b3, acc, 0, 0, 0); \
    } while (0)

// hidden layer: Zn = Wl * act(Z)
template <int HT>
__device__ __forceinline__ void hidden_forward(const float* Wl, int H, int act, int c, int g, const f32x4 Z[HT], f32x4 Zn[HT], int row_stride = 0) {
    if (row_stride == 0) row_stride = H;               // floats between consecutive rows of Wl (padded copies in LDS: lnr_w_stride)
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) Zn[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kt = 0; kt < HT; ++kt) {
        const float a0 = act_fwd(Z[kt].x, act), a1 = act_fwd(Z[kt].y, act), a2 = act_fwd(Z[kt].z, act), a3 = act_fwd(Z[kt].w, act);
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) {
            const float4 wa = *reinterpret_cast<const float4*>(Wl + (16 * jt + c) * row_stride + 16 * kt + 4 * g);
            MFMA4(Zn[jt], wa, a0, a1, a2, a3);
        }
    }
}

// ================================================================================================
// MLP kernels on feature planes (level-major pipeline: lnr_encode.hip produces / consumes the planes)
//   feat [enc_dim][m_pad]   dfeat [enc_dim][m_pad]     padded inputs (k >= enc_dim) are the constant 1
// ================================================================================================
template <int HT, int KT>
__device__ __forceinline__ void layer1_from_planes(const LnrNetSpec& spec, const float* W1, const float* __restrict__ feat,
                                                   int64_t m_pad, int64_t m, int c, int g, f32x4 Z[HT], int row_stride = 0) {
    const int in_dim = KT > 0 ? 16 * KT : spec.in_dim;
    if (row_stride == 0) row_stride = in_dim;
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) Z[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kt = 0; kt < in_dim / 16; ++kt) {
        const int k0 = 16 * kt + 4 * g;
        float xf[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xf[r] = (k0 + r < spec.enc_dim) ? feat[(size_t)(k0 + r) * m_pad + m] : 1.0f;
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) {
            const float4 wa = *reinterpret_cast<const float4*>(W1 + (16 * jt + c) * row_stride + k0);
            MFMA4(Z[jt], wa, xf[0], xf[1], xf[2], xf[3]);
        }
    }
}

// LDS copies of the weight matrices are stored with rows 4 floats longer than the matrix: the MFMA A fragments are read as one float4
// per lane from 16 consecutive rows (lane & 15), and with row lengths of 16 k floats all 16 rows start in the same LDS bank - a 16-way
// conflict on every fragment (SQ_LDS_BANK_CONFLICT 17 x SQ_ACTIVE_INST_LDS in mlp_forward_kernel, frequency-12 -> 128 x 2: 33 % of the
// fp32 MFMA peak); 4 floats of padding spread them over all banks.  Layout: [W1: H rows of in_dim + 4][hidden matrices: H rows of H + 4
// each][output rows: 16 x H, unpadded].
// (lnr_w_stride / lnr_w_lds_floats: lnr_density_api.h, shared with the host-side launch plan)
// params (tinycudann layout) -> the padded copy; with_first = false: everything but the first layer's matrix
__device__ __forceinline__ void lnr_fill_w_lds(float* dst, const float* __restrict__ params, int H, int in_dim, int n_hidden, bool with_first) {
    const int n1 = H * in_dim, nh = (n_hidden - 1) * H * H;
    float* d = dst;
    if (with_first) {
        for (int i = threadIdx.x; i < n1; i += blockDim.x) d[(i / in_dim) * lnr_w_stride(in_dim) + i % in_dim] = params[i];
        d += H * lnr_w_stride(in_dim);
    }
    for (int i = threadIdx.x; i < nh; i += blockDim.x) d[(i / H) * lnr_w_stride(H) + i % H] = params[n1 + i];
    d += (n_hidden - 1) * H * lnr_w_stride(H);
    for (int i = threadIdx.x; i < 16 * H; i += blockDim.x) d[i] = params[n1 + nh + i];
}

// The lane's first-layer inputs of one 16-sample tile: features 16 kt + 4 g + r of sample m, for every K block.  They come from HBM
// (streamed feature planes), ~2 us away: loaded one STEP ahead (load_tile_inputs for step + 1 is issued before step's products).
template <int KT1M>
__device__ __forceinline__ void load_tile_inputs(const LnrNetSpec& spec, const float* __restrict__ feat, int64_t m_pad, int64_t m, int g, int kt1,
                                                 float xf[KT1M][4]) {
#pragma unroll
    for (int kt = 0; kt < KT1M; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * kt + 4 * g + r;
            const bool real = kt < kt1 && k < spec.enc_dim;
            const float v = feat[(size_t)(real ? k : 0) * m_pad + m];          // (unconditional load, clamped plane)
            xf[kt][r] = real ? v : 1.0f;                                        // the encoding's padding is the constant 1
        }
    }
}
// First layer from those registers; the weight fragments of a K block are requested together, then multiplied.
template <int HT, int KT1M>
__device__ __forceinline__ void layer1_from_regs(const float* W1, int c, int g, int kt1, int row_stride, const float xf[KT1M][4], f32x4 Z[HT]) {
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) Z[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kt = 0; kt < KT1M; ++kt) {
        if (kt < kt1) {
            float4 wa[HT];
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) wa[jt] = *reinterpret_cast<const float4*>(W1 + (16 * jt + c) * row_stride + 16 * kt + 4 * g);
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) MFMA4(Z[jt], wa[jt], xf[kt][0], xf[kt][1], xf[kt][2], xf[kt][3]);
        }
    }
}

// ACT >= 0: activation fixed at compile time (the runtime switch over nine activations, sinf/tanhf/... inlined at
// every use, costs ~4x the registers and code of the ReLU network the reference configures); ACT < 0: spec.activation.
template <int HT, bool W_LDS, int ACT>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK)
mlp_forward_kernel(const LnrNetSpec spec, const float* __restrict__ params, const float* __restrict__ feat, int64_t m_pad,
                   int64_t n_points, const int32_t* __restrict__ n_rays_dev, int n_rays, int n_samples, float* __restrict__ sigma, int32_t* __restrict__ clip_flag) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = 16 * HT;
    const int n_mlp = spec.n_mlp_params;
    const int nw = blockDim.x >> 6;
    if (W_LDS) lnr_fill_w_lds(smem, params, H, spec.in_dim, spec.n_hidden, true);
    __syncthreads();
    const int s1 = W_LDS ? lnr_w_stride(spec.in_dim) : spec.in_dim, sh = W_LDS ? lnr_w_stride(H) : H;       // row strides
    const float* W1 = W_LDS ? smem : params;
    const float* Wh = W1 + H * s1;
    const float* Wo = Wh + (spec.n_hidden - 1) * H * sh;
    const int act = ACT >= 0 ? ACT : spec.activation;
    const int64_t M = n_rays_dev ? (int64_t)lnr_live_rays(n_rays, n_rays_dev) * n_samples : n_points;
    if (M <= 0) return;
    const int64_t n_tiles = (M + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    // up to 128 (padded) inputs: the features of the NEXT tile are requested before this tile's products (they come from HBM, and at
    // one wave per SIMD nothing else covers the ~2 us)
    constexpr int KTP = 8;
    const int kt1 = spec.in_dim >> 4;
    const bool ahead = kt1 <= KTP && HT <= 8;          // (256 neurons: the 64 registers of staging cost more than the wait, 0.65 -> 0.72 ms)
    const int64_t tile_step = (int64_t)gridDim.x * nw;
    float xf_n[KTP][4];
    {
        const int64_t t0 = (int64_t)blockIdx.x * nw + wave;
        const int64_t m0 = t0 * 16 + c;
        if (ahead && t0 < n_tiles) load_tile_inputs<KTP>(spec, feat, m_pad, m0 < M ? m0 : M - 1, g, kt1, xf_n);
    }
    for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < n_tiles; tile += tile_step) {
        int64_t m = tile * 16 + c;
        const bool valid = m < M;
        if (!valid) m = M - 1;
        f32x4 Z[HT];
        if (ahead) {
            float xf[KTP][4];
#pragma unroll
            for (int kt = 0; kt < KTP; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) xf[kt][r] = xf_n[kt][r];
            const int64_t tn = tile + tile_step < n_tiles ? tile + tile_step : tile;
            const int64_t mn = tn * 16 + c;
            load_tile_inputs<KTP>(spec, feat, m_pad, mn < M ? mn : M - 1, g, kt1, xf_n);
            layer1_from_regs<HT, KTP>(W1, c, g, kt1, s1, xf, Z);
        } else {
            layer1_from_planes<HT, 0>(spec, W1, feat, m_pad, m, c, g, Z, s1);
        }
        for (int l = 1; l < spec.n_hidden; ++l) {
            f32x4 Zn[HT];
            hidden_forward<HT>(Wh + (l - 1) * H * sh, H, act, c, g, Z, Zn, sh);
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
        if (g == 0 && valid) sigma[m] = finite_or_clipped<false>(part, clip_flag);
    }
}

// Backward of the MLP on feature planes: weight gradients (slabs) and dfeat planes.  No gathers, no scatters.
// LDS map (floats): [W if W_LDS][dW][per-wave: T_dz [H*16] | if n_hidden>1: T_a [H*16] | zsave [n_hidden*H*16]]
// General shapes (any depth, width <= 256, any activation): weight gradients are summed into an LDS copy with LDS atomics -
// DW64: in 64-bit fixed point (integer LDS atomics run ~16x faster than float ones on CDNA4) when that copy fits,
// else in fp32.
template <bool DW64> struct DwAcc;
template <> struct DwAcc<true> {
    typedef long long type;
    static __device__ __forceinline__ void add(long long* p, float v) {
        atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)lnr_to_fix(v));
    }
    static __device__ __forceinline__ float get(long long v) { return (float)((double)v * (1.0 / (double)LNR_FIX_SCALE)); }
};
template <> struct DwAcc<false> {
    typedef float type;
    static __device__ __forceinline__ void add(float* p, float v) { atomicAdd(p, v); }
    static __device__ __forceinline__ float get(float v) { return v; }
};

template <int HT, bool W_LDS, int ACT, bool DW64>
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
    typedef typename DwAcc<DW64>::type dw_t;
    dw_t* dW = reinterpret_cast<dw_t*>(smem + (W_LDS ? n_mlp : 0));        // n_mlp is a multiple of 16: 8-byte aligned
    for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) { if (W_LDS) smem[i] = params[i]; dW[i] = (dw_t)0; }
    __syncthreads();
    const float* W = W_LDS ? smem : params;
    const float* W1 = W;
    const float* Wh = W + H * in_dim;
    const float* Wo = Wh + (NH - 1) * H * H;
    dw_t* dW1 = dW;
    dw_t* dWh = dW + H * in_dim;
    dw_t* dWo = dWh + (NH - 1) * H * H;
    const int act = ACT >= 0 ? ACT : spec.activation;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int scratch_per_wave = H * 16 + (NH > 1 ? (NH + 1) * H * 16 : 0);
    float* T_dz = reinterpret_cast<float*>(dW + n_mlp) + wave * scratch_per_wave;
    float* T_a = T_dz + H * 16;
    float* zsave = T_a + H * 16;

    f32x4 dWo_acc[HT];
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) dWo_acc[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

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
        layer1_from_planes<HT, 0>(spec, W1, feat, m_pad, m, c, g, Z);
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
            dw_t* dWl = (l == 0) ? dW1 : dWh + (l - 1) * H * H;
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
                for (int kt = 0; kt < K / 16; ++kt) {
                    const float4 b4 = load_b(kt);
#pragma unroll
                    for (int jt = 0; jt < HT; ++jt) {
                        const float4 a4 = *reinterpret_cast<const float4*>(T_dz + (16 * jt + c) * 16 + 4 * g);
                        f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        MFMA4(acc, a4, b4.x, b4.y, b4.z, b4.w);
#pragma unroll
                        for (int r = 0; r < 4; ++r) DwAcc<DW64>::add(dWl + (16 * jt + 4 * g + r) * K + 16 * kt + c, acc[r]);
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
                        for (int r = 0; r < 4; ++r) DwAcc<DW64>::add(dWl + (16 * jt + 4 * g + r) * K + 16 * kt + c, acc[r]);
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
#pragma unroll
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
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = dWo_acc[jt][r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (c == 0) DwAcc<DW64>::add(dWo + 16 * jt + 4 * g + r, v);
        }
    }
    __syncthreads();
    float* slab = slabs + (size_t)blockIdx.x * n_mlp;
    for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) slab[i] = DwAcc<DW64>::get(dW[i]);
}

// ================================================================================================
// Fast path for the reference's default shape class: 32 encoded features -> 16*HT ReLU neurons -> 1, weights in LDS.
// Everything is compile-time: the weight fragments a lane needs live in registers for the whole kernel, the next
// tile's inputs are loaded while the current tile runs through the MFMAs (two waves per SIMD cover the rest).
// Requires the feature planes to be readable (finite) up to the next multiple of 16 samples - the encode kernels
// zero that padding.
// ================================================================================================
#define LNR_TDZ_STRIDE 20   // floats per neuron row of the dZ transpose buffer: conflict-free 128-bit reads

struct Tile32 {
    float ds;
    float xf[2][4];     // features 16kt+4g+r of sample c        (B operand of layer 1)
    float4 xb[2];       // feature 16kt+c of samples 4g..4g+3    (B operand of the weight gradient)
};

// Addressing: plane base pointers are wave-uniform (SGPR pairs), the per-lane part is one 32-bit byte offset shared by
// all planes (global_load saddr+voffset form) - 64-bit per-plane pointers would cost ~40 VGPRs here.
// byte offset of element (plane 4g, sample tile*16+c) / (plane c, sample tile*16+4g); needs 17*m_pad*4 < 2^32
__device__ __forceinline__ uint32_t tile32_off_x(int64_t m_pad, int64_t tile, int c, int g) {
    return ((uint32_t)(4 * g) * (uint32_t)m_pad + (uint32_t)(tile * 16) + (uint32_t)c) * 4u;
}
__device__ __forceinline__ uint32_t tile32_off_b(int64_t m_pad, int64_t tile, int c, int g) {
    return ((uint32_t)c * (uint32_t)m_pad + (uint32_t)(tile * 16) + (uint32_t)(4 * g)) * 4u;
}
template <typename T>
__device__ __forceinline__ T ld_off(const float* base, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)byte_off);
}
__device__ __forceinline__ void st_off(float* base, uint32_t byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + (size_t)byte_off) = v;
}

template <bool BWD>
__device__ __forceinline__ void load_tile32(const float* __restrict__ feat, const float* __restrict__ d_sigma, int64_t m_pad, int64_t M,
                                            int64_t tile, int c, int g, Tile32& t) {
    const uint32_t ox = tile32_off_x(m_pad, tile, c, g), ob = tile32_off_b(m_pad, tile, c, g);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t.xf[kt][r] = ld_off<float>(feat + (size_t)(16 * kt + r) * m_pad, ox);
        if (BWD) t.xb[kt] = ld_off<float4>(feat + (size_t)(16 * kt) * m_pad, ob);
    }
    if (BWD) {
        const int64_t m = tile * 16 + c;
        const float v = d_sigma[m < M ? m : M - 1];      // unconditional load: a static number of loads in flight
        t.ds = m < M ? v : 0.0f;
    }
}

template <int HT>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK, 2)
mlp_forward_relu32_kernel(const LnrNetSpec spec, const float* __restrict__ params, const float* __restrict__ feat, int64_t m_pad,
                          int64_t n_points, const int32_t* __restrict__ n_rays_dev, int n_rays, int n_samples, float* __restrict__ sigma, int32_t* __restrict__ clip_flag) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 16 * HT;
    const int n_mlp = spec.n_mlp_params;
    const int nw = blockDim.x >> 6;
    for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) smem[i] = params[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    float4 wa[2][HT], wo[HT];
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) wa[kt][jt] = *reinterpret_cast<const float4*>(smem + (16 * jt + c) * 32 + 16 * kt + 4 * g);
        wo[jt] = *reinterpret_cast<const float4*>(smem + H * 32 + 16 * jt + 4 * g);
    }
    const int64_t M = n_rays_dev ? (int64_t)lnr_live_rays(n_rays, n_rays_dev) * n_samples : n_points;
    const int64_t n_tiles = M > 0 ? (M + 15) / 16 : 0;
    const int64_t stride = (int64_t)gridDim.x * nw;
    int64_t tile = (int64_t)blockIdx.x * nw + wave;
    if (tile >= n_tiles) return;
    // the inputs are requested TWO tiles ahead: 8 dword loads per tile and wave, 16 waves per CU - one tile ahead that is 8 MB in
    // flight on the chip, ~4 TB/s at 2 us of loaded-HBM latency (Little), and the kernel streamed at 2.6 TB/s
    Tile32 cur, nxt;
    load_tile32<false>(feat, nullptr, m_pad, M, tile, c, g, cur);
    load_tile32<false>(feat, nullptr, m_pad, M, tile + stride < n_tiles ? tile + stride : tile, c, g, nxt);
    while (tile < n_tiles) {
        const int64_t nt = tile + stride, nt2 = nt + stride;
        Tile32 nx2;       // prefetch is unconditional (the last iterations reload their own tile): the compiler can then
        load_tile32<false>(feat, nullptr, m_pad, M, nt2 < n_tiles ? nt2 : tile, c, g, nx2);   // count the loads in flight
        __builtin_amdgcn_sched_barrier(0);          // (and the scheduler must not sink them below the products)
        float part = 0.0f;
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) {
            f32x4 Z = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) MFMA4(Z, wa[kt][jt], cur.xf[kt][0], cur.xf[kt][1], cur.xf[kt][2], cur.xf[kt][3]);
            part += wo[jt].x * fmaxf(Z.x, 0.0f) + wo[jt].y * fmaxf(Z.y, 0.0f) + wo[jt].z * fmaxf(Z.z, 0.0f) + wo[jt].w * fmaxf(Z.w, 0.0f);
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        const int64_t m = tile * 16 + c;
        if (g == 0 && m < M) sigma[m] = finite_or_clipped<false>(part, clip_flag);
        cur = nxt;
        nxt = nx2;
        tile = nt;
    }
}

// LDS map (floats): [W n_mlp][dW n_mlp][per wave: T_dz H*LNR_TDZ_STRIDE]
template <int HT>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK, 2)
mlp_backward_relu32_kernel(const LnrNetSpec spec, const float* __restrict__ params, const float* __restrict__ feat, int64_t m_pad,
                           int64_t n_points, const int32_t* __restrict__ n_rays_dev, int n_rays, int n_samples,
                           const float* __restrict__ d_sigma, float* __restrict__ dfeat, float* __restrict__ slabs, int want_dfeat) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 16 * HT;
    const int n_mlp = spec.n_mlp_params;      // H*32 + H
    const int nw = blockDim.x >> 6;
    float* dW = smem + n_mlp;
    for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) { smem[i] = params[i]; dW[i] = 0.0f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    float* T_dz = dW + n_mlp + wave * (H * LNR_TDZ_STRIDE);

    float4 wa[2][HT], wo[HT];      // layer-1 A fragments, output weights of this lane's neurons: registers for the whole kernel
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) wa[kt][jt] = *reinterpret_cast<const float4*>(smem + (16 * jt + c) * 32 + 16 * kt + 4 * g);
        wo[jt] = *reinterpret_cast<const float4*>(smem + H * 32 + 16 * jt + 4 * g);
    }
    f32x4 dWo_acc[HT], dW1_acc[HT][2];
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) {
        dWo_acc[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        dW1_acc[jt][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        dW1_acc[jt][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }

    const int64_t M = n_rays_dev ? (int64_t)lnr_live_rays(n_rays, n_rays_dev) * n_samples : n_points;
    const int64_t n_tiles = M > 0 ? (M + 15) / 16 : 0;
    const int64_t stride = (int64_t)gridDim.x * nw;
    int64_t tile = (int64_t)blockIdx.x * nw + wave;
    Tile32 cur;
    if (tile < n_tiles) load_tile32<true>(feat, d_sigma, m_pad, M, tile, c, g, cur);
    while (tile < n_tiles) {
        const int64_t nt = tile + stride;
        Tile32 nxt;       // prefetch is unconditional (the last iteration reloads its own tile): the compiler can then
        load_tile32<true>(feat, d_sigma, m_pad, M, nt < n_tiles ? nt : tile, c, g, nxt);    // count the loads in flight
        // (requested two tiles ahead like the forward's: 0.271 -> 0.288 ms - 252 registers leave the allocator no slack)
        const int64_t m = tile * 16 + c;
        const bool valid = m < M;
        if (__ballot(cur.ds != 0.0f) == 0ull) {          // nothing flows back into this tile
            if (want_dfeat && valid) {
                const uint32_t ox = tile32_off_x(m_pad, tile, c, g);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st_off(dfeat + (size_t)(16 * kt + r) * m_pad, ox, 0.0f);
            }
        } else {
            f32x4 dZ[HT];
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) {
                f32x4 Z = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) MFMA4(Z, wa[kt][jt], cur.xf[kt][0], cur.xf[kt][1], cur.xf[kt][2], cur.xf[kt][3]);
                const float wv[4] = {wo[jt].x, wo[jt].y, wo[jt].z, wo[jt].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dWo_acc[jt][r] += cur.ds * fmaxf(Z[r], 0.0f);
                    dZ[jt][r] = Z[r] > 0.0f ? cur.ds * wv[r] : 0.0f;
                    T_dz[(16 * jt + 4 * g + r) * LNR_TDZ_STRIDE + c] = dZ[jt][r];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // dW1[neuron][feature] += sum over the tile's samples of dZ[neuron][s] * X[feature][s]
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) {
                const float4 a4 = *reinterpret_cast<const float4*>(T_dz + (16 * jt + c) * LNR_TDZ_STRIDE + 4 * g);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) MFMA4(dW1_acc[jt][kt], a4, cur.xb[kt].x, cur.xb[kt].y, cur.xb[kt].z, cur.xb[kt].w);
            }
            if (want_dfeat) {
                // A fragments of dX = W1^T dZ come from LDS every tile (the opaque offset keeps the compiler from hoisting
                // these 8*HT loads into registers the kernel does not have at two waves per SIMD)
                int woff = c;
                asm volatile("" : "+v"(woff));
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    f32x4 D = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            D = __builtin_amdgcn_mfma_f32_16x16x4f32(smem[(16 * jt + 4 * g + r) * 32 + 16 * kt + woff], dZ[jt][r], D, 0, 0, 0);
                    if (valid) {
                        const uint32_t ox = tile32_off_x(m_pad, tile, c, g);
#pragma unroll
                        for (int r = 0; r < 4; ++r) st_off(dfeat + (size_t)(16 * kt + r) * m_pad, ox, D[r]);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                // T_dz is rewritten by the next tile
        }
        cur = nxt;
        tile = nt;
    }
    // The waves add their register accumulators to the workgroup's LDS copy one after the other (every lane owns distinct
    // elements): plain read-modify-writes in a fixed order, so the slab - and the weight gradient - is bit-reproducible.
    float* dW1 = dW;
    float* dWo = dW + H * 32;
    for (int turn = 0; turn < nw; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dW1[(16 * jt + 4 * g + r) * 32 + 16 * kt + c] += dW1_acc[jt][kt][r];
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
