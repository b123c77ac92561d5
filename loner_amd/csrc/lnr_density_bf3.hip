// Density MLP of the reference's default shape class in the fp32 mode on the bf16 matrix pipe: every fp32 operand is split into
// THREE bf16 terms (x = x0 + x1 + x2 exactly: 8 + 8 + 8 significand bits, round-to-nearest at each step) and a product a*b becomes
// the six partial products a0b2 + a2b0 + a1b1 + a0b1 + a1b0 + a0b0 on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (gfx950).
//
// Why: v_mfma_f32_16x16x4_f32 - the exact-fma-chain instruction of lnr_density_impl.h - runs at 1/16 of the bf16 rate (32 cycles per
// SIMD for 1024 multiply-adds against ~17 for 8192), and the default network's MLP kernels sat at 57-60 % of THAT peak (0.10 + 0.27 ms
// of a 2.0 ms iteration, 5.7 ms of a rendered scan).  Six bf16 products cost 6/16 of one fp32 product; bf16 has fp32's exponent range,
// so no scaling is needed anywhere (the fp16 mode needs a power-of-two scale per tile).
// Accuracy: the three dropped products (a1b2, a2b1, a2b2) are below 2^-24 |a||b| (|x1| <= 2^-9 |x|, |x2| <= 2^-17 |x|), every kept
// product of two bf16 values is exact in fp32, sums accumulate in fp32 smallest terms first: on the default network's layer
// (numpy emulation, 32 x 64) the result is 4e-8 from the float64 product, the sequential fp32 fma chain 2e-7.  Operands with at most
// 16 significant bits (x2 = 0 and b2 = 0: every dropped product is zero) give the exact fp32 result.
// Non-finite operands: inf splits into (inf, NaN, NaN), so a non-finite weight or feature gives NaN where the fp32 chain could give
// +-inf; both are "not finite" to everything downstream (the clip of sigma: NaN -> 0, +-inf -> +-FLT_MAX; the pose-gradient guard).
//
// Replaces the fully-fused MLP half of the tinycudann NetworkWithInputEncoding the reference calls at src/models/nerf_tcnn.py:63-72
// (forward) and through loss.backward() (src/mapping/optimizer.py:366) for 32 encoded features -> 16*HT <= 64 ReLU neurons -> 1, fp32
// feature planes [feature][sample] (lnr_encode.hip).  Semantics = oracle/network.py (fp32).  spec.precision = LNR_PREC_F32 selects
// these kernels; LNR_PREC_F32_CHAIN keeps the exact-chain kernels of lnr_density_impl.h.
//
// Lane / fragment conventions as in lnr_density_f16.hip: lane = (c = lane & 15, g = lane >> 4); A operand: row c, K entries 8g..8g+7;
// B: column c, K entries 8g..8g+7; C/D: column c, rows 4g..4g+3.  The C layout of the first product is the B layout of the
// input-gradient product under the K permutation "slot 8g+i of a 32-neuron block = neuron 4g+i of its first 16-neuron tile (i < 4) or
// 4g+i-4 of its second".
#include "lnr_f16_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define BF3_TS 36            // floats per neuron row of the dZ transpose buffer (32 samples + pad: 144-byte rows, 16-byte aligned)
// A/B knobs (tagged development builds): workgroups of the forward per CU (160 registers allow three: 12 waves per CU keep more plane
// lines in flight), steps the backward requests its layer-1 operands ahead (2: 248 registers, still two waves per SIMD)
#ifndef LNR_BF3_FWD_OCC
#define LNR_BF3_FWD_OCC 3
#endif
#ifndef LNR_BF3_BWD_AHEAD
#define LNR_BF3_BWD_AHEAD 2
#endif

struct Frag3 { u32x4 t[3]; };        // the three bf16x8 terms of one MFMA operand fragment, as dwords

// (x, y) -> three packed bf16 pairs, x = x0 + x1 + x2 and y likewise, exactly (each residual is representable: Sterbenz)
__device__ __forceinline__ void bf3_split_pair(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = __builtin_bit_cast(uint32_t, bf16x2{(__bf16)x, (__bf16)y});
    asm volatile("" : "+v"(p0));                      // (the compiler otherwise converts x a second time to get the low half alone)
    float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xFFFF0000u);
    p1 = __builtin_bit_cast(uint32_t, bf16x2{(__bf16)rx, (__bf16)ry});
    asm volatile("" : "+v"(p1));
    rx -= __uint_as_float(p1 << 16);
    ry -= __uint_as_float(p1 & 0xFFFF0000u);
    p2 = __builtin_bit_cast(uint32_t, bf16x2{(__bf16)rx, (__bf16)ry});
}
__device__ __forceinline__ void bf3_split8(const float v[8], Frag3& f) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t p0, p1, p2;
        bf3_split_pair(v[2 * q], v[2 * q + 1], p0, p1, p2);
        f.t[0][q] = p0; f.t[1][q] = p1; f.t[2][q] = p2;
    }
}
#define BF3_M(a, b, acc) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0)
// the six kept term pairs (a_i, b_j) of a split product, smallest first; X(i, j) is expanded once per pair
#define BF3_FOR_TERMS(X) X(0, 2) X(2, 0) X(1, 1) X(0, 1) X(1, 0) X(0, 0)
__device__ __forceinline__ f32x4 bf3_mfma6(const Frag3& a, const Frag3& b, f32x4 acc) {
    acc = BF3_M(a.t[0], b.t[2], acc);
    acc = BF3_M(a.t[2], b.t[0], acc);
    acc = BF3_M(a.t[1], b.t[1], acc);
    acc = BF3_M(a.t[0], b.t[1], acc);
    acc = BF3_M(a.t[1], b.t[0], acc);
    acc = BF3_M(a.t[0], b.t[0], acc);
    return acc;
}

// Sample order inside a 32-sample step: column c of column tile t stands for sample 2c + t.  A lane's first-layer inputs of BOTH
// tiles - features 8g .. 8g+7 (fb = plane 8g) of samples base + 2c and base + 2c + 1 - are then eight 8-byte loads whose 16 lanes cover
// one whole 128-byte line per plane (one sample per lane and instruction asked for half a line: the planes streamed at 3.5 TB/s);
// sigma, d_sigma and the d_feature planes move as pairs the same way.  The planes are zero-filled up to the next multiple of 32
// samples (encode_forward_kernel), so a ragged last step needs no clamping.
__device__ __forceinline__ void bf3_load_x2(const float* __restrict__ fb, uint32_t plane_bytes, uint32_t m_even, float x[2][8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float2 v = ld32<float2>(fb, (uint32_t)i * plane_bytes + m_even * 4u);
        x[0][i] = v.x; x[1][i] = v.y;
    }
}

template <int HT>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK, LNR_BF3_FWD_OCC)
mlp_forward_bf3_kernel(const float* __restrict__ params, const float* __restrict__ feat, int64_t m_pad, int64_t n_points,
                       const int32_t* __restrict__ n_rays_dev, int n_rays, int n_samples, float* __restrict__ sigma, int32_t* __restrict__ clip_flag) {
    constexpr int H = 16 * HT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    Frag3 wa[HT]; float wo[HT][4];           // layer-1 A fragments of the lane's rows (split once), the output row: registers for the whole kernel
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = params[(16 * jt + c) * 32 + 8 * g + i];
        bf3_split8(v, wa[jt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) wo[jt][r] = params[H * 32 + 16 * jt + 4 * g + r];
    }
    const int64_t M = live_samples(n_points, n_rays_dev, n_rays, n_samples);
    const int64_t n_tiles = M > 0 ? (M + 31) / 32 : 0;            // a step = 32 samples = two 16-column tiles
    const int64_t stride = (int64_t)gridDim.x * nw;
    int64_t tile = (int64_t)blockIdx.x * nw + wave;
    if (tile >= n_tiles) return;
    const float* fb = feat + (size_t)(8 * g) * m_pad;
    const uint32_t plane_bytes = (uint32_t)m_pad * 4u;
    // the inputs are requested TWO steps ahead (as mlp_forward_relu32_kernel: one ahead left the planes streaming at 2.6 TB/s)
    float cur[2][8], nxt[2][8];
    bf3_load_x2(fb, plane_bytes, (uint32_t)(tile * 32 + 2 * c), cur);
    bf3_load_x2(fb, plane_bytes, (uint32_t)((tile + stride < n_tiles ? tile + stride : tile) * 32 + 2 * c), nxt);
    while (tile < n_tiles) {
        const int64_t nt = tile + stride, nt2 = nt + stride;
        float nx2[2][8];                        // (unconditional prefetch: a static number of loads in flight)
        bf3_load_x2(fb, plane_bytes, (uint32_t)((nt2 < n_tiles ? nt2 : tile) * 32 + 2 * c), nx2);
        __builtin_amdgcn_sched_barrier(0);      // (the scheduler must not sink the prefetch below the products)
        Frag3 xb0, xb1;
        bf3_split8(cur[0], xb0);
        bf3_split8(cur[1], xb1);
        f32x4 Z0[HT], Z1[HT];
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) { Z0[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; Z1[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
        // term by term across the 2 HT independent accumulators: a dependent MFMA never follows its producer directly
#define BF3_X(IA, IB)                                                                          \
    _Pragma("unroll") for (int jt = 0; jt < HT; ++jt) { Z0[jt] = BF3_M(wa[jt].t[IA], xb0.t[IB], Z0[jt]); Z1[jt] = BF3_M(wa[jt].t[IA], xb1.t[IB], Z1[jt]); }
        BF3_FOR_TERMS(BF3_X)
#undef BF3_X
        float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
        for (int jt = 0; jt < HT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { p0 += wo[jt][r] * fmaxf(Z0[jt][r], 0.0f); p1 += wo[jt][r] * fmaxf(Z1[jt][r], 0.0f); }
        p0 += __shfl_xor(p0, 16, 64); p1 += __shfl_xor(p1, 16, 64);
        p0 += __shfl_xor(p0, 32, 64); p1 += __shfl_xor(p1, 32, 64);
        const int64_t m = tile * 32 + 2 * c;
        if (g == 0 && m < M) {
            p0 = finite_or_clipped<false>(p0, clip_flag);
            if (m + 1 < M) { p1 = finite_or_clipped<false>(p1, clip_flag); *reinterpret_cast<float2*>(sigma + m) = make_float2(p0, p1); }
            else sigma[m] = p0;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) { cur[t][i] = nxt[t][i]; nxt[t][i] = nx2[t][i]; }
        tile = nt;
    }
}

// LDS (bytes): [dW: n_mlp floats][per wave: T  H x BF3_TS floats][WA: 3 x HT x 64 lanes x 16][WT: 3 x 2 x KB x 64 lanes x 16]
template <int HT>
struct Bf3Lds {
    static constexpr int H = 16 * HT, KB = (HT + 1) / 2, NW = LNR_DENSITY_BLOCK / 64;
    static constexpr size_t N_MLP = (size_t)H * 32 + 16 * H;
    static constexpr size_t OFF_T = N_MLP * sizeof(float);
    static constexpr size_t OFF_WA = OFF_T + (size_t)NW * H * BF3_TS * sizeof(float);
    static constexpr size_t OFF_WT = OFF_WA + (size_t)3 * HT * 64 * 16;
    static constexpr size_t BYTES = OFF_WT + (size_t)3 * 2 * KB * 64 * 16;
};

template <int HT>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK, 2)
mlp_backward_bf3_kernel(const float* __restrict__ params, int n_mlp, const float* __restrict__ feat, int64_t m_pad, int64_t n_points,
                        const int32_t* __restrict__ n_rays_dev, int n_rays, int n_samples, const float* __restrict__ d_sigma,
                        float* __restrict__ dfeat, float* __restrict__ slabs, int want_dfeat) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = Bf3Lds<HT>;
    constexpr int H = 16 * HT, KB = L::KB;
    const int nw = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    float* dW = smem;
    float* T = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + L::OFF_T) + wave * (H * BF3_TS);
    u32x4* WA = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(smem) + L::OFF_WA);       // [term][jt][lane]
    u32x4* WT = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(smem) + L::OFF_WT);       // [term][it][kb][lane]
    for (int i = threadIdx.x; i < n_mlp; i += blockDim.x) dW[i] = 0.0f;
    // The operand fragments of W1 are the same for every wave and every step: split once per workgroup, re-read from LDS per use
    // (as registers - 48 + 48 of them at HT = 4 - they would not fit beside the accumulators at two waves per SIMD).
    //   WA: A fragments of Z = W1 X        row = neuron 16jt + c,  K entries = features 8g .. 8g+7
    //   WT: A fragments of dX = W1^T dZ    row = feature 16it + c, K slot 8g+i = neuron 32kb + (i < 4 ? 4g+i : 16 + 4g+i-4)
    for (int job = wave; job < HT + 2 * KB; job += nw) {
        float v[8];
        if (job < HT) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = params[(16 * job + c) * 32 + 8 * g + i];
        } else {
            const int it = (job - HT) / KB, kb = (job - HT) % KB;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int n = 32 * kb + (i < 4 ? 4 * g + i : 16 + 4 * g + (i - 4));
                v[i] = n < H ? params[n * 32 + 16 * it + c] : 0.0f;
            }
        }
        Frag3 f;
        bf3_split8(v, f);
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            if (job < HT) WA[(t3 * HT + job) * 64 + lane] = f.t[t3];
            else WT[(t3 * 2 * KB + (job - HT)) * 64 + lane] = f.t[t3];
        }
    }
    float wo[HT][4];
#pragma unroll
    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) wo[jt][r] = params[H * 32 + 16 * jt + 4 * g + r];
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

    const int64_t M = live_samples(n_points, n_rays_dev, n_rays, n_samples);
    const int64_t n_tiles = M > 0 ? (M + 31) / 32 : 0;            // a step = 32 samples = two 16-column tiles
    const int64_t stride = (int64_t)gridDim.x * nw;
    const uint32_t plane_bytes = (uint32_t)m_pad * 4u;
    const float* fb = feat + (size_t)(8 * g) * m_pad;              // layer-1 B operand: planes 8g .. 8g+7
    const float* fc = feat + (size_t)c * m_pad;                    // weight-gradient B operand: plane 16it + c
    float* db = dfeat + (size_t)(4 * g) * m_pad;                   // d_feature rows 16it + 4g + r
    int64_t tile = (int64_t)blockIdx.x * nw + wave;
    float xcur[2][8], dcur[2];
    // column c of tile t = sample 2c + t of the step (see bf3_load_x2); d_sigma is not padded: its pair is read element by element
    auto load_front = [&](int64_t tl, float (&x)[2][8], float (&ds)[2]) {
        const int64_t m = tl * 32 + 2 * c;
        bf3_load_x2(fb, plane_bytes, (uint32_t)m, x);
        const float v0 = d_sigma[m < M ? m : M - 1], v1 = d_sigma[m + 1 < M ? m + 1 : M - 1];
        ds[0] = m < M ? v0 : 0.0f;
        ds[1] = m + 1 < M ? v1 : 0.0f;
    };
#if LNR_BF3_BWD_AHEAD >= 2
    float xnxt[2][8], dnxt[2];
    if (tile < n_tiles) { load_front(tile, xcur, dcur); load_front(tile + stride < n_tiles ? tile + stride : tile, xnxt, dnxt); }
#else
    if (tile < n_tiles) load_front(tile, xcur, dcur);
#endif
    while (tile < n_tiles) {
        const int64_t nt = tile + stride;
#if LNR_BF3_BWD_AHEAD >= 2
        const int64_t nt2 = nt + stride;
        float xnx2[2][8], dnx2[2];
        load_front(nt2 < n_tiles ? nt2 : tile, xnx2, dnx2);        // unconditional prefetch of the layer-1 operands, two steps ahead
#else
        float xnxt[2][8], dnxt[2];
        load_front(nt < n_tiles ? nt : tile, xnxt, dnxt);          // unconditional prefetch of the next step's layer-1 operands
#endif
        const bool any = (dcur[0] != 0.0f) | (dcur[1] != 0.0f);
        if (__ballot(any) == 0ull) {                               // nothing flows back into this step
            if (want_dfeat) {           // (the d_feature planes are padded like the feature planes: whole pairs are stored)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    st32<float2>(db, (uint32_t)(16 * (k >> 2) + (k & 3)) * plane_bytes + (uint32_t)(tile * 32 + 2 * c) * 4u, make_float2(0.0f, 0.0f));
            }
        } else {
            // this step's weight-gradient B operand: samples tile*32 + 8g .. +7 of plane 16it + c (the planes are zero-filled up to the
            // next multiple of 32 samples); requested now, used after the products below
            float xraw[2][8];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const uint32_t off = (uint32_t)(16 * it) * plane_bytes + (uint32_t)(tile * 32 + 8 * g) * 4u;
                const float4 a = ld32<float4>(fc, off), b = ld32<float4>(fc, off + 16u);
                xraw[it][0] = a.x; xraw[it][1] = a.y; xraw[it][2] = a.z; xraw[it][3] = a.w;
                xraw[it][4] = b.x; xraw[it][5] = b.y; xraw[it][6] = b.z; xraw[it][7] = b.w;
            }
            Frag3 xf[2];
            bf3_split8(xcur[0], xf[0]);
            bf3_split8(xcur[1], xf[1]);
            int wl = lane;
            asm volatile("" : "+v"(wl));                  // opaque: keeps the compiler from hoisting the LDS fragment reads out of the loop
            float dz[HT][2][4];                           // dZ of the lane's neurons 16jt + 4g + r at samples 16t + c
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) {
                Frag3 wa;
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) wa.t[t3] = WA[(t3 * HT + jt) * 64 + wl];
                f32x4 Z0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, Z1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#define BF3_X(IA, IB) Z0 = BF3_M(wa.t[IA], xf[0].t[IB], Z0); Z1 = BF3_M(wa.t[IA], xf[1].t[IB], Z1);
                BF3_FOR_TERMS(BF3_X)
#undef BF3_X
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dWo_acc[jt][r] += dcur[0] * fmaxf(Z0[r], 0.0f) + dcur[1] * fmaxf(Z1[r], 0.0f);
                    dz[jt][0][r] = Z0[r] > 0.0f ? dcur[0] * wo[jt][r] : 0.0f;
                    dz[jt][1][r] = Z1[r] > 0.0f ? dcur[1] * wo[jt][r] : 0.0f;
                    *reinterpret_cast<float2*>(T + (16 * jt + 4 * g + r) * BF3_TS + 2 * c) = make_float2(dz[jt][0][r], dz[jt][1][r]);   // samples 2c, 2c + 1
                }
            }
            // The weight gradient's B operand is taken over (split into its bf16 terms) HERE, in front of the d_feature stores: loads
            // and stores share one in-order counter (vmcnt) and the stores sit in a branch the compiler cannot count, so a wait for
            // these loads placed behind them was a wait for the eight stores' acknowledgement (profiles/r06_fp16_mlp_phases.txt has
            // the finding for the fp16 kernels; here two waves per SIMD covered most of it: 0.209 -> 0.206 ms in the default bench).
            Frag3 xs[2];
            bf3_split8(xraw[0], xs[0]);
            bf3_split8(xraw[1], xs[1]);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) asm volatile("" : "+v"(xs[it].t[t3]));
            // dX = W1^T dZ: the lane's own dZ values are the B operand under the K permutation (slots 0-3: row tile 2kb, 4-7: 2kb+1)
            if (want_dfeat) {
                f32x4 D[2][2];                                // [t][it]
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    D[t][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; D[t][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        constexpr bool odd = (HT & 1) != 0;                      // HT == 1: the upper half of the K block is padding
                        float v[8];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[i] = dz[2 * kb][t][i];
                            v[4 + i] = (odd && 2 * kb + 1 >= HT) ? 0.0f : dz[(2 * kb + 1 < HT) ? 2 * kb + 1 : 2 * kb][t][i];
                        }
                        Frag3 b;
                        bf3_split8(v, b);
#pragma unroll
                        for (int it = 0; it < 2; ++it) {
                            Frag3 a;
#pragma unroll
                            for (int t3 = 0; t3 < 3; ++t3) a.t[t3] = WT[((t3 * 2 + it) * KB + kb) * 64 + wl];
                            D[t][it] = bf3_mfma6(a, b, D[t][it]);
                        }
                    }
                }
                // rows 16it + 4g + r of the d_feature planes, samples 2c and 2c + 1 of the step as one 8-byte store (samples past the end
                // carry dZ = 0 and land in the planes' padding)
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        st32<float2>(db, (uint32_t)(16 * it + r) * plane_bytes + (uint32_t)(tile * 32 + 2 * c) * 4u, make_float2(D[0][it][r], D[1][it][r]));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // dW1[neuron][feature] += sum over the 32 samples of dZ[neuron][s] X[feature][s]
#pragma unroll
            for (int jt = 0; jt < HT; ++jt) {
                const float4 a0 = *reinterpret_cast<const float4*>(T + (16 * jt + c) * BF3_TS + 8 * g);
                const float4 a1 = *reinterpret_cast<const float4*>(T + (16 * jt + c) * BF3_TS + 8 * g + 4);
                const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                Frag3 a;
                bf3_split8(v, a);
#define BF3_X(IA, IB) dW1_acc[jt][0] = BF3_M(a.t[IA], xs[0].t[IB], dW1_acc[jt][0]); dW1_acc[jt][1] = BF3_M(a.t[IA], xs[1].t[IB], dW1_acc[jt][1]);
                BF3_FOR_TERMS(BF3_X)
#undef BF3_X
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                // T is rewritten by the next step
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#if LNR_BF3_BWD_AHEAD >= 2
#pragma unroll
            for (int i = 0; i < 8; ++i) { xcur[t][i] = xnxt[t][i]; xnxt[t][i] = xnx2[t][i]; }
            dcur[t] = dnxt[t]; dnxt[t] = dnx2[t];
#else
#pragma unroll
            for (int i = 0; i < 8; ++i) xcur[t][i] = xnxt[t][i];
            dcur[t] = dnxt[t];
#endif
        }
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

// ------------------------------------------------------------------------------------------------ host side
bool lnr_bf3_class(const LnrNetSpec* spec, int64_t n_points) {
    // (32-bit byte offsets from a plane-group base reach plane 16 it + r + ... <= 31 of the PADDED planes: m_pad x 32 planes x 4 bytes must
    // stay below 2^32, and m_pad is the point count rounded up to 64 plus the 1088-float skew of make_layout - ADVICE r5)
    return spec->precision == LNR_PREC_F32 && spec->encoding == LNR_ENC_HASHGRID && spec->activation == LNR_ACT_RELU && spec->n_hidden == 1 &&
           spec->in_dim == 32 && spec->enc_dim == 32 && spec->n_neurons <= 64 && (spec->n_neurons == 16 || spec->n_neurons == 32 || spec->n_neurons == 64) &&
           n_points <= (1ll << 25) - 2048;
}

int lnr_mlp_fwd_bf3(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, float* sigma, hipStream_t st) {
    const int64_t tiles = (pt->n_points + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > LNR_DENSITY_MAX_BLOCKS) blocks = LNR_DENSITY_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), block(LNR_DENSITY_BLOCK);
#define LNR_BF3_FWD(HT) hipLaunchKernelGGL(mlp_forward_bf3_kernel<HT>, grid, block, 0, st, params, feat, m_pad, pt->n_points, pt->n_rays_dev, pt->n_rays, pt->n_samples, sigma, pt->clip_flag)
    switch (spec->n_neurons / 16) {
        case 1: LNR_BF3_FWD(1); break;
        case 2: LNR_BF3_FWD(2); break;
        default: LNR_BF3_FWD(4); break;
    }
#undef LNR_BF3_FWD
    return LNR_OK;
}

// weight-gradient slabs lnr_mlp_bwd_bf3 writes for up to n_points points (one per workgroup)
int lnr_bf3_bwd_slabs(const LnrNetSpec* spec, int64_t n_points) {
    (void)spec;
    const int64_t tiles = (n_points + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > LNR_BWD_MAX_BLOCKS) blocks = LNR_BWD_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int lnr_mlp_bwd_bf3(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                    float* dfeat, float* slabs, int want_dfeat, int* n_slabs, hipStream_t st) {
    const int blocks = lnr_bf3_bwd_slabs(spec, pt->n_points);
    *n_slabs = blocks;
    const dim3 grid((unsigned)blocks), block(LNR_DENSITY_BLOCK);
#define LNR_BF3_BWD(HT)                                                                                                          \
    do {                                                                                                                         \
        int rc_ = f16_set_lds(mlp_backward_bf3_kernel<HT>, Bf3Lds<HT>::BYTES, "lnr_density_backward");                           \
        if (rc_) return rc_;                                                                                                     \
        hipLaunchKernelGGL(mlp_backward_bf3_kernel<HT>, grid, block, Bf3Lds<HT>::BYTES, st, params, spec->n_mlp_params, feat, m_pad, \
                           pt->n_points, pt->n_rays_dev, pt->n_rays, pt->n_samples, d_sigma, dfeat, slabs, want_dfeat);         \
    } while (0)
    switch (spec->n_neurons / 16) {
        case 1: LNR_BF3_BWD(1); break;
        case 2: LNR_BF3_BWD(2); break;
        default: LNR_BF3_BWD(4); break;
    }
#undef LNR_BF3_BWD
    return LNR_OK;
}

// ------------------------------------------------------------------------------------------------ layout / exactness self-test
// D = A(16x32) * B(32x16) through the split path against the exact product: operands are 12-bit integers times 2^-6 (a2 = b2 = 0,
// a1, b1 != 0 for most), so every partial product is exact and the six kept terms ARE the product; err = max |difference| (must be 0).
__global__ void selftest_mfma_bf3_kernel(float* out) {
    const int lane = threadIdx.x;
    const int c = lane & 15, g = lane >> 4;
    auto A = [](int i, int k) { return (float)((i * 397 + k * 1013) % 4093 - 2046) * 0.015625f; };
    auto B = [](int k, int j) { return (float)((k * 743 - j * 211) % 3571 - 1785) * 0.015625f; };
    float av[8], bv[8];
    for (int i = 0; i < 8; ++i) { av[i] = A(c, 8 * g + i); bv[i] = B(8 * g + i, c); }
    Frag3 a, b;
    bf3_split8(av, a);
    bf3_split8(bv, b);
    const f32x4 d = bf3_mfma6(a, b, f32x4{0.0f, 0.0f, 0.0f, 0.0f});
    float err = 0.0f;
    for (int r = 0; r < 4; ++r) {
        double ref = 0.0;
        for (int k = 0; k < 32; ++k) ref += (double)A(4 * g + r, k) * (double)B(k, c);
        err = fmaxf(err, fabsf((float)(ref - (double)d[r])));
    }
    // and the split itself: x0 + x1 + x2 == x for a value with all 24 significand bits in use
    const float x = 1.0f + (float)(lane * 2654435 % 8388607) * 1.1920929e-7f;
    uint32_t p0, p1, p2;
    bf3_split_pair(x, -x, p0, p1, p2);
    const float back = (__uint_as_float(p0 << 16) + __uint_as_float(p1 << 16)) + __uint_as_float(p2 << 16);
    const float backy = (__uint_as_float(p0 & 0xFFFF0000u) + __uint_as_float(p1 & 0xFFFF0000u)) + __uint_as_float(p2 & 0xFFFF0000u);
    if (back != x || backy != -x) err = 1e9f;
    err = wave_max(err);
    if (lane == 0) out[2] = err;
}

int lnr_selftest_mfma_bf3(float* out, hipStream_t st) {
    hipLaunchKernelGGL(selftest_mfma_bf3_kernel, dim3(1), dim3(64), 0, st, out);
    return LNR_OK;
}
