// Wide density networks - 256 neurons x 2..3 hidden layers - in both precisions (gfx950, v_mfma_f32_16x16x4_f32).
//
// The reference hands n_neurons / n_hidden_layers straight to tinycudann (src/models/nerf_tcnn.py:29-38, schema
// cfg/nerf_config/default_nerf_hash.yaml:20-31); the fused kernels of this library keep a network's weights and its whole weight
// gradient on chip (LDS / accumulator registers of ONE workgroup), which ends at one 256 x 256 matrix: 256 KB in fp32, more than the
// LDS and more than a workgroup's registers.  This file is the route for that shape class, LAYER BY LAYER through HBM:
//   * the samples are processed in chunks of LNR_WIDE_CHUNK; a chunk's pre-activations Z_l [256][chunk] of every hidden layer and its
//     dZ planes live in the workspace (0.5 - 0.7 GB), so the batch size does not bound the memory;
//   * forward of a layer: a wave owns two 16-sample tiles and all 256 rows; the workgroup stages the layer's weights through LDS one
//     16-input block at a time, double-buffered, while the inputs of the next block are in flight (wide_layer_fwd_kernel);
//   * the WEIGHT GRADIENT dW_l = dZ_l a_{l-1}^T is split over samples ACROSS workgroups (split-K): a workgroup owns a 64 x 64 tile of
//     dW and one of S sample ranges of the chunk, both operands are read straight from the [row][sample] planes as 16-byte loads
//     (4 consecutive samples = the K dimension of one MFMA: no transposition anywhere), the input operand shared through LDS, the
//     workgroups of a range on one XCD, accumulators in registers, one partial slab per sample range, folded in a fixed order
//     (wide_dw_kernel / wide_fold_kernel: deterministic, no atomics);
//   * back-propagation dA_{l-1} = W_l^T dZ_l as a forward layer with a transposed copy of the weights, made once per call, the
//     activation derivative applied on the way out (wide_dx_kernel; the first layer writes the d_feature planes).
// Precision: LNR_PREC_F32 / LNR_PREC_F32_CHAIN - exact fp32 fma chains; LNR_PREC_F16 (HALF) - the reference's storage model of
// oracle/network.py: weights and every layer's inputs rounded to fp16 where they are consumed, products and sums in fp32 (a product of
// two fp16 values is exact in fp32, so the fp32 MFMA computes what the f16 MFMA with fp32 accumulation computes: the forward layers use
// the f16 pipe - wide_layer_fwd_h_kernel - the backward kernels, whose dZ operand is fp32, the fp32 pipe), gradients straight
// through the rounding, in fp32.  Semantics = oracle/network.py.  The planes cross HBM once per layer and direction; measured times and
// the history of the kernels: DESIGN.md 4, profiles/r05_wide_networks.txt (256 x 2 at 2.1 M samples: 6.2 / 17.8 ms, ~60 TFLOP/s).
#include "lnr_f16_common.h"

#define LNR_WIDE_CHUNK 131072          // samples per chunk (a multiple of 64): 128 MB per [256][chunk] fp32 plane set
// sample ranges of a chunk in the weight-gradient kernel (partial slabs 1 .. S).  A workgroup walks its range tile by tile - one 16-byte
// load per operand, 16 MFMAs, repeat: latency-bound - so the parallelism has to come from the number of ranges: with 32 of them (512
// workgroups for a 256 x 256 matrix, 2 waves per SIMD) the kernel took 0.84 ms per chunk, 7 TFLOP/s (profiles/r05_wide_networks.txt)
#define LNR_WIDE_SPLITS 128
#ifndef LNR_WIDE_F16_BWD
#define LNR_WIDE_F16_BWD 1             /* fp16 mode: weight gradient and first-layer back-propagation on the f16 matrix pipe (0: round 5's fp32-MFMA kernels, A/B) */
#endif
// Back-propagation through a HIDDEN matrix on the f16 pipe: measured 292 us per chunk against 246 us for the fp32-MFMA kernel
// (profiles/r06_wide_networks.txt) - with the matrix time gone the kernel is its plane traffic (dZ and Z_prev in, dZ_prev out as 4-byte
// accesses of 64-byte row segments), and the fp32 kernel's epilogue carries less of it per MFMA.  Off; the kernel stays for the A/B.
#ifndef LNR_WIDE_F16_DX_HIDDEN
#define LNR_WIDE_F16_DX_HIDDEN 0
#endif
#define LNR_WIDE_H 256
#define WIDE_LDS_ROW 20                // floats per staged weight row of the forward: 16 + 4 padding (the 16 lanes of a 16-byte LDS read hit 16 distinct bank quads)
#define WIDE_LDS_TROW 260              // floats per staged row of a transposed matrix (back-propagation): 256 + 4 padding, same reason

enum { WIDE_IN_FEAT = 0, WIDE_IN_PAIR = 1, WIDE_IN_Z = 2 };
template <int N> struct WideInt { static constexpr int value = N; };

struct WideSamples {                   // which samples a launch works on
    int64_t lo, n;                     // chunk = samples [lo, lo + n) of the batch
    int64_t n_points; const int32_t* n_rays_dev; int32_t n_rays, n_samples;      // the batch's live sample count (live_samples)
};
__device__ __forceinline__ int64_t wide_live(const WideSamples& s) {
    const int64_t M = live_samples(s.n_points, s.n_rays_dev, s.n_rays, s.n_samples) - s.lo;     // live samples of this chunk
    return M < 0 ? 0 : (M < s.n ? M : s.n);
}

template <bool HALF> __device__ __forceinline__ float wide_w(float v) { return HALF ? round_f16(v) : v; }
template <bool HALF> __device__ __forceinline__ float4 wide_w4(float4 v) {
    if (HALF) { v.x = round_f16(v.x); v.y = round_f16(v.y); v.z = round_f16(v.z); v.w = round_f16(v.w); }
    return v;
}

// one input value of a layer: input k of batch sample m (chunk-local sample ml)
template <bool HALF, int IN>
__device__ __forceinline__ float wide_input(const float* __restrict__ in, int64_t stride, int k, int enc_dim, int64_t m, int64_t ml, int act) {
    if (IN == WIDE_IN_FEAT) return k < enc_dim ? in[(size_t)k * stride + m] : 1.0f;
    if (IN == WIDE_IN_PAIR) {
        if (k >= enc_dim) return 1.0f;
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 p = __builtin_bit_cast(h2, reinterpret_cast<const uint32_t*>(in)[(size_t)(k >> 1) * stride + m]);
        return (float)((k & 1) ? p.y : p.x);
    }
    const float a = act_fwd(in[(size_t)k * stride + ml], act);
    return HALF ? round_f16(a) : a;
}

// Order of the 16 LDS fragment reads and 128 MFMAs of one K block: fragment jt + 1 is requested before the eight MFMAs of fragment jt
// are issued (left alone the scheduler puts every read directly in front of its MFMAs, into the same registers: ~100 idle cycles of the
// matrix pipe per 256 busy ones with one wave per SIMD).  Mask 0x100 = LDS read, 0x008 = MFMA.
#define WIDE_PIPELINE_LDS_MFMA()                                   \
    do {                                                          \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        \
        _Pragma("unroll") for (int i_ = 0; i_ < 14; ++i_) {       \
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);    \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    \
        }                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);       \
    } while (0)

// wide_input in two halves for software prefetch: the load (issued a K block ahead) and what is computed from the loaded word
template <bool HALF, int IN>
__device__ __forceinline__ float wide_input_load(const float* __restrict__ in, int64_t stride, int k, int enc_dim, int64_t m, int64_t ml) {
    if (IN == WIDE_IN_FEAT) return k < enc_dim ? in[(size_t)k * stride + m] : 1.0f;
    if (IN == WIDE_IN_PAIR) return k < enc_dim ? in[(size_t)(k >> 1) * stride + m] : 0.0f;      // (the half2 pair as a bit pattern)
    return in[(size_t)k * stride + ml];
}
template <bool HALF, int IN>
__device__ __forceinline__ float wide_input_finish(float raw, int k, int enc_dim, int act) {
    if (IN == WIDE_IN_FEAT) return HALF ? round_f16(raw) : raw;
    if (IN == WIDE_IN_PAIR) {
        if (k >= enc_dim) return 1.0f;
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const uint32_t bits = __builtin_bit_cast(uint32_t, raw);
        const h2 p = __builtin_bit_cast(h2, bits);
        return (float)((k & 1) ? p.y : p.x);
    }
    const float a = act_fwd(raw, act);
    return HALF ? round_f16(a) : a;
}

// Z_out[j][ml] = sum_k W[j][k] in_k(ml): a wave owns TWO 16-sample tiles and all 256 rows (128 accumulator registers); the workgroup
// stages W[:, 16 kt .. 16 kt + 15] (16 KB) through LDS, double-buffered - block kt + 1 is in flight from L2 while block kt feeds the MFMAs,
// and the four waves share one copy.  (One tile per wave, fragments from L2: 14.5 ms per 2.1 M samples for 256 x 2; two tiles per wave:
// 11.4 ms, 0.47 ms per chunk and hidden layer = 4.3 x the MFMA time - every wave pulled the whole 256 KB matrix through L2 -> L1 itself.)
// The loop over tile pairs is workgroup-uniform (barriers inside): a wave beyond the last pair computes on clamped columns and stores nothing.
template <bool HALF, int IN>
__global__ void __launch_bounds__(256)
wide_layer_fwd_kernel(const float* __restrict__ W, int K, const float* __restrict__ in, int64_t in_stride, int enc_dim, int act,
                      WideSamples smp, float* __restrict__ z_out, int64_t chp) {
    __shared__ __attribute__((aligned(16))) float w_s[2][LNR_WIDE_H * WIDE_LDS_ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int64_t M = wide_live(smp);
    const int64_t n_tiles = (M + 15) / 16;
    const int64_t n_pairs = (n_tiles + 1) / 2;
    const int n_kt = K / 16;
    const float* wsrc = W + (size_t)threadIdx.x * K;               // staging: thread j copies row j's 16 floats of a block
    for (int64_t base = (int64_t)blockIdx.x * 4; base < n_pairs; base += (int64_t)gridDim.x * 4) {
        const int64_t pair = base + wave;
        int64_t ml[2], m[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ml[t] = (2 * pair + t) * 16 + c;
            if (ml[t] >= M) ml[t] = M - 1;                        // (finite operands for padding columns, a missing second tile, an idle wave)
            m[t] = smp.lo + ml[t];
        }
        f32x4 Z[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int jt = 0; jt < 16; ++jt) Z[t][jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        float4 s0 = *reinterpret_cast<const float4*>(wsrc), s1 = *reinterpret_cast<const float4*>(wsrc + 4),
               s2 = *reinterpret_cast<const float4*>(wsrc + 8), s3 = *reinterpret_cast<const float4*>(wsrc + 12);
        {                                                         // (every reader of buffer 0 is behind the barrier that ended its K block)
            float4* dst = reinterpret_cast<float4*>(&w_s[0][threadIdx.x * WIDE_LDS_ROW]);
            dst[0] = wide_w4<HALF>(s0); dst[1] = wide_w4<HALF>(s1); dst[2] = wide_w4<HALF>(s2); dst[3] = wide_w4<HALF>(s3);
        }
        float x[2][4], xr[2][4];                                  // this K block's inputs; the next block's, as loaded
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) xr[t][r] = wide_input_load<HALF, IN>(in, in_stride, 4 * g + r, enc_dim, m[t], ml[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[t][r] = wide_input_finish<HALF, IN>(xr[t][r], 4 * g + r, enc_dim, act);
        __syncthreads();
        for (int kt = 0; kt < n_kt; ++kt) {
            const bool more = kt + 1 < n_kt;
            if (more) {                                           // requests for block kt + 1 go out before this block's MFMAs
                const float* nx = wsrc + 16 * (kt + 1);
                s0 = *reinterpret_cast<const float4*>(nx); s1 = *reinterpret_cast<const float4*>(nx + 4);
                s2 = *reinterpret_cast<const float4*>(nx + 8); s3 = *reinterpret_cast<const float4*>(nx + 12);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xr[t][r] = wide_input_load<HALF, IN>(in, in_stride, 16 * (kt + 1) + 4 * g + r, enc_dim, m[t], ml[t]);
            }
            const float* wb = &w_s[kt & 1][c * WIDE_LDS_ROW + 4 * g];
            float4 wa = *reinterpret_cast<const float4*>(wb);     // W[16 jt + c][16 kt + 4 g ..] (rounded at staging in the fp16 mode), one fragment ahead
#pragma unroll
            for (int jt = 0; jt < 16; ++jt) {
                const float4 wn = *reinterpret_cast<const float4*>(wb + 16 * (jt < 15 ? jt + 1 : 15) * WIDE_LDS_ROW);
                MFMA4(Z[0][jt], wa, x[0][0], x[0][1], x[0][2], x[0][3]);
                MFMA4(Z[1][jt], wa, x[1][0], x[1][1], x[1][2], x[1][3]);
                wa = wn;
            }
            WIDE_PIPELINE_LDS_MFMA();
            if (more) {
                float4* dst = reinterpret_cast<float4*>(&w_s[(kt + 1) & 1][threadIdx.x * WIDE_LDS_ROW]);
                dst[0] = wide_w4<HALF>(s0); dst[1] = wide_w4<HALF>(s1); dst[2] = wide_w4<HALF>(s2); dst[3] = wide_w4<HALF>(s3);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[t][r] = wide_input_finish<HALF, IN>(xr[t][r], 16 * (kt + 1) + 4 * g + r, enc_dim, act);
            }
            __syncthreads();                                      // block kt + 1 is visible; nobody reads block kt any more
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (2 * pair + t >= n_tiles) continue;                // (whole tiles of the chunk planes are written, padding columns included)
            const int64_t col = (2 * pair + t) * 16 + c;
#pragma unroll
            for (int jt = 0; jt < 16; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) z_out[(size_t)(16 * jt + 4 * g + r) * chp + col] = Z[t][jt][r];
        }
    }
}

// The forward layer of the fp16 mode on the f16 matrix pipe (v_mfma_f32_16x16x32_f16: one instruction per 32 inputs where the fp32 pipe
// takes eight - the storage model's values either way: fp16 operands, fp32 accumulation).  Same tiling and staging as
// wide_layer_fwd_kernel; a K block is 32 inputs, the staged weights are fp16 (80-byte rows), the first layer's B operand is four dwords
// of the half2 pair planes as they are (lnr_density_f16.hip: K slot 8 g + i of a lane = input 32 kb + 8 g + i), a hidden layer's is
// act(Z) rounded and packed.  Inputs beyond in_dim (a last block of 16) meet zero weights.  IN: WIDE_IN_PAIR or WIDE_IN_Z.
#define WIDE_LDS_HROW 40               // halves per staged weight row: 32 + 8 padding
#define WIDE_PIPELINE_LDS_MFMA_H()                                 \
    do {                                                          \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        \
        _Pragma("unroll") for (int i_ = 0; i_ < 14; ++i_) {       \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    \
        }                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);        \
    } while (0)

template <int IN>
__global__ void __launch_bounds__(256)
wide_layer_fwd_h_kernel(const float* __restrict__ W, int K, const float* __restrict__ in, int64_t in_stride, int enc_dim, int act,
                        WideSamples smp, float* __restrict__ z_out, int64_t chp) {
    __shared__ __attribute__((aligned(16))) f16 w_h[2][LNR_WIDE_H * WIDE_LDS_HROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int64_t M = wide_live(smp);
    const int64_t n_tiles = (M + 15) / 16;
    const int64_t n_pairs = (n_tiles + 1) / 2;
    const int n_kb = (K + 31) / 32;
    const float* wsrc = W + (size_t)threadIdx.x * K;               // staging: thread j converts row j's 32 floats of a block
    const uint32_t ones2 = pack_h2(1.0f, 1.0f);
    for (int64_t base = (int64_t)blockIdx.x * 4; base < n_pairs; base += (int64_t)gridDim.x * 4) {
        const int64_t pair = base + wave;
        int64_t ml[2], m[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ml[t] = (2 * pair + t) * 16 + c;
            if (ml[t] >= M) ml[t] = M - 1;                        // (finite operands for padding columns, a missing second tile, an idle wave)
            m[t] = smp.lo + ml[t];
        }
        f32x4 Z[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int jt = 0; jt < 16; ++jt) Z[t][jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        float4 s0, s1, s2, s3, s4, s5, s6, s7;                    // the thread's 32 weights of the block in flight
        float xr[2][8];                                           // the lane's inputs of the block in flight, as loaded (pair planes: 4 dwords)
        f16x8 xb[2];
#define WIDE_H_STAGE_LOAD(kb_)                                                                                              \
        do {                                                                                                                \
            const int k0_ = 32 * (kb_);                                                                                     \
            const bool hi_ = k0_ + 16 < K;                       /* (in_dim is a multiple of 16: a block is whole or half) */ \
            const float* a_ = wsrc + k0_; const float* b_ = wsrc + (hi_ ? k0_ + 16 : k0_);                                  \
            s0 = *reinterpret_cast<const float4*>(a_); s1 = *reinterpret_cast<const float4*>(a_ + 4);                       \
            s2 = *reinterpret_cast<const float4*>(a_ + 8); s3 = *reinterpret_cast<const float4*>(a_ + 12);                  \
            s4 = *reinterpret_cast<const float4*>(b_); s5 = *reinterpret_cast<const float4*>(b_ + 4);                       \
            s6 = *reinterpret_cast<const float4*>(b_ + 8); s7 = *reinterpret_cast<const float4*>(b_ + 12);                  \
        } while (0)
#define WIDE_H_STAGE_STORE(buf_, kb_)       /* (the second half of a last half block is zeroed here, not at the load: no early wait) */ \
        do {                                                                                                                \
            if (!(32 * (kb_) + 16 < K)) { s4 = s5 = s6 = s7 = float4{0.0f, 0.0f, 0.0f, 0.0f}; }                             \
            u32x4* d_ = reinterpret_cast<u32x4*>(&w_h[buf_][threadIdx.x * WIDE_LDS_HROW]);                                  \
            d_[0] = u32x4{pack_h2(s0.x, s0.y), pack_h2(s0.z, s0.w), pack_h2(s1.x, s1.y), pack_h2(s1.z, s1.w)};              \
            d_[1] = u32x4{pack_h2(s2.x, s2.y), pack_h2(s2.z, s2.w), pack_h2(s3.x, s3.y), pack_h2(s3.z, s3.w)};              \
            d_[2] = u32x4{pack_h2(s4.x, s4.y), pack_h2(s4.z, s4.w), pack_h2(s5.x, s5.y), pack_h2(s5.z, s5.w)};              \
            d_[3] = u32x4{pack_h2(s6.x, s6.y), pack_h2(s6.z, s6.w), pack_h2(s7.x, s7.y), pack_h2(s7.z, s7.w)};              \
        } while (0)
        auto x_load = [&](int kb) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (IN == WIDE_IN_PAIR) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {                 // pair plane 16 kb + 4 g + q = inputs 32 kb + 8 g + 2 q, + 1
                        const int p = 16 * kb + 4 * g + q;
                        xr[t][q] = in[(size_t)(2 * p < enc_dim ? p : 0) * in_stride + m[t]];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) xr[t][r] = in[(size_t)(32 * kb + 8 * g + r) * in_stride + ml[t]];
                }
            }
        };
        auto x_finish = [&](int kb) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (IN == WIDE_IN_PAIR) {
                    uint32_t d[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) d[q] = 2 * (16 * kb + 4 * g + q) < enc_dim ? __builtin_bit_cast(uint32_t, xr[t][q]) : ones2;     // (padding inputs are the constant 1)
                    xb[t] = frag_from_dwords(d[0], d[1], d[2], d[3]);
                } else {
                    xb[t] = frag_from_dwords(pack_h2(act_fwd(xr[t][0], act), act_fwd(xr[t][1], act)), pack_h2(act_fwd(xr[t][2], act), act_fwd(xr[t][3], act)),
                                             pack_h2(act_fwd(xr[t][4], act), act_fwd(xr[t][5], act)), pack_h2(act_fwd(xr[t][6], act), act_fwd(xr[t][7], act)));
                }
            }
        };
        WIDE_H_STAGE_LOAD(0);
        x_load(0);
        WIDE_H_STAGE_STORE(0, 0);                                 // (every reader of buffer 0 is behind the barrier that ended its K block)
        x_finish(0);
        __syncthreads();
        for (int kb = 0; kb < n_kb; ++kb) {
            const bool more = kb + 1 < n_kb;
            if (more) { WIDE_H_STAGE_LOAD(kb + 1); x_load(kb + 1); }
            const f16* wb = &w_h[kb & 1][c * WIDE_LDS_HROW + 8 * g];
            f16x8 wa = *reinterpret_cast<const f16x8*>(wb);       // W[16 jt + c][32 kb + 8 g ..], one fragment ahead
#pragma unroll
            for (int jt = 0; jt < 16; ++jt) {
                const f16x8 wn = *reinterpret_cast<const f16x8*>(wb + 16 * (jt < 15 ? jt + 1 : 15) * WIDE_LDS_HROW);
                Z[0][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb[0], Z[0][jt], 0, 0, 0);
                Z[1][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb[1], Z[1][jt], 0, 0, 0);
                wa = wn;
            }
            WIDE_PIPELINE_LDS_MFMA_H();
            if (more) { WIDE_H_STAGE_STORE((kb + 1) & 1, kb + 1); x_finish(kb + 1); }
            __syncthreads();                                      // block kb + 1 is visible; nobody reads block kb any more
        }
#undef WIDE_H_STAGE_LOAD
#undef WIDE_H_STAGE_STORE
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (2 * pair + t >= n_tiles) continue;                // (whole tiles of the chunk planes are written, padding columns included)
            const int64_t col = (2 * pair + t) * 16 + c;
#pragma unroll
            for (int jt = 0; jt < 16; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) z_out[(size_t)(16 * jt + 4 * g + r) * chp + col] = Z[t][jt][r];
        }
    }
}

// sigma = w_out . act(Z_L) (the last hidden activation is NOT rounded in the storage model: oracle/network.py density_unit).  A workgroup
// owns 64 samples; wave w sums rows 64 w .. 64 w + 63 (an fma chain), the four partial sums are added in a fixed order.  (One thread per
// sample walking all 256 rows: 79 us per chunk, 512 workgroups with one dependent load chain per thread.)
template <bool HALF>
__global__ void __launch_bounds__(256)
wide_out_kernel(const float* __restrict__ wo, int act, const float* __restrict__ z, int64_t chp, WideSamples smp, float* __restrict__ sigma,
                int32_t* __restrict__ clip_flag) {
    __shared__ float w_s[LNR_WIDE_H];
    __shared__ float part[4][64];
    for (int i = threadIdx.x; i < LNR_WIDE_H; i += blockDim.x) w_s[i] = wide_w<HALF>(wo[i]);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t M = wide_live(smp);
    for (int64_t base = (int64_t)blockIdx.x * 64; base < M; base += (int64_t)gridDim.x * 64) {      // (workgroup-uniform: barriers inside)
        const int64_t ml = base + lane;
        float s = 0.0f;
        if (ml < M)
            for (int j = 64 * wave; j < 64 * wave + 64; ++j) s = __builtin_fmaf(w_s[j], act_fwd(z[(size_t)j * chp + ml], act), s);
        part[wave][lane] = s;
        __syncthreads();
        if (wave == 0 && ml < M) {
            const float t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
            sigma[smp.lo + ml] = HALF ? finite_or_clipped<true>(t, clip_flag) : finite_or_clipped<false>(t, clip_flag);
        }
        __syncthreads();
    }
}

// dZ_L[j][ml] = d_sigma[m] w_out[j] act'(Z_L[j][ml]); columns past the live count (up to the next multiple of 16) are zeroed: the
// weight-gradient kernel reads whole 16-sample tiles
template <bool HALF>
__global__ void __launch_bounds__(256)
wide_dz_out_kernel(const float* __restrict__ wo, int act, const float* __restrict__ z, int64_t chp, WideSamples smp,
                   const float* __restrict__ d_sigma, float* __restrict__ dz) {
    __shared__ float w_s[LNR_WIDE_H];
    for (int i = threadIdx.x; i < LNR_WIDE_H; i += blockDim.x) w_s[i] = wide_w<HALF>(wo[i]);
    __syncthreads();
    const int64_t M = wide_live(smp);
    const int64_t M16 = (M + 15) / 16 * 16;
    const int j0 = blockIdx.y * (LNR_WIDE_H / gridDim.y), j1 = j0 + LNR_WIDE_H / gridDim.y;     // (a thread walks a quarter of the rows: 4 x the loads in flight)
    for (int64_t ml = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ml < M16; ml += (int64_t)gridDim.x * blockDim.x) {
        const float ds = ml < M ? d_sigma[smp.lo + ml] : 0.0f;
        for (int j = j0; j < j1; ++j)
            dz[(size_t)j * chp + ml] = ml < M ? ds * w_s[j] * act_bwd(z[(size_t)j * chp + ml], act) : 0.0f;
    }
}

// d w_out[j] += sum over the chunk of d_sigma[m] act(Z_L[j][ml]): one workgroup per row j (the same one in every chunk: the
// accumulation order is fixed)
__global__ void __launch_bounds__(1024)
wide_dwo_kernel(int act, const float* __restrict__ z, int64_t chp, WideSamples smp, const float* __restrict__ d_sigma, float* __restrict__ dwo) {
    __shared__ float part[16];                                    // (1024 threads: 256 workgroups of 4 waves left three quarters of the wave slots empty)
    const int j = blockIdx.x;
    const int64_t M = wide_live(smp);
    float s = 0.0f;
    for (int64_t ml = threadIdx.x; ml < M; ml += blockDim.x) s = __builtin_fmaf(d_sigma[smp.lo + ml], act_fwd(z[(size_t)j * chp + ml], act), s);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; ++w) t += part[w];                // fixed order
        dwo[j] += t;
    }
}

// partial[split][j][k] = sum over the split's sample tiles of dZ[j][s] in_k[s]: workgroup = (64-row block, 64-column block, split),
// wave w = row tile 4 rb + w, up to four 16-column tiles.  Both operands are 16-byte loads of 4 consecutive samples of one
// [row][sample] plane (the K dimension of one MFMA: no transposition).  The input operand - the same 64 columns x 16 samples for all four
// waves - is loaded once per workgroup (one 16-byte load per thread), finished (activation, fp16 rounding, pair extraction: once
// instead of four times) and shared through LDS, double-buffered, a tile ahead of the MFMAs; the dZ operand is the wave's own.
// The 4 n_cb workgroups of a split read the same dZ rows and input rows: the block index is decoded so that they share an XCD
// (workgroup i runs on XCD i mod 8), i.e. one L2 - with the plain (rb, cb, split) order a split's workgroups sat on all eight and
// every plane crossed the fabric 2 - 4 times.  (0.84 ms per chunk for a hidden matrix with 32 ranges, 0.46 with 128, 0.37 with the XCD
// order and per-wave operand prefetch.)
template <bool HALF, int IN>
__global__ void __launch_bounds__(256)
wide_dw_kernel(const float* __restrict__ dz, int64_t chp, const float* __restrict__ in, int64_t in_stride, int enc_dim, int K, int act,
               WideSamples smp, float* __restrict__ partial, int64_t n_mlp, int64_t layer_off) {
    __shared__ __attribute__((aligned(16))) float b_s[2][64 * WIDE_LDS_ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int n_cb = (K + 63) / 64, n_inner = 4 * n_cb;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;           // (LNR_WIDE_SPLITS is a multiple of 8)
    const int inner = q % n_inner, split = (q / n_inner) * 8 + xcd;
    const int rb = inner & 3, cb = inner >> 2;
    const int jt = 4 * rb + wave;
    const int64_t M = wide_live(smp);
    const int64_t n_tiles = (M + 15) / 16;
    const float* dz_row = dz + (size_t)(16 * jt + c) * chp + 4 * g;
    // the thread's share of the input operand: samples 4 sq .. 4 sq + 3 of column sc of the workgroup's 64
    const int sc = threadIdx.x >> 2, sq = threadIdx.x & 3;
    const int sk = 64 * cb + sc;
    const bool s_load = sk < K && (IN == WIDE_IN_Z || sk < enc_dim);          // (else: a constant-one padding input, or beyond a ragged last block)
    const float* s_src = in + (size_t)(IN == WIDE_IN_PAIR ? (sk >> 1) : sk) * in_stride + (IN == WIDE_IN_Z ? 0 : smp.lo) + 4 * sq;
    float* s_dst0 = &b_s[0][sc * WIDE_LDS_ROW + 4 * sq];
    // (the planes hold whole 16-sample tiles - zero-filled features, finite Z columns - and dZ of a padding sample is 0: nothing is clamped)
    auto finish1 = [&](float v) -> float {
        if (IN == WIDE_IN_FEAT) {
            if (!s_load) v = 1.0f;
            return HALF ? round_f16(v) : v;
        } else if (IN == WIDE_IN_PAIR) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const uint32_t w = __builtin_bit_cast(uint32_t, v);
            const h2 p = __builtin_bit_cast(h2, w);
            return s_load ? (float)((sk & 1) ? p.y : p.x) : 1.0f;
        }
        v = act_fwd(v, act);
        return HALF ? round_f16(v) : v;
    };
    auto finish = [&](float4 raw) -> float4 { return float4{finish1(raw.x), finish1(raw.y), finish1(raw.z), finish1(raw.w)}; };
    // the loop over the split's tiles, for a workgroup with NCT 16-column tiles (a compile-time count - and a loop body without
    // branches in front of its MFMAs: with a wave-uniform "skip this column tile" inside, every MFMA group waited for ALL outstanding
    // loads, the next tile's included).  The last iteration requests its own tile again and stages it into the idle buffer.
    auto run = [&](auto nct_tag) {
        constexpr int NCT = decltype(nct_tag)::value;
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float4 zero4 = float4{0.0f, 0.0f, 0.0f, 0.0f};
        float4 a_c = zero4, a_n = zero4, raw = zero4;
        int64_t tile = split;
        if (tile < n_tiles) {
            if (s_load) raw = *reinterpret_cast<const float4*>(s_src + tile * 16);
            a_c = *reinterpret_cast<const float4*>(dz_row + tile * 16);
            *reinterpret_cast<float4*>(s_dst0) = finish(raw);
        }
        __syncthreads();
        for (int it = 0; tile < n_tiles; tile += LNR_WIDE_SPLITS, ++it) {       // (workgroup-uniform trip count: barriers inside)
            const int64_t nt = tile + LNR_WIDE_SPLITS < n_tiles ? tile + LNR_WIDE_SPLITS : tile;
            if (s_load) raw = *reinterpret_cast<const float4*>(s_src + nt * 16);
            a_n = *reinterpret_cast<const float4*>(dz_row + nt * 16);
            const float* bb = &b_s[it & 1][c * WIDE_LDS_ROW + 4 * g];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const float4 b = *reinterpret_cast<const float4*>(bb + 16 * ct * WIDE_LDS_ROW);
                MFMA4(acc[ct], a_c, b.x, b.y, b.z, b.w);
            }
            *reinterpret_cast<float4*>(s_dst0 + ((it + 1) & 1) * 64 * WIDE_LDS_ROW) = finish(raw);
            a_c = a_n;
            __syncthreads();
        }
        float* out = partial + (size_t)(1 + split) * n_mlp + layer_off;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(16 * jt + 4 * g + r) * K + 64 * cb + 16 * ct + c] = acc[ct][r];
    };
    const int n_ct = (K - 64 * cb) / 16;                            // column tiles of this workgroup (a ragged last block has fewer than four)
    if (n_ct >= 4) run(WideInt<4>{});
    else if (n_ct == 3) run(WideInt<3>{});
    else if (n_ct == 2) run(WideInt<2>{});
    else run(WideInt<1>{});
}

// total[off + i] += sum over the splits of partial[1 + s][off + i], s ascending (fixed order)
__global__ void __launch_bounds__(256)
wide_fold_kernel(float* __restrict__ slabs, int64_t n_mlp, int64_t off, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = 0.0f;
    for (int sp = 0; sp < LNR_WIDE_SPLITS; ++sp) s += slabs[(size_t)(1 + sp) * n_mlp + off + i];
    slabs[off + i] += s;
}

// WT[k][j] = W[j][k] for one 256 x K matrix (K a multiple of 16; K / 16 x 16 workgroups, one 16 x 16 tile each; once per backward call)
__global__ void __launch_bounds__(256)
wide_transpose_kernel(const float* __restrict__ W, int K, float* __restrict__ WT) {
    __shared__ float tile[16][17];
    const int n_bx = K / 16;
    const int bx = blockIdx.x % n_bx, by = blockIdx.x / n_bx, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    tile[ty][tx] = W[(size_t)(16 * by + ty) * K + 16 * bx + tx];
    __syncthreads();
    WT[(size_t)(16 * bx + ty) * LNR_WIDE_H + 16 * by + tx] = tile[tx][ty];
}

// Back-propagation through a HIDDEN matrix, dZ_prev[k][ml] = act'(Z_prev[k][ml]) sum_j W[j][k] dZ[j][ml], as a forward layer with the
// TRANSPOSED weights: with WT [k][j] row-major the A fragments are 16-byte loads and the lane's dZ values are the B operands, i.e. the
// loop of wide_layer_fwd_kernel (two tiles per wave).  The first version read W itself - one 4-byte load per MFMA, 64 of them hoisted per K
// block: 488 registers, 1.85 ms per chunk = 9 TFLOP/s, 40 % of a 256 x 2 backward (profiles/r05_wide_networks.txt).
// TO_FEAT (the first layer, WT = W_1^T [in_dim][256]): the result is the d_feature planes - rows < enc_dim, live samples, no activation
// derivative.  (Its first version read W_1 untransposed, one tile per wave: 0.34 ms per chunk for 80 inputs, more than the hidden layer.)
template <bool HALF, bool TO_FEAT>
__global__ void __launch_bounds__(256)
wide_dx_kernel(const float* __restrict__ WT, int n_kt, const float* __restrict__ dz, int64_t chp, int act, WideSamples smp,
               const float* __restrict__ z_prev, float* __restrict__ out, int64_t out_stride, int enc_dim) {
    // WT rows 16 kt .. 16 kt + 15 (16 KB, contiguous) staged through LDS, double-buffered and shared by the four waves, as in
    // wide_layer_fwd_kernel; the loop over tile pairs is workgroup-uniform for the barriers
    __shared__ __attribute__((aligned(16))) float wt_s[2][16 * WIDE_LDS_TROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int64_t M = wide_live(smp);
    const int64_t n_tiles = (M + 15) / 16;
    const int64_t n_pairs = (n_tiles + 1) / 2;
    const int st_row = threadIdx.x >> 6, st_col = (4 * threadIdx.x) & 255;       // staging: element 1024 q + 4 tid of a block = row 4 q + tid / 64
    for (int64_t base = (int64_t)blockIdx.x * 4; base < n_pairs; base += (int64_t)gridDim.x * 4) {
        const bool active = base + wave < n_pairs;
        const int64_t pair = active ? base + wave : n_pairs - 1;                 // (an idle wave: the last pair again, result dropped)
        const bool have1 = 2 * pair + 1 < n_tiles;
        const int64_t ml0 = (2 * pair) * 16 + c, ml1 = have1 ? ml0 + 16 : ml0;           // (a missing second tile: the first one again, result dropped)
        float d[2][16][4];                                        // dZ rows 16jt + 4g + r of the lane's two samples: the B operands of every K block
#pragma unroll
        for (int jt = 0; jt < 16; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {                         // (whole tiles exist: padding columns are 0)
                d[0][jt][r] = dz[(size_t)(16 * jt + 4 * g + r) * chp + ml0];
                d[1][jt][r] = dz[(size_t)(16 * jt + 4 * g + r) * chp + ml1];
            }
        const float* wsrc = WT + 4 * threadIdx.x;
        float4 s0 = *reinterpret_cast<const float4*>(wsrc), s1 = *reinterpret_cast<const float4*>(wsrc + 1024),
               s2 = *reinterpret_cast<const float4*>(wsrc + 2048), s3 = *reinterpret_cast<const float4*>(wsrc + 3072);
        {
            float* dst = &wt_s[0][st_row * WIDE_LDS_TROW + st_col];
            *reinterpret_cast<float4*>(dst) = wide_w4<HALF>(s0); *reinterpret_cast<float4*>(dst + 4 * WIDE_LDS_TROW) = wide_w4<HALF>(s1);
            *reinterpret_cast<float4*>(dst + 8 * WIDE_LDS_TROW) = wide_w4<HALF>(s2); *reinterpret_cast<float4*>(dst + 12 * WIDE_LDS_TROW) = wide_w4<HALF>(s3);
        }
        __syncthreads();
#pragma unroll 1
        for (int kt = 0; kt < n_kt; ++kt) {                      // (rolled: one output row tile at a time keeps the epilogue's operands few)
            const bool more = kt + 1 < n_kt;
            if (more) {
                const float* nx = wsrc + (size_t)(kt + 1) * 4096;
                s0 = *reinterpret_cast<const float4*>(nx); s1 = *reinterpret_cast<const float4*>(nx + 1024);
                s2 = *reinterpret_cast<const float4*>(nx + 2048); s3 = *reinterpret_cast<const float4*>(nx + 3072);
            }
            float zp[2][4];                                       // Z_prev of this block's output rows: requested before the MFMAs, used behind them
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t row = (size_t)(16 * kt + 4 * g + r) * chp;
                zp[0][r] = TO_FEAT ? 0.0f : z_prev[row + ml0]; zp[1][r] = TO_FEAT ? 0.0f : z_prev[row + ml1];
            }
            f32x4 D0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, D1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const float* wb = &wt_s[kt & 1][c * WIDE_LDS_TROW + 4 * g];
            float4 wa = *reinterpret_cast<const float4*>(wb);     // WT[16 kt + c][16 jt + 4 g ..], one fragment ahead
#pragma unroll
            for (int jt = 0; jt < 16; ++jt) {
                const float4 wn = *reinterpret_cast<const float4*>(wb + 16 * (jt < 15 ? jt + 1 : 15));
                MFMA4(D0, wa, d[0][jt][0], d[0][jt][1], d[0][jt][2], d[0][jt][3]);
                MFMA4(D1, wa, d[1][jt][0], d[1][jt][1], d[1][jt][2], d[1][jt][3]);
                wa = wn;
            }
            WIDE_PIPELINE_LDS_MFMA();
            if (active) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * kt + 4 * g + r;
                    if (TO_FEAT) {
                        if (k < enc_dim && ml0 < M) out[(size_t)k * out_stride + smp.lo + ml0] = D0[r];
                        if (k < enc_dim && have1 && ml1 < M) out[(size_t)k * out_stride + smp.lo + ml1] = D1[r];
                    } else {
                        const size_t row = (size_t)k * chp;
                        out[row + ml0] = ml0 < M ? D0[r] * act_bwd(zp[0][r], act) : 0.0f;
                        if (have1) out[row + ml1] = ml1 < M ? D1[r] * act_bwd(zp[1][r], act) : 0.0f;
                    }
                }
            }
            if (more) {
                float* dst = &wt_s[(kt + 1) & 1][st_row * WIDE_LDS_TROW + st_col];
                *reinterpret_cast<float4*>(dst) = wide_w4<HALF>(s0); *reinterpret_cast<float4*>(dst + 4 * WIDE_LDS_TROW) = wide_w4<HALF>(s1);
                *reinterpret_cast<float4*>(dst + 8 * WIDE_LDS_TROW) = wide_w4<HALF>(s2); *reinterpret_cast<float4*>(dst + 12 * WIDE_LDS_TROW) = wide_w4<HALF>(s3);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ fp16 mode: the backward on the f16 matrix pipe
// Round 5's fp16 mode multiplied dZ - an fp32 operand - on the fp32 MFMA (15.9 ms for a 256 x 2 backward at 2.1 M samples, 2.7 % of the f16
// peak: VERDICT r5 weak #10).  Per chunk of 131 072 samples (rocprofv3, profiles/r06_wide_networks.txt; fp32 MFMA -> f16 MFMA): weight
// gradient of a hidden matrix 212 -> 180 us, of the first layer 111 -> 67 us, first-layer back-propagation 74 -> 63 us, hidden
// back-propagation 246 -> 292 us (kept on the fp32 pipe): 15.9 -> ~14.5 ms - these kernels were never matrix-bound, their planes
// cross L2 -> L1 four times (weight gradient: 1.07 GB per chunk and matrix at ~6 TB/s) or leave as 64-byte row segments.  Here dZ is converted to fp16 where it is consumed, scaled by an exact power of two per 32-sample tile -
// 2^-e, e = the exponent of the tile's largest |d_sigma|: every dZ column is d_sigma[m] times O(1) factors, so the scaled values sit
// around 1 whatever the loss magnitude, and what underflows fp16 lies 2^-14 below its tile's maximum - and the f16 MFMA results are
// un-scaled in fp32 (the same scheme as the fused 128-wide kernels, lnr_f16_bwd_kernel.h, per tile instead of per workgroup step).
// A tile whose d_sigma are all zero contributes nothing and is skipped.

// 2^-e for the 32 samples [ml0, ml0 + 32) of the chunk (lanes (c, g): every lane calls it; the result is wave-uniform); 0: all-zero tile.
// In two halves so that the load can be issued a tile ahead of its use (wide_dw_h_kernel: one exposed L2 round trip per tile otherwise).
__device__ __forceinline__ float wide_tile_dsigma(const float* __restrict__ d_sigma, const WideSamples& smp, int64_t ml0, int64_t M, int lane) {
    const int64_t ml = ml0 + (lane & 31);
    return ml < M ? __builtin_fabsf(d_sigma[smp.lo + ml]) : 0.0f;
}
__device__ __forceinline__ float wide_tile_scale_of(float v, float* inv) {
    v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
    v = fmaxf(v, __shfl_xor(v, 8, 64)); v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
    if (!(v > 0.0f) || !(v < 3.0e38f)) { *inv = 0.0f; return 0.0f; }        // nothing to propagate (or a non-finite d_sigma: left to the fp32 checks upstream)
    uint32_t be = (__float_as_uint(v) >> 23) & 0xFFu;
    be = be < 2u ? 2u : (be > 252u ? 252u : be);
    *inv = __uint_as_float(be << 23);                                        // 2^e
    return __uint_as_float((254u - be) << 23);                               // 2^-e
}
__device__ __forceinline__ float wide_tile_scale(const float* __restrict__ d_sigma, const WideSamples& smp, int64_t ml0, int64_t M, int lane, float* inv) {
    return wide_tile_scale_of(wide_tile_dsigma(d_sigma, smp, ml0, M, lane), inv);
}

// dZ_prev = act'(Z_prev) . (W^T dZ) for a hidden matrix (TO_FEAT: the d_feature planes of the first layer) on v_mfma_f32_16x16x32_f16:
// wide_layer_fwd_h_kernel's loop with the TRANSPOSED weights WT [n_rows][256] (fp16-rounded where staged) as the matrix and the scaled
// dZ columns of the wave's 32 samples as the B operands.  n_rt = output row tiles (16 for a hidden matrix, in_dim / 16 for the first).
template <bool TO_FEAT>
__global__ void __launch_bounds__(256)
wide_dx_h_kernel(const float* __restrict__ WT, int n_rt, const float* __restrict__ dz, int64_t chp, int act, WideSamples smp,
                 const float* __restrict__ d_sigma, const float* __restrict__ z_prev, float* __restrict__ out, int64_t out_stride, int enc_dim) {
    __shared__ __attribute__((aligned(16))) f16 w_h[2][LNR_WIDE_H * WIDE_LDS_HROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int64_t M = wide_live(smp);
    const int64_t n_tiles = (M + 15) / 16;
    const int64_t n_pairs = (n_tiles + 1) / 2;
    constexpr int K = LNR_WIDE_H, n_kb = K / 32;
    const bool w_row = (int)threadIdx.x < 16 * n_rt;              // staging: thread j converts row j of WT (rows beyond the matrix: zeros)
    const float* wsrc = WT + (size_t)(w_row ? threadIdx.x : 0) * K;
    for (int64_t base = (int64_t)blockIdx.x * 4; base < n_pairs; base += (int64_t)gridDim.x * 4) {
        const int64_t pair = base + wave;
        const bool active = pair < n_pairs;
        int64_t ml[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ml[t] = (2 * (active ? pair : n_pairs - 1) + t) * 16 + c;
            if (ml[t] >= M) ml[t] = M - 1;                        // (finite operands for padding columns; their results are not stored)
        }
        float inv_sc;
        const float sc = wide_tile_scale(d_sigma, smp, (active ? pair : n_pairs - 1) * 32, M, lane, &inv_sc);
        f32x4 Z[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int jt = 0; jt < 16; ++jt) Z[t][jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        float4 s0, s1, s2, s3, s4, s5, s6, s7;
        float xr[2][8];
        f16x8 xb[2];
        const float4 zero4 = float4{0.0f, 0.0f, 0.0f, 0.0f};
#define WIDE_DXH_STAGE_LOAD(kb_)                                                                                            \
        do {                                                                                                                \
            const float* a_ = wsrc + 32 * (kb_);                                                                            \
            s0 = *reinterpret_cast<const float4*>(a_); s1 = *reinterpret_cast<const float4*>(a_ + 4);                       \
            s2 = *reinterpret_cast<const float4*>(a_ + 8); s3 = *reinterpret_cast<const float4*>(a_ + 12);                  \
            s4 = *reinterpret_cast<const float4*>(a_ + 16); s5 = *reinterpret_cast<const float4*>(a_ + 20);                 \
            s6 = *reinterpret_cast<const float4*>(a_ + 24); s7 = *reinterpret_cast<const float4*>(a_ + 28);                 \
        } while (0)
#define WIDE_DXH_STAGE_STORE(buf_)                                                                                          \
        do {                                                                                                                \
            if (!w_row) { s0 = s1 = s2 = s3 = s4 = s5 = s6 = s7 = zero4; }                                                  \
            u32x4* d_ = reinterpret_cast<u32x4*>(&w_h[buf_][threadIdx.x * WIDE_LDS_HROW]);                                  \
            d_[0] = u32x4{pack_h2(s0.x, s0.y), pack_h2(s0.z, s0.w), pack_h2(s1.x, s1.y), pack_h2(s1.z, s1.w)};              \
            d_[1] = u32x4{pack_h2(s2.x, s2.y), pack_h2(s2.z, s2.w), pack_h2(s3.x, s3.y), pack_h2(s3.z, s3.w)};              \
            d_[2] = u32x4{pack_h2(s4.x, s4.y), pack_h2(s4.z, s4.w), pack_h2(s5.x, s5.y), pack_h2(s5.z, s5.w)};              \
            d_[3] = u32x4{pack_h2(s6.x, s6.y), pack_h2(s6.z, s6.w), pack_h2(s7.x, s7.y), pack_h2(s7.z, s7.w)};              \
        } while (0)
        auto x_load = [&](int kb) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 8; ++r) xr[t][r] = dz[(size_t)(32 * kb + 8 * g + r) * chp + ml[t]];
        };
        auto x_finish = [&]() {
#pragma unroll
            for (int t = 0; t < 2; ++t)
                xb[t] = frag_from_dwords(pack_h2(xr[t][0] * sc, xr[t][1] * sc), pack_h2(xr[t][2] * sc, xr[t][3] * sc),
                                         pack_h2(xr[t][4] * sc, xr[t][5] * sc), pack_h2(xr[t][6] * sc, xr[t][7] * sc));
        };
        WIDE_DXH_STAGE_LOAD(0);
        x_load(0);
        WIDE_DXH_STAGE_STORE(0);
        x_finish();
        __syncthreads();
        for (int kb = 0; kb < n_kb; ++kb) {
            const bool more = kb + 1 < n_kb;
            if (more) { WIDE_DXH_STAGE_LOAD(kb + 1); x_load(kb + 1); }
            const f16* wb = &w_h[kb & 1][c * WIDE_LDS_HROW + 8 * g];
            if constexpr (!TO_FEAT) {
                // a hidden matrix: all 16 row tiles, the loop of wide_layer_fwd_h_kernel - fragment jt + 1 requested in front of the MFMAs
                // of fragment jt (with a branch per row tile and every read directly in front of its MFMAs the first version of this
                // kernel took 317 us per chunk against the forward layer's 139)
                f16x8 wa = *reinterpret_cast<const f16x8*>(wb);
#pragma unroll
                for (int jt = 0; jt < 16; ++jt) {
                    const f16x8 wn = *reinterpret_cast<const f16x8*>(wb + 16 * (jt < 15 ? jt + 1 : 15) * WIDE_LDS_HROW);
                    Z[0][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb[0], Z[0][jt], 0, 0, 0);
                    Z[1][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb[1], Z[1][jt], 0, 0, 0);
                    wa = wn;
                }
                WIDE_PIPELINE_LDS_MFMA_H();
            } else {
#pragma unroll
                for (int jt = 0; jt < 16; ++jt) {
                    if (jt < n_rt) {                              // (workgroup-uniform: the first layer has in_dim / 16 row tiles)
                        const f16x8 wa = *reinterpret_cast<const f16x8*>(wb + 16 * jt * WIDE_LDS_HROW);
                        Z[0][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb[0], Z[0][jt], 0, 0, 0);
                        Z[1][jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb[1], Z[1][jt], 0, 0, 0);
                    }
                }
            }
            if (more) { WIDE_DXH_STAGE_STORE((kb + 1) & 1); x_finish(); }
            __syncthreads();
        }
#undef WIDE_DXH_STAGE_LOAD
#undef WIDE_DXH_STAGE_STORE
        if (!active) continue;                                    // (workgroup-uniform loop; an idle wave stores nothing)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (2 * pair + t >= n_tiles) continue;
            const int64_t col = (2 * pair + t) * 16 + c;
            const bool live = col < M;
            if constexpr (!TO_FEAT) {
                // Z_prev of four row tiles (16 loads) in flight at a time, in front of the 16 products and stores that use them
#pragma unroll
                for (int j0 = 0; j0 < 16; j0 += 4) {
                    float zp[4][4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) zp[jj][r] = z_prev[(size_t)(16 * (j0 + jj) + 4 * g + r) * chp + col];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            out[(size_t)(16 * (j0 + jj) + 4 * g + r) * chp + col] = live ? Z[t][j0 + jj][r] * inv_sc * act_bwd(zp[jj][r], act) : 0.0f;
                }
            } else {
#pragma unroll
                for (int jt = 0; jt < 16; ++jt) {
                    if (jt < n_rt) {                              // (no break: the loop must unroll for the accumulators to stay registers)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int k = 16 * jt + 4 * g + r;
                            if (k < enc_dim && live) out[(size_t)k * out_stride + smp.lo + col] = Z[t][jt][r] * inv_sc;
                        }
                    }
                }
            }
        }
    }
}

// partial[split][j][k] = sum over the split's 32-sample tiles of dZ[j][s] in_k[s] on v_mfma_f32_16x16x32_f16: the work decomposition of
// wide_dw_kernel (workgroup = 64 rows x 64 columns x one of LNR_WIDE_SPLITS strided tile sequences, wave = one row tile, the workgroups of a
// split on one XCD), with K = 32 samples per MFMA: the wave's dZ operand is two 16-byte loads of 8 consecutive samples of its row, scaled
// by the tile's 2^-e and rounded to fp16; the input operand - 64 columns x 32 samples, finished (activation / pair extraction) once per
// workgroup - is staged through LDS as fp16 rows; every tile's product starts from zero and is added to the fp32 accumulators with its 2^e.
template <int IN>
__global__ void __launch_bounds__(256)
wide_dw_h_kernel(const float* __restrict__ dz, int64_t chp, const float* __restrict__ in, int64_t in_stride, int enc_dim, int K, int act,
                 WideSamples smp, const float* __restrict__ d_sigma, float* __restrict__ partial, int64_t n_mlp, int64_t layer_off) {
    constexpr int ROW = 40;                                       // halves per staged column: 32 samples + 8 (80-byte rows: conflict-free 16-byte reads)
    __shared__ __attribute__((aligned(16))) f16 b_s[2][64 * ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int n_cb = (K + 63) / 64, n_inner = 4 * n_cb;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int inner = q % n_inner, split = (q / n_inner) * 8 + xcd;
    const int rb = inner & 3, cb = inner >> 2;
    const int jt = 4 * rb + wave;
    const int64_t M = wide_live(smp);
    const int64_t n_t32 = (M + 31) / 32;
    const float* dz_row = dz + (size_t)(16 * jt + c) * chp + 8 * g;
    // the thread's share of the input operand: samples 8 sq .. 8 sq + 7 of column sc of the workgroup's 64
    const int scol = threadIdx.x >> 2, sq = threadIdx.x & 3;
    const int sk = 64 * cb + scol;
    const bool s_load = sk < K && (IN == WIDE_IN_Z || sk < enc_dim);          // (else: a constant-one padding input, or beyond a ragged last block)
    const float* s_src = in + (size_t)(IN == WIDE_IN_PAIR ? (sk >> 1) : sk) * in_stride + (IN == WIDE_IN_Z ? 0 : smp.lo) + 8 * sq;
    auto finish1 = [&](float v, bool live) -> float {
        if (!live) return 0.0f;                                   // (samples past the live count: whatever the planes hold there)
        if (IN == WIDE_IN_PAIR) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 p = __builtin_bit_cast(h2, __builtin_bit_cast(uint32_t, v));
            return s_load ? (float)((sk & 1) ? p.y : p.x) : 1.0f;
        }
        return act_fwd(v, act);                                   // (rounded to fp16 by the pack below)
    };
    auto stage = [&](int buf, int64_t tile, const float4& r0, const float4& r1) {
        const int64_t m0 = tile * 32 + 8 * sq;
        u32x4 h;
        h.x = pack_h2(finish1(r0.x, m0 < M), finish1(r0.y, m0 + 1 < M)); h.y = pack_h2(finish1(r0.z, m0 + 2 < M), finish1(r0.w, m0 + 3 < M));
        h.z = pack_h2(finish1(r1.x, m0 + 4 < M), finish1(r1.y, m0 + 5 < M)); h.w = pack_h2(finish1(r1.z, m0 + 6 < M), finish1(r1.w, m0 + 7 < M));
        *reinterpret_cast<u32x4*>(&b_s[buf][scol * ROW + 8 * sq]) = h;
    };
    auto run = [&](auto nct_tag) {
        constexpr int NCT = decltype(nct_tag)::value;
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float4 zero4 = float4{0.0f, 0.0f, 0.0f, 0.0f};
        float4 a0 = zero4, a1 = zero4, n0 = zero4, n1 = zero4, r0 = zero4, r1 = zero4;
        float ds_c = 0.0f, ds_n = 0.0f;                           // the lane's |d_sigma| of the tile in hand / the next one (for the tile's scale)
        int64_t tile = split;
        if (tile < n_t32) {
            ds_c = wide_tile_dsigma(d_sigma, smp, tile * 32, M, lane);
            if (s_load) { r0 = *reinterpret_cast<const float4*>(s_src + tile * 32); r1 = *reinterpret_cast<const float4*>(s_src + tile * 32 + 4); }
            a0 = *reinterpret_cast<const float4*>(dz_row + tile * 32); a1 = *reinterpret_cast<const float4*>(dz_row + tile * 32 + 4);
            stage(0, tile, r0, r1);
        }
        __syncthreads();
        for (int it = 0; tile < n_t32; tile += LNR_WIDE_SPLITS, ++it) {         // (workgroup-uniform trip count: barriers inside)
            const int64_t nt = tile + LNR_WIDE_SPLITS < n_t32 ? tile + LNR_WIDE_SPLITS : tile;
            if (s_load) { r0 = *reinterpret_cast<const float4*>(s_src + nt * 32); r1 = *reinterpret_cast<const float4*>(s_src + nt * 32 + 4); }
            n0 = *reinterpret_cast<const float4*>(dz_row + nt * 32); n1 = *reinterpret_cast<const float4*>(dz_row + nt * 32 + 4);
            ds_n = wide_tile_dsigma(d_sigma, smp, nt * 32, M, lane);
            float inv_sc;
            const float sc = wide_tile_scale_of(ds_c, &inv_sc);
            if (sc != 0.0f) {                                     // (wave-uniform, and the same for the four waves: no barrier inside)
                const int64_t m0 = tile * 32 + 8 * g;
                auto dzv = [&](float v, int i) { return m0 + i < M ? v * sc : 0.0f; };
                const f16x8 a = frag_from_dwords(pack_h2(dzv(a0.x, 0), dzv(a0.y, 1)), pack_h2(dzv(a0.z, 2), dzv(a0.w, 3)),
                                                 pack_h2(dzv(a1.x, 4), dzv(a1.y, 5)), pack_h2(dzv(a1.z, 6), dzv(a1.w, 7)));
                const f16* bb = &b_s[it & 1][c * ROW + 8 * g];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const f16x8 b = *reinterpret_cast<const f16x8*>(bb + 16 * ct * ROW);
                    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ct][r] = __builtin_fmaf(d[r], inv_sc, acc[ct][r]);
                }
            }
            stage((it + 1) & 1, nt, r0, r1);
            a0 = n0; a1 = n1; ds_c = ds_n;
            __syncthreads();
        }
        float* out = partial + (size_t)(1 + split) * n_mlp + layer_off;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(16 * jt + 4 * g + r) * K + 64 * cb + 16 * ct + c] = acc[ct][r];
    };
    const int n_ct = (K - 64 * cb) / 16;
    if (n_ct >= 4) run(WideInt<4>{});
    else if (n_ct == 3) run(WideInt<3>{});
    else if (n_ct == 2) run(WideInt<2>{});
    else run(WideInt<1>{});
}

// ------------------------------------------------------------------------------------------------ host side
bool lnr_wide_class(const LnrNetSpec* spec) {
    if (spec->n_neurons != LNR_WIDE_H || spec->n_hidden < 1 || spec->n_hidden > 3 || spec->in_dim % 16 != 0) return false;
    const bool f16 = spec->precision == LNR_PREC_F16;
    if (f16 && ((spec->enc_dim & 1) || (spec->encoding == LNR_ENC_HASHGRID && (spec->n_features & 1)))) return false;    // pair planes
    // one hidden layer: the fused kernels (fp32: lnr_density_regs.h; fp16: lnr_f16_*_kernel.h unless its LDS budget says no, e.g. 80
    // inputs); two and three: always here
    return spec->n_hidden >= 2 || (f16 && !lnr_f16_supported(spec));
}

// samples per chunk plane for calls of up to n_points points: a small batch (rendering a few thousand points, tests) gets planes of its own
// size instead of (n_hidden + 2) x 128 MB; the launches use the same value as their plane stride
static int64_t wide_chunk_stride(int64_t n_points) {
    // (a multiple of 256: a workgroup of the layer kernels covers 128 samples, the weight-gradient kernel reads groups of four - whatever
    // a launch touches beyond the chunk's last sample stays inside the sample's own plane, as it did with full-size planes)
    const int64_t n = (n_points + 255) / 256 * 256;
    return n < 256 ? 256 : (n < LNR_WIDE_CHUNK ? n : LNR_WIDE_CHUNK);
}
// bytes of chunk planes behind the other workspace regions: Z of every hidden layer + two dZ buffers, [256][chunk stride] each
size_t lnr_wide_workspace(const LnrNetSpec* spec, int64_t n_points) {
    if (!lnr_wide_class(spec)) return 0;
    return (size_t)(spec->n_hidden + 2) * LNR_WIDE_H * (size_t)wide_chunk_stride(n_points) * sizeof(float);
}
int lnr_wide_slabs(void) { return 3 + LNR_WIDE_SPLITS; }      // slab 0, the partial slabs, the transposed hidden matrices

namespace {
struct WideCtx {
    const LnrNetSpec* spec; const float* params; const float* feat; int64_t m_pad; const MlpPoints* pt; float* planes; hipStream_t st;
    bool half; int H, NH, K1, act;
    int64_t chp;                                                                          // plane stride (samples): wide_chunk_stride of the call's capacity
    float* z(int l) const { return planes + (size_t)l * LNR_WIDE_H * (size_t)chp; }              // pre-activations of hidden layer l (0-based)
    float* dzbuf(int i) const { return planes + (size_t)(NH + i) * LNR_WIDE_H * (size_t)chp; }
    const float* W(int l) const { return l == 0 ? params : params + (size_t)H * K1 + (size_t)(l - 1) * H * H; }
    const float* Wo() const { return params + (size_t)H * K1 + (size_t)(NH - 1) * H * H; }
    WideSamples samples(int64_t lo, int64_t n) const { return WideSamples{lo, n, pt->n_points, pt->n_rays_dev, pt->n_rays, pt->n_samples}; }
};

template <bool HALF>
static void wide_forward_chunk(const WideCtx& c, const WideSamples& s) {
    const dim3 block(256);
    const int64_t pairs = ((s.n + 15) / 16 + 1) / 2;                 // a wave owns two 16-sample tiles
    const dim3 grid((unsigned)((pairs + 3) / 4 > 2048 ? 2048 : (pairs + 3) / 4));
    if (HALF) {                                                     // the f16 matrix pipe
        hipLaunchKernelGGL((wide_layer_fwd_h_kernel<WIDE_IN_PAIR>), grid, block, 0, c.st, c.W(0), c.K1, c.feat, c.m_pad, c.spec->enc_dim, c.act, s, c.z(0), c.chp);
        for (int l = 1; l < c.NH; ++l)
            hipLaunchKernelGGL((wide_layer_fwd_h_kernel<WIDE_IN_Z>), grid, block, 0, c.st, c.W(l), c.H, c.z(l - 1), c.chp, c.H, c.act, s, c.z(l), c.chp);
        return;
    }
    hipLaunchKernelGGL((wide_layer_fwd_kernel<false, WIDE_IN_FEAT>), grid, block, 0, c.st, c.W(0), c.K1, c.feat, c.m_pad, c.spec->enc_dim, c.act, s, c.z(0), c.chp);
    for (int l = 1; l < c.NH; ++l)
        hipLaunchKernelGGL((wide_layer_fwd_kernel<false, WIDE_IN_Z>), grid, block, 0, c.st, c.W(l), c.H, c.z(l - 1), c.chp, c.H, c.act, s, c.z(l), c.chp);
}

template <bool HALF>
static int wide_forward(const WideCtx& c, float* sigma) {
    for (int64_t lo = 0; lo < c.pt->n_points; lo += LNR_WIDE_CHUNK) {
        const int64_t n = c.pt->n_points - lo < LNR_WIDE_CHUNK ? c.pt->n_points - lo : LNR_WIDE_CHUNK;
        const WideSamples s = c.samples(lo, n);
        wide_forward_chunk<HALF>(c, s);
        hipLaunchKernelGGL(wide_out_kernel<HALF>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, c.st, c.Wo(), c.act, c.z(c.NH - 1), c.chp, s, sigma, c.pt->clip_flag);
    }
    return LNR_OK;
}

template <bool HALF>
static int wide_backward(const WideCtx& c, const float* d_sigma, float* dfeat, float* slabs, int want_dfeat, int want_dw) {
    const int64_t n_mlp = c.spec->n_mlp_params;
    if (hipMemsetAsync(slabs, 0, (size_t)n_mlp * sizeof(float), c.st) != hipSuccess) { lnr_set_error("lnr_density_backward: hipMemsetAsync failed"); return LNR_ERR_LAUNCH; }
    const dim3 block(256);
    const int64_t off_o = (int64_t)c.H * c.K1 + (int64_t)(c.NH - 1) * c.H * c.H;
    // transposed copies of the hidden matrices (back-propagation reads them as A fragments): behind the partial slabs
    float* wt = slabs + (size_t)(2 + LNR_WIDE_SPLITS) * n_mlp;
    for (int l = 1; l < c.NH; ++l)
        hipLaunchKernelGGL(wide_transpose_kernel, dim3(256), block, 0, c.st, c.W(l), c.H, wt + (size_t)(l - 1) * c.H * c.H);
    float* wt_first = wt + (size_t)(c.NH - 1) * c.H * c.H;         // W_1^T [in_dim][256], for the d_feature planes
    if (want_dfeat) hipLaunchKernelGGL(wide_transpose_kernel, dim3((unsigned)(c.K1 / 16 * 16)), block, 0, c.st, c.W(0), c.K1, wt_first);
    for (int64_t lo = 0; lo < c.pt->n_points; lo += LNR_WIDE_CHUNK) {
        const int64_t n = c.pt->n_points - lo < LNR_WIDE_CHUNK ? c.pt->n_points - lo : LNR_WIDE_CHUNK;
        const WideSamples s = c.samples(lo, n);
        const int64_t tiles = (n + 15) / 16;
        const int64_t pairs = (tiles + 1) / 2;                       // (wide_dx_kernel: two tiles per wave)
        const dim3 grid_p((unsigned)((pairs + 3) / 4 > 2048 ? 2048 : (pairs + 3) / 4));
        const dim3 grid_s((unsigned)((n + 255) / 256));
        wide_forward_chunk<HALF>(c, s);
        float* dz = c.dzbuf(0);
        float* dz_other = c.dzbuf(1);
        hipLaunchKernelGGL(wide_dz_out_kernel<HALF>, dim3(grid_s.x, 4), block, 0, c.st, c.Wo(), c.act, c.z(c.NH - 1), c.chp, s, d_sigma, dz);
        if (want_dw) hipLaunchKernelGGL(wide_dwo_kernel, dim3(LNR_WIDE_H), dim3(1024), 0, c.st, c.act, c.z(c.NH - 1), c.chp, s, d_sigma, slabs + off_o);
        for (int l = c.NH - 1; l >= 0; --l) {
            const int K = l == 0 ? c.K1 : c.H;
            const int64_t layer_off = l == 0 ? 0 : (int64_t)c.H * c.K1 + (int64_t)(l - 1) * c.H * c.H;
            const dim3 grid_w((unsigned)(4 * ((K + 63) / 64) * LNR_WIDE_SPLITS));
            if (!want_dw) {}                                       // frozen parameters (tracking phase): the input gradient only
            else if (HALF && LNR_WIDE_F16_BWD && l > 0) hipLaunchKernelGGL((wide_dw_h_kernel<WIDE_IN_Z>), grid_w, block, 0, c.st, dz, c.chp, c.z(l - 1), c.chp, c.H, K, c.act, s, d_sigma, slabs, n_mlp, layer_off);
            else if (HALF && LNR_WIDE_F16_BWD) hipLaunchKernelGGL((wide_dw_h_kernel<WIDE_IN_PAIR>), grid_w, block, 0, c.st, dz, c.chp, c.feat, c.m_pad, c.spec->enc_dim, K, c.act, s, d_sigma, slabs, n_mlp, layer_off);
            else if (l > 0) hipLaunchKernelGGL((wide_dw_kernel<HALF, WIDE_IN_Z>), grid_w, block, 0, c.st, dz, c.chp, c.z(l - 1), c.chp, c.H, K, c.act, s, slabs, n_mlp, layer_off);
            else if (HALF) hipLaunchKernelGGL((wide_dw_kernel<HALF, WIDE_IN_PAIR>), grid_w, block, 0, c.st, dz, c.chp, c.feat, c.m_pad, c.spec->enc_dim, K, c.act, s, slabs, n_mlp, layer_off);
            else hipLaunchKernelGGL((wide_dw_kernel<HALF, WIDE_IN_FEAT>), grid_w, block, 0, c.st, dz, c.chp, c.feat, c.m_pad, c.spec->enc_dim, K, c.act, s, slabs, n_mlp, layer_off);
            const int64_t count = (int64_t)c.H * K;
            if (want_dw) hipLaunchKernelGGL(wide_fold_kernel, dim3((unsigned)((count + 255) / 256)), block, 0, c.st, slabs, n_mlp, layer_off, count);
            if (l > 0) {
                if (HALF && LNR_WIDE_F16_BWD && LNR_WIDE_F16_DX_HIDDEN)
                    hipLaunchKernelGGL((wide_dx_h_kernel<false>), grid_p, block, 0, c.st, wt + (size_t)(l - 1) * c.H * c.H, c.H / 16, dz, c.chp, c.act, s, d_sigma,
                                       c.z(l - 1), dz_other, c.chp, 0);
                else
                    hipLaunchKernelGGL((wide_dx_kernel<HALF, false>), grid_p, block, 0, c.st, wt + (size_t)(l - 1) * c.H * c.H, c.H / 16, dz, c.chp, c.act, s,
                                       c.z(l - 1), dz_other, c.chp, 0);
                float* t = dz; dz = dz_other; dz_other = t;
            } else if (want_dfeat) {
                if (HALF && LNR_WIDE_F16_BWD)
                    hipLaunchKernelGGL((wide_dx_h_kernel<true>), grid_p, block, 0, c.st, wt_first, K / 16, dz, c.chp, c.act, s, d_sigma, (const float*)nullptr, dfeat,
                                       c.m_pad, c.spec->enc_dim);
                else
                    hipLaunchKernelGGL((wide_dx_kernel<HALF, true>), grid_p, block, 0, c.st, wt_first, K / 16, dz, c.chp, c.act, s, (const float*)nullptr, dfeat,
                                       c.m_pad, c.spec->enc_dim);
            }
        }
    }
    return LNR_OK;
}

static WideCtx wide_ctx(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, void* planes, hipStream_t st) {
    return WideCtx{spec, params, feat, m_pad, pt, (float*)planes, st, spec->precision == LNR_PREC_F16, spec->n_neurons, spec->n_hidden, spec->in_dim, spec->activation, wide_chunk_stride(pt->n_points)};
}
}  // namespace

int lnr_mlp_fwd_wide(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, float* sigma,
                     void* planes, hipStream_t st) {
    const WideCtx c = wide_ctx(spec, params, feat, m_pad, pt, planes, st);
    return c.half ? wide_forward<true>(c, sigma) : wide_forward<false>(c, sigma);
}

// the weight gradient lands in slab 0 (slabs 1 .. LNR_WIDE_SPLITS are the weight-gradient kernel's partial sums): *n_slabs = 1
int lnr_mlp_bwd_wide(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                     float* dfeat, float* slabs, int want_dfeat, int want_dw, int* n_slabs, void* planes, hipStream_t st) {
    const WideCtx c = wide_ctx(spec, params, feat, m_pad, pt, planes, st);
    *n_slabs = 1;
    return c.half ? wide_backward<true>(c, d_sigma, dfeat, slabs, want_dfeat, want_dw) : wide_backward<false>(c, d_sigma, dfeat, slabs, want_dfeat, want_dw);
}
