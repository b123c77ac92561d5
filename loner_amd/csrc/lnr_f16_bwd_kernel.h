// General fp16-mode backward of the density MLP (any supported width / depth / activation): weight gradients of every layer, the
// output row's gradient and the feature gradient (d_feature planes) in ONE launch.  Host dispatch: lnr_density_f16_bwd.hip.
// Semantics = oracle/network.py with precision="fp16": dZ is scaled by an exact power of two before its fp16 conversion so that no
// gradient underflows whatever the loss magnitude, and un-scaled in fp32; products accumulate in fp32.  The oracle scales every
// 32-sample tile by the power of two of its largest |d_sigma|; this kernel keeps ONE unit 2^e for the workgroup - the exponent of the
// largest |d_sigma| of a step's 128 samples, kept while that maximum stays within [2^-4, 2) of it - and rescales the fp32
// accumulators by the exact power of two whenever the unit moves.  The weight-gradient MFMAs then accumulate straight into their
// registers (a per-wave scale meant a separate product, four v_accvgpr_read and two v_pk_fma per MFMA, 1000 VALU instructions a
// step); values differ from the oracle's only where an element lies 2^-20 below its tile's maximum (fp16 subnormals).
//
// Shapes are compile time (HT row tiles, NH hidden layers, KT first-layer K blocks), the weights sit in LDS in the forward kernel's
// layout (lnr_f16_fwd_kernel.h: first-layer rows zero-padded to KT blocks, hidden rows K-permuted, 256-byte rows XOR-swizzled, the
// constant-one input padding as an fp32 bias) and the forward is recomputed ONCE per step with that kernel's row pipeline.
// The round-2 kernel this replaces ran one launch per layer (each re-walking the chain), recomputed every layer's inputs from the
// features, built the sample-major operands of the weight gradient with one ds_write_b16 per element and the W^T operands with one
// ds_read_u16 per element: 2.9 ms for the 128 x 2 network at 2.1 M samples, 4.6 % of the MFMA peak.
//
// Where samples must become the contraction index (dW = dZ X^T), both operands go through LDS as [32 samples][16 columns] row-major
// images - written straight from the MFMA layouts with ds_write_b64 / b128 - and come back through gfx950's transposing read
// (ds_read_b64_tr_b16: a 16-lane group reads a [4][16] block, lane c receives column c), two reads per K = 32 fragment.  The same
// read yields the W^T fragments of dA_in = W^T dZ directly from the row-major weight copy (4 rows x 16 columns per group, the
// lane's address following the K permutation and the swizzle).
// dW ownership: wave w owns row tiles jt = w + 4i of every layer's gradient and runs them over the images of all four waves, so a
// step has two workgroup barriers per layer (images written -> read -> rewritten).
#pragma once
#include <type_traits>
#include "lnr_f16_fwd_kernel.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <int HT, int NH, int KT>
struct BwdLds {
    using W = FwdLds<HT, NH, KT>;
    static constexpr int NT = (NH > 1 && HT > 2 * KT) ? HT : 2 * KT;       // 16-column tiles of the widest layer input
    static constexpr int TILE = 32 * 16;                                    // halves of one [32 samples][16 columns] image tile
    static constexpr int IMG_X = HT * TILE, IMG_WAVE = (HT + NT) * TILE;    // per wave: dZ tiles, then the layer-input tiles
    static constexpr int OFF_IMG = W::N_W + 2 * W::H;                       // halves: behind the weights and the fp32 bias
    static constexpr int OFF_SC = OFF_IMG + 4 * IMG_WAVE;                   // 2 x 4 floats (the waves' |d_sigma| maxima, this step / next), then dWo partials [4][H]
    static constexpr size_t BYTES = (size_t)OFF_SC * sizeof(f16) + 8 * sizeof(float) + 4 * (size_t)W::H * sizeof(float);
};

// K = 32 fragment (8 halves per lane) from two transposing reads: p = the lane's address in the first [4][16] block, the second
// block `second` halves further
__device__ __forceinline__ f16x8 tr_frag(const f16* p, int second) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + second));
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ f16x8 tr_frag_lo(const f16* p) {               // upper four K slots zero (odd tile counts)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p);
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, s16x4{0, 0, 0, 0}, 0, 1, 2, 3, 4, 5, 6, 7));
}

// dZ of a ReLU layer from its packed activations: the fp16 pair survives where the activation is non-zero (two packed integer ops
// per pair instead of a compare and a select per element)
__device__ __forceinline__ uint32_t relu_gate(uint32_t d_pair, uint32_t act_pair) {
    const u16x2 one = {1, 1};
    const u16x2 m = __builtin_elementwise_min(__builtin_bit_cast(u16x2, act_pair), one);
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, d_pair) * m));
}

// FQ: the network's input is the frequency encoding of `src`'s points, evaluated here (lnr_f16_freq.h): no feature planes are read, and
// instead of d_feature planes the kernel writes the input gradient d_pts [n][3] itself (want_dfeat = d_pts wanted): the product
// dX = W1^T dZ is taken in a ROW ORDER that hands every lane the gradients of its OWN slots (rows 4g .. 4g+3 of row tile `it` = the
// (sin, cos) pairs 2 it, 2 it + 1 of lane group g), so the chain rule through sin / cos is lane-local register arithmetic beside the
// re-evaluation of the features for the weight gradient's input image.
#ifdef LNR_PHASE_TIMING
static __device__ unsigned long long lnr_f16_bwd_phase_cycles[LNR_N_PHASES];
#endif

template <int HT, int ACT, int NH, int KT, bool FQ = false>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK, 1)          // one wave per SIMD: the gradient accumulators of all layers live in registers
mlp_backward_f16_gen_kernel(const LnrNetSpec spec, const float* __restrict__ params, const uint32_t* __restrict__ featp, int64_t m_pad,
                            int64_t n_points, const int32_t* __restrict__ n_rays_dev, int n_rays, int n_samples,
                            const float* __restrict__ d_sigma, float* __restrict__ dfeat, float* __restrict__ slabs, int want_dfeat,
                            const PointSrc src, float* __restrict__ d_pts) {
    extern __shared__ __attribute__((aligned(16))) f16 Ws[];
    using L = FwdLds<HT, NH, KT>;
    using B = BwdLds<HT, NH, KT>;
    PHASE_INIT();
    constexpr int H = 16 * HT, KBH = L::KBH;
    constexpr int NO = HT >= 4 ? HT / 4 : 1;                               // row tiles of a gradient a wave owns: jt = wave + 4 i
    constexpr int NHID = NH - 1;                                            // hidden matrices
    constexpr bool RELU = ACT == LNR_ACT_RELU;
    static_assert(NH >= 1 && NH <= F16_NH_MAX && KT >= 1 && KT <= F16_KB_MAX, "shape");
    static_assert(KBH <= F16_KB_MAX || NH == 1, "256 neurons: one hidden layer");
    fwd_fill_weights<HT, NH, KT, FQ>(Ws, params, spec.in_dim, spec.enc_dim, spec.n_frequencies);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int c = lane & 15, g = lane >> 4;                                       // (re-laundered every step: see the loop head)
    const int act = spec.activation, enc_pairs = spec.enc_dim / 2;
    const float* bias_lane = reinterpret_cast<const float*>(Ws + L::N_W) + 4 * g;
    f16* img_all = Ws + B::OFF_IMG;
    f16* img = img_all + wave * B::IMG_WAVE;                                // own images: dZ tiles [HT], input tiles [NT]
    float* mx_s = reinterpret_cast<float*>(Ws + B::OFF_SC);
    float* dwo_s = mx_s + 8;
    __syncthreads();

    const int64_t M = live_samples(n_points, n_rays_dev, n_rays, n_samples);
    const int64_t n_tiles = M > 0 ? (M + 31) / 32 : 0;
    const int64_t per_step = (int64_t)gridDim.x * 4;
    const int64_t n_steps = (n_tiles + per_step - 1) / per_step;            // workgroup-uniform: the loop body has barriers
    const uint32_t plane_bytes = (uint32_t)m_pad * 4u;
    __amdgpu_buffer_rsrc_t rsrc[KT];
#pragma unroll
    for (int kb = 0; kb < KT; ++kb) rsrc[kb] = fwd_block_rsrc(featp, plane_bytes, kb, enc_pairs);
    uint32_t qoff[4];                                                       // byte offsets of the lane's four planes inside a K block, + its column
#pragma unroll
    for (int q = 0; q < 4; ++q) qoff[q] = (uint32_t)(4 * g + q) * plane_bytes + (uint32_t)c * 4u;
    int koff0[F16_KB_MAX], koffh[F16_KB_MAX];
    fwd_frag_offsets<KT, L::S0, L::SWZ0>(c, g, koff0);
    fwd_frag_offsets<(NH > 1 ? KBH : 0), L::SH, L::SWZH>(c, g, koffh);
    float dwo[HT][4];                                                       // the lane's entries of the output row's gradient
#pragma unroll
    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dwo[jt][r] = 0.0f;
    const f16* wo_lane = Ws + L::OFF_O + 4 * g;                             // the output row: the lane's entries of row tile jt at + 16 jt
    f32x4 acc0[NO][2 * KT];                                                 // dW of the first layer: owned row tiles x 16-column tiles of the inputs
    f32x4 acch[NHID > 0 ? NHID : 1][NO][KBH <= F16_KB_MAX ? HT : 1];       // dW of the hidden matrices
#pragma unroll
    for (int i = 0; i < NO; ++i) {
#pragma unroll
        for (int k = 0; k < 2 * KT; ++k) acc0[i][k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (NHID > 0) {
#pragma unroll
            for (int l = 0; l < NHID; ++l)
#pragma unroll
                for (int k = 0; k < HT; ++k) acch[l][i][k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    f32x4 accb[NO];                                                         // first layer, constant-one padding columns: every column = the row sums of dZ
#pragma unroll
    for (int i = 0; i < NO; ++i) accb[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int nt0 = FQ ? 2 * KT : (spec.in_dim + 15) / 16;                  // column tiles of the first layer's gradient
    // FQ (lnr_f16_freq.h): slot sl of this lane = coordinate sl % 3, frequency 4 (sl / 3) + g; a compile-time number of slots
    constexpr int fq_slots = LNR_FREQ_SLOTS_OF_KT(KT);
    const float fq_pg = __uint_as_float((uint32_t)(127 + g) << 23);        // 2^g
    const bool fq_uni = FQ && ray_uniform(src, 32u);                       // a step's 32 samples lie on one ray: its record through the scalar cache
    const int ns_shift = (src.n_samples > 0 && (src.n_samples & (src.n_samples - 1)) == 0) ? __builtin_ctz((unsigned)src.n_samples) : -1;   // samples per ray a power of two: a shift
    // FQ: the points of a step's two column tiles are REQUESTED (loads only) well before they are turned into unit-cube coordinates:
    // with one wave per SIMD nothing else covers a memory round trip (phase timers, profiles/r06_fp16_mlp_phases.txt: requested and
    // consumed in one place, the next step's points were 12 % of the kernel, the current step's re-read part of another 22 %).
    auto request_points = [&](int64_t tile, RawPoint (&rp)[2]) __attribute__((always_inline)) {
        uint32_t mm[2], rr[2];
        if (fq_uni) {
            // the step's samples lie on ONE ray: its index from wave-uniform operands (a shift or one scalar division instead of a
            // per-lane division for every column tile); nothing to clamp (the sample count is a multiple of the step)
            const uint32_t m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tile * 32));
            const uint32_t ray = ns_shift >= 0 ? m0 >> ns_shift : m0 / (uint32_t)__builtin_amdgcn_readfirstlane(src.n_samples);
#pragma unroll
            for (int t = 0; t < 2; ++t) { mm[t] = m0 + 16u * t + (uint32_t)c; rr[t] = ray; }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                int64_t m = tile * 32 + 16 * t + c;
                if (m >= M) m = M - 1;                                      // (clamped to the last live sample)
                mm[t] = (uint32_t)m;
                rr[t] = src.pts ? 0u : (uint32_t)m / (uint32_t)src.n_samples;
            }
        }
        // The SAME seven vector loads per column tile whatever the source, only their addresses differ (points: origin = the point,
        // direction / depth = any valid address, ignored by unit_point).  Loads inside the branches of load_raw_point ended in
        // register copies of the loaded values at the join - a wait for them right where they had been issued - and a scalar load
        // of the ray record would share its counter with the LDS (lgkmcnt): the next weight fragment would wait for it.
        const bool is_pts = src.pts != nullptr;
        const uint64_t b_od = uniform_base(is_pts ? src.pts : src.rays);   // (as scalar registers: base + 32-bit lane offset, no 64-bit lane arithmetic)
        const uint64_t b_z = uniform_base(is_pts ? src.pts : src.z);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t oo = is_pts ? mm[t] * 12u : rr[t] * (uint32_t)(LNR_RAY_STRIDE * 4);
            const uint32_t od = is_pts ? 0u : oo + 12u, oz = is_pts ? 0u : mm[t] * 4u;
            rp[t].o0 = ld32_at<float>(b_od, oo); rp[t].o1 = ld32_at<float>(b_od, oo + 4u); rp[t].o2 = ld32_at<float>(b_od, oo + 8u);
            rp[t].d0 = ld32_at<float>(b_od, od); rp[t].d1 = ld32_at<float>(b_od, od + 4u); rp[t].d2 = ld32_at<float>(b_od, od + 8u);
            rp[t].z = ld32_at<float>(b_z, oz);
        }
    };
    auto unit_points = [&](const RawPoint (&rp)[2], float (&xu)[2][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unit_point(src, rp[t], xu[t]);
#pragma unroll
            for (int d = 0; d < 3; ++d) xu[t][d] *= fq_pg;                  // the lane's share of the frequency (exact)
        }
    };
    // one slot of a step's features (both column tiles); pinned where it is written (see fq_slot of the forward kernel)
    auto fq_slot = [&](int sl, const float (&xu)[2][3], u32x4 (&x)[F16_KB_MAX][2]) __attribute__((always_inline)) {
        if (sl < fq_slots) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float d0, d1;
                float y = xu[t][sl % 3];
                asm volatile("" : "+v"(y));
                uint32_t v = freq_pair<false>(y * lnr_freq_slot_scale(sl), 0.0f, d0, d1);
                asm volatile("" : "+v"(v));
                x[sl >> 2][t][sl & 3] = v;
            }
        }
    };
    auto zero_x = [&](u32x4 (&x)[F16_KB_MAX][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < F16_KB_MAX; ++kb) { x[kb][0] = u32x4{0u, 0u, 0u, 0u}; x[kb][1] = u32x4{0u, 0u, 0u, 0u}; }
    };

    // features (B operands of the first layer; the constant-one padding reads as zero, its weights' gradient is accb below) and
    // d_sigma of a step; a wave without a tile re-reads the last one with d_sigma = 0 (finite operands, zero gradient)
    float xu_c[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};          // FQ: the current step's points (x 2^g), kept for its first-layer stage
    auto load_step = [&](int64_t step, u32x4 (&x)[F16_KB_MAX][2], float (&ds)[2]) {     // (FQ: the first step only)
        const int64_t tile = step * per_step + (int64_t)blockIdx.x * 4 + wave;
        const bool have = tile < n_tiles;
        const int64_t tc = have ? tile : n_tiles - 1;
        const uint32_t m0 = (uint32_t)(tc * 32) * 4u;
        if constexpr (FQ) {
            RawPoint rp[2];
            request_points(tc, rp);
            unit_points(rp, xu_c);
            zero_x(x);
#pragma unroll
            for (int sl = 0; sl < 4 * KT; ++sl) fq_slot(sl, xu_c, x);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if constexpr (!FQ) {
#pragma unroll
            for (int kb = 0; kb < F16_KB_MAX; ++kb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    x[kb][t][q] = kb < KT ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc[kb < KT ? kb : 0], (int)(qoff[q] + m0) + 64 * t, 0, 0) : 0u;
            }
            const int64_t m = tc * 32 + 16 * t + c;
            const bool ok = have && m < M;
            const float v = d_sigma[ok ? m : 0];
            ds[t] = ok ? v : 0.0f;
        }
    };

    // ---- one forward layer with the row pipeline of fwd_layer, keeping what the backward needs.
    // LASTH = false: Aout = the activations as the next layer's B operands, Dout = the activation's derivative in the same packing
    //                (not for ReLU: the gate is read off the activations).
    // LASTH = true:  dzp = dZ of this layer (scaled domain) = ds sc_dn wo act'(Z), and dwo += ds act(Z).
    auto fwd_keep = [&](auto last_tag, auto bias_tag, auto kb_tag, auto s_tag, const f16* Wl, const int (&koff)[F16_KB_MAX],
                        const u32x4 (&Bin)[F16_KB_MAX][2], u32x4 (&Aout)[F16_KB_MAX][2], uint32_t (&Dout)[F16_KB_MAX][2][4],
                        uint32_t (&dzp)[HT][2][2], const float (&dsd)[2], const float (&ds)[2]) {
        constexpr bool LASTH = decltype(last_tag)::value, BIAS = decltype(bias_tag)::value;
        constexpr int KB = decltype(kb_tag)::value, S = decltype(s_tag)::value;
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x8 a[2][KB];
        f32x4 Z[2][2], z0[2];
        f16x4 wo4[3];                                                       // LASTH: the output row's entries of row tiles jt - 1 (being finished), jt, jt + 1 (in flight)
        auto frags = [&](int jt, f16x8 (&dst)[KB], f32x4& zb, f16x4& wv) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) dst[kb] = *reinterpret_cast<const f16x8*>(Wl + 16 * jt * S + koff[kb]);
            zb = BIAS ? *reinterpret_cast<const f32x4*>(bias_lane + 16 * jt) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (LASTH) wv = *reinterpret_cast<const f16x4*>(wo_lane + 16 * jt);
        };
        auto finish = [&](int jt, const f32x4 (&z)[2], const f16x4& wv) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if constexpr (LASTH) {
                    float dz[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dwo[jt][r] = __builtin_fmaf(ds[t], fwd_act<ACT>(z[t][r], act), dwo[jt][r]);
                        dz[r] = (dsd[t] * (float)wv[r]) * gact_d<ACT>(z[t][r], act);
                    }
                    dzp[jt][t][0] = pack_h2(dz[0], dz[1]);
                    dzp[jt][t][1] = pack_h2(dz[2], dz[3]);
                } else {
                    if constexpr (RELU) {
                        Aout[jt >> 1][t][2 * (jt & 1)] = relu_pack_h2(z[t][0], z[t][1]);
                        Aout[jt >> 1][t][2 * (jt & 1) + 1] = relu_pack_h2(z[t][2], z[t][3]);
                    } else {
                        Aout[jt >> 1][t][2 * (jt & 1)] = pack_h2(fwd_act<ACT>(z[t][0], act), fwd_act<ACT>(z[t][1], act));
                        Aout[jt >> 1][t][2 * (jt & 1) + 1] = pack_h2(fwd_act<ACT>(z[t][2], act), fwd_act<ACT>(z[t][3], act));
                    }
                    if constexpr (!RELU) {
                        Dout[jt >> 1][t][2 * (jt & 1)] = pack_h2(gact_d<ACT>(z[t][0], act), gact_d<ACT>(z[t][1], act));
                        Dout[jt >> 1][t][2 * (jt & 1) + 1] = pack_h2(gact_d<ACT>(z[t][2], act), gact_d<ACT>(z[t][3], act));
                    }
                }
            }
        };
        if constexpr (!LASTH) {                                            // every lane of the operand vectors defined before the lane-wise writes
#pragma unroll
            for (int kb = 0; kb < F16_KB_MAX; ++kb)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    Aout[kb][t] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int q = 0; q < 4; ++q) Dout[kb][t][q] = 0u;
                }
        }
        frags(0, a[0], z0[0], wo4[0]);
#pragma unroll
        for (int jt = 0; jt < HT; ++jt) {
            if (jt + 1 < HT) frags(jt + 1, a[(jt + 1) & 1], z0[(jt + 1) & 1], wo4[(jt + 1) % 3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) Z[jt & 1][t] = z0[jt & 1];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    Z[jt & 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[jt & 1][kb], __builtin_bit_cast(f16x8, Bin[kb][t]), Z[jt & 1][t], 0, 0, 0);
            if (jt > 0) finish(jt - 1, Z[(jt - 1) & 1], wo4[(jt - 1) % 3]);
        }
        finish(HT - 1, Z[(HT - 1) & 1], wo4[(HT - 1) % 3]);
    };

    // ---- images: dZ tiles and the layer-input tiles, [32 samples][16 columns] halves each, in 8-byte pieces (sample s, column
    // quad q).  A fragment's first transposing read takes samples 4g..4g+3, its second 16+4g..: K slot 8g+i = sample 4g+i (i < 4) or
    // 16+4g+i-4 - the same permutation on both operands of the product, and the four lane groups of a read cover one contiguous
    // 256-byte block per 32 lanes (rows 8g.. put every group on the same banks: SQ_LDS_BANK_CONFLICT was 56 % of the LDS cycles).
    // Pieces are swizzled inside their 32-byte row so that the 16 lanes of a store, whose rows are 32 bytes apart, spread over the
    // banks too: q ^ (s / 4 % 4) for the 8-byte stores, 16-byte halves p ^ (s / 4 % 2) for the 16-byte stores of the features.
    auto write_dz_image = [&](const uint32_t (&dzp)[HT][2][2]) {
#pragma unroll
        for (int jt = 0; jt < HT; ++jt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                *reinterpret_cast<uint2*>(img + jt * B::TILE + (16 * t + c) * 16 + 4 * (g ^ (c >> 2))) = make_uint2(dzp[jt][t][0], dzp[jt][t][1]);
    };
    auto write_x_image_first = [&](const u32x4 (&x)[F16_KB_MAX][2]) {       // natural K order: 8 consecutive inputs per lane
#pragma unroll
        for (int kb = 0; kb < KT; ++kb)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                *reinterpret_cast<u32x4*>(img + B::IMG_X + (2 * kb + (g >> 1)) * B::TILE + (16 * t + c) * 16 + 8 * ((g & 1) ^ ((c >> 2) & 1))) = x[kb][t];
    };
    auto write_x_image_hidden = [&](const u32x4 (&A)[F16_KB_MAX][2]) {      // permuted K order: neurons 4g..4g+3 of tiles 2kb and 2kb+1
#pragma unroll
        for (int kb = 0; kb < (KBH <= F16_KB_MAX ? KBH : 0); ++kb)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                *reinterpret_cast<uint2*>(img + B::IMG_X + (2 * kb) * B::TILE + (16 * t + c) * 16 + 4 * (g ^ (c >> 2))) = make_uint2(A[kb][t][0], A[kb][t][1]);
                if (2 * kb + 1 < HT)
                    *reinterpret_cast<uint2*>(img + B::IMG_X + (2 * kb + 1) * B::TILE + (16 * t + c) * 16 + 4 * (g ^ (c >> 2))) = make_uint2(A[kb][t][2], A[kb][t][3]);
            }
    };
    // the lane's address inside an image tile (halves): row 4g + c/4 (s / 4 % 4 = g), column quad c % 4; second read 16 rows on
    const int tr_lane_q = (4 * g + (c >> 2)) * 16 + 4 * ((c & 3) ^ g);                                  // 8-byte swizzle
    const int tr_lane_p = (4 * g + (c >> 2)) * 16 + 8 * (((c >> 1) & 1) ^ (g & 1)) + 4 * (c & 1);      // 16-byte swizzle

    // dW += dZ^T (own row tiles) x inputs^T over the four waves' images, accumulated by the MFMAs themselves; NT_L column tiles
    // (nt of them live).  Work units = (wave image, half of the column tiles); the fragments of the next unit are requested before
    // the products of the current one are issued.
    // side(u): independent VALU work issued behind the MFMAs of unit u (FQ: one slot of the next step's features)
    auto accumulate_dw = [&](auto nt_tag, auto ones_tag, f32x4 (&acc)[NO][decltype(nt_tag)::value], int nt, int xlane, auto side) {
        constexpr int NT_L = decltype(nt_tag)::value;
        constexpr bool ONES = decltype(ones_tag)::value;                    // also accumulate dZ^T x 1 (the padding columns of the first layer)
        constexpr int HALF = NT_L >= 4 ? NT_L / 2 : NT_L, NU = 4 * (NT_L / HALF);
        f16x8 a[2][NO], b[2][HALF];
        auto frags = [&](int u, f16x8 (&af)[NO], f16x8 (&bf)[HALF]) {
            const int w2 = u / (NT_L / HALF), k0 = (u % (NT_L / HALF)) * HALF;
            const f16* iw = img_all + w2 * B::IMG_WAVE;
#pragma unroll
            for (int i = 0; i < NO; ++i) {
                const int jt = wave + 4 * i;
                af[i] = tr_frag(iw + (jt < HT ? jt : 0) * B::TILE + tr_lane_q, 256);
            }
#pragma unroll
            for (int k = 0; k < HALF; ++k) bf[k] = tr_frag(iw + B::IMG_X + (k0 + k < nt ? k0 + k : 0) * B::TILE + xlane, 256);
        };
        frags(0, a[0], b[0]);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + 1 < NU) frags(u + 1, a[(u + 1) & 1], b[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int k0 = (u % (NT_L / HALF)) * HALF;
            if constexpr (ONES) {
                if (k0 == 0) {
                    const f16x8 ones = frag_from_dwords(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);
#pragma unroll
                    for (int i = 0; i < NO; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 1][i], ones, accb[i], 0, 0, 0);
                }
            }
#pragma unroll
            for (int k = 0; k < HALF; ++k) {
                if (k0 + k < nt) {
#pragma unroll
                    for (int i = 0; i < NO; ++i) acc[i][k0 + k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 1][i], b[u & 1][k], acc[i][k0 + k], 0, 0, 0);
                }
            }
            side(u);
        }
    };
    constexpr int NU_FIRST = 4 * ((2 * KT) / ((2 * KT) >= 4 ? KT : 2 * KT));   // units of the first layer's accumulate_dw (its NU)

    // W^T fragments through the transposing read: rows = four consecutive output neurons j, columns = 16 inputs of tile `it`.
    // Hidden matrices: input neuron n of tile it sits at K slot 32 (it >> 1) + 8 (n % 16 / 4) + 4 (it & 1) + n % 4 of its row.
    auto wt_frag_hidden = [&](const f16* Wl, int it, int kb) -> f16x8 {
        const int j0 = 32 * kb + 4 * g + (c >> 2);                          // the lane's row of the first block (second: + 16)
        const int p0 = 32 * (it >> 1) + 8 * (c & 3) + 4 * (it & 1);
        const f16* p = Wl + j0 * L::SH + fwd_slot<L::SWZH>(p0, j0);          // (j0 + 16) & 15 == j0 & 15: one swizzle for both blocks
        return 2 * kb + 1 < HT ? tr_frag(p, 16 * L::SH) : tr_frag_lo(p);
    };
    auto wt_frag_first = [&](int it, int kb) -> f16x8 {                    // natural K order: input k = 16 it + 4 (c & 3) ..
        const int j0 = 32 * kb + 4 * g + (c >> 2);
        // (FQ: row 4u + i of row tile `it` = K position of lane group u's slot 2 it + i / 2, half i % 2 - see the kernel's head)
        const int p0 = FQ ? 32 * (it >> 1) + 8 * (c & 3) + 4 * (it & 1) : 16 * it + 4 * (c & 3);
        const f16* p = Ws + j0 * L::S0 + fwd_slot<L::SWZ0>(p0, j0);
        return 2 * kb + 1 < HT ? tr_frag(p, 16 * L::S0) : tr_frag_lo(p);
    };
    auto dz_frag = [&](const uint32_t (&dzp)[HT][2][2], int kb, int t) -> f16x8 {   // dZ as a B operand (K slots = neurons, permuted order)
        const bool second = 2 * kb + 1 < HT;
        const int j1 = second ? 2 * kb + 1 : 2 * kb;
        return frag_from_dwords(dzp[2 * kb][t][0], dzp[2 * kb][t][1], second ? dzp[j1][t][0] : 0u, second ? dzp[j1][t][1] : 0u);
    };

    u32x4 x[F16_KB_MAX][2], xn[F16_KB_MAX][2];
    float ds[2], dsn[2];
    uint32_t e_unit = 127u;                                                 // biased exponent of the unit dZ is expressed in (workgroup-uniform)
    bool have_unit = false;
    if (n_steps > 0) {
        load_step(0, x, ds);
        const float mx0 = wave_max(fmaxf(fabsf(ds[0]), fabsf(ds[1])));
        if (lane == 0) mx_s[wave] = mx0;
    }
    __syncthreads();
    PHASE(0);
    for (int64_t step = 0; step < n_steps; ++step) {
        const int64_t tile = step * per_step + (int64_t)blockIdx.x * 4 + wave;
        const bool have_tile = tile < n_tiles;
        // The lane addresses of the W^T fragments and the image stores (dozens of distinct values of c and g) are cheap to recompute;
        // hoisted out of this loop as invariants they were spilled to scratch and re-read every step (85 scratch loads a step).
        asm volatile("" : "+v"(c), "+v"(g));
        // the unit: exponent of the step's largest |d_sigma| (its maximum then lies in [1, 2)), kept while that stays within [2^-4, 2)
        const float* mxb = mx_s + 4 * (int)(step & 1);
        const float mxg = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(fmaxf(fmaxf(mxb[0], mxb[1]), fmaxf(mxb[2], mxb[3])))));
        uint32_t be = (__float_as_uint(mxg) >> 23) & 0xFFu;
        be = be < 1u ? 1u : (be > 253u ? 253u : be);
        if (mxg > 0.0f && (!have_unit || be > e_unit || be + 4u < e_unit)) {
            if (have_unit) {
                if (be + 96u < e_unit) be = e_unit - 96u;                   // the accumulators grow by 2^(e_unit - be): bounded
                const int sh = (int)e_unit - (int)be;                       // exact: a power of two (0 when nothing of the old sum would survive)
                const float f = sh < -126 ? 0.0f : __uint_as_float((uint32_t)(127 + sh) << 23);
#pragma unroll
                for (int i = 0; i < NO; ++i) {
#pragma unroll
                    for (int k = 0; k < 2 * KT; ++k) acc0[i][k] *= f;
                    accb[i] *= f;
                    if constexpr (NHID > 0) {
#pragma unroll
                        for (int l = 0; l < NHID; ++l)
#pragma unroll
                            for (int k = 0; k < HT; ++k) acch[l][i][k] *= f;
                    }
                }
            }
            e_unit = be;
            have_unit = true;
        }
        const float sc_dn = __uint_as_float((254u - e_unit) << 23), sc_up = __uint_as_float(e_unit << 23);
        const float dsd[2] = {ds[0] * sc_dn, ds[1] * sc_dn};
        PHASE(1);

        // ---- forward, once: A[l] = inputs of hidden matrix l + 1 (B operands), Dv[l] = derivative of layer l's activation
        u32x4 A[NHID > 0 ? NHID : 1][F16_KB_MAX][2];
        uint32_t Dv[(!RELU && NHID > 0) ? NHID : 1][F16_KB_MAX][2][4];
        uint32_t dzp[HT][2][2];
        using T = std::true_type; using F = std::false_type;
        using KT_ = std::integral_constant<int, KT>; using KH_ = std::integral_constant<int, (NH > 1 ? KBH : 1)>;
        using S0_ = std::integral_constant<int, L::S0>; using SH_ = std::integral_constant<int, L::SH>;
        if constexpr (NH == 1) {
            fwd_keep(T{}, T{}, KT_{}, S0_{}, Ws, koff0, x, A[0], Dv[0], dzp, dsd, ds);
        } else {
            fwd_keep(F{}, T{}, KT_{}, S0_{}, Ws, koff0, x, A[0], Dv[0], dzp, dsd, ds);
            PHASE(2);
#pragma unroll
            for (int l = 1; l < NH - 1; ++l)
                fwd_keep(F{}, F{}, KH_{}, SH_{}, Ws + L::OFF_H + (l - 1) * H * L::SH, koffh, A[l - 1], A[l], Dv[RELU ? 0 : l], dzp, dsd, ds);
            fwd_keep(T{}, F{}, KH_{}, SH_{}, Ws + L::OFF_H + (NH - 2) * H * L::SH, koffh, A[NH - 2], A[0], Dv[0], dzp, dsd, ds);
        }

        PHASE(3);
        // ---- backward through the hidden matrices l = NH-1 .. 1 (matrix l maps A[l-1] to layer l's pre-activations)
        if constexpr (NH > 1) {
#pragma unroll
            for (int l = NH - 1; l >= 1; --l) {
                const f16* Wl = Ws + L::OFF_H + (l - 1) * H * L::SH;
                write_dz_image(dzp);
                write_x_image_hidden(A[l - 1]);
                PHASE(4);
                // dA_{l-1} = W_l^T dZ_l (scaled domain), gated by layer l-1's derivative -> dZ_{l-1}
                uint32_t dzn[HT][2][2];
                f16x8 wt[2][KBH];                                          // W^T fragments of row tile it + 1 in flight behind the products of it
#pragma unroll
                for (int kb = 0; kb < KBH; ++kb) wt[0][kb] = wt_frag_hidden(Wl, 0, kb);
#pragma unroll
                for (int it = 0; it < HT; ++it) {
                    if (it + 1 < HT) {
#pragma unroll
                        for (int kb = 0; kb < KBH; ++kb) wt[(it + 1) & 1][kb] = wt_frag_hidden(Wl, it + 1, kb);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 Dq[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
                    for (int kb = 0; kb < KBH; ++kb) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) Dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wt[it & 1][kb], dz_frag(dzp, kb, t), Dq[t], 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if constexpr (RELU) {
                            dzn[it][t][0] = relu_gate(pack_h2(Dq[t][0], Dq[t][1]), A[l - 1][it >> 1][t][2 * (it & 1)]);
                            dzn[it][t][1] = relu_gate(pack_h2(Dq[t][2], Dq[t][3]), A[l - 1][it >> 1][t][2 * (it & 1) + 1]);
                        } else {
                            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                            const h2 d01 = __builtin_bit_cast(h2, Dv[l - 1][it >> 1][t][2 * (it & 1)]), d23 = __builtin_bit_cast(h2, Dv[l - 1][it >> 1][t][2 * (it & 1) + 1]);
                            dzn[it][t][0] = pack_h2(Dq[t][0] * (float)d01[0], Dq[t][1] * (float)d01[1]);
                            dzn[it][t][1] = pack_h2(Dq[t][2] * (float)d23[0], Dq[t][3] * (float)d23[1]);
                        }
                    }
                }
                PHASE(5);
                __syncthreads();
                PHASE(6);
                accumulate_dw(std::integral_constant<int, HT>{}, F{}, acch[l - 1], HT, tr_lane_q, FwdNoSide());
                PHASE(7);
                __syncthreads();                                           // the images are rewritten by the next layer
                PHASE(8);
#pragma unroll
                for (int jt = 0; jt < HT; ++jt)
#pragma unroll
                    for (int t = 0; t < 2; ++t) { dzp[jt][t][0] = dzn[jt][t][0]; dzp[jt][t][1] = dzn[jt][t][1]; }
            }
        }
        // ---- first layer: its weight gradient and the feature gradient (un-scaled, to the d_feature planes)
        write_dz_image(dzp);
        // The features are not kept across the hidden layers (32 registers at the point of highest pressure): they are read again
        // here (L2) for the input image, together with the next step's operands (a static number of loads: the last step re-reads itself)
        RawPoint rn[2];                                                    // FQ: the next step's points and d_sigma, requested here ...
        float xu_n[2][3];                                                  // ... and consumed behind the first layer's products
        float dsn_raw[2] = {0.0f, 0.0f};
        bool dsn_ok[2] = {false, false};
        if constexpr (!FQ) {
            load_step(step, x, ds);
            load_step(step + 1 < n_steps ? step + 1 : step, xn, dsn);
        } else {
            const int64_t ntile = (step + 1 < n_steps ? step + 1 : step) * per_step + (int64_t)blockIdx.x * 4 + wave;
            const bool nhave = ntile < n_tiles;
            const int64_t ntc = nhave ? ntile : n_tiles - 1;
            request_points(ntc, rn);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t m = ntc * 32 + 16 * t + c;
                dsn_ok[t] = nhave && m < M;
                dsn_raw[t] = d_sigma[dsn_ok[t] ? m : 0];
            }
        }
        PHASE(9);
        if constexpr (FQ) {
            // the features again (input image of the weight gradient) and, with them, the input gradient: slot pair `it` of every lane
            float (&xu)[2][3] = xu_c;
            float acc[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
            const float dph_g = fq_pg * LNR_PI_F;                          // d(phase)/dx of the lane's share; x 2^(4 (sl / 3)) per slot
#pragma unroll
            for (int kb = 0; kb < F16_KB_MAX; ++kb) { x[kb][0] = u32x4{0u, 0u, 0u, 0u}; x[kb][1] = u32x4{0u, 0u, 0u, 0u}; }
#pragma unroll
            for (int it = 0; it < 2 * KT; ++it) {
                if (2 * it >= fq_slots) break;                             // (compile-time) no slot from here on
                f32x4 Dq[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
                if (want_dfeat) {
#pragma unroll
                    for (int kb = 0; kb < KBH; ++kb) {
                        const f16x8 a = wt_frag_first(it, kb);
#pragma unroll
                        for (int t = 0; t < 2; ++t) Dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, dz_frag(dzp, kb, t), Dq[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int sl = 2 * it + j;
                    if (sl >= fq_slots) break;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        float dsn_, dcs_;
                        x[it >> 1][t][2 * (it & 1) + j] = freq_pair<true>(xu[t][sl % 3] * lnr_freq_slot_scale(sl), dph_g * lnr_freq_slot_scale(sl), dsn_, dcs_);
                        acc[t][sl % 3] += __builtin_fmaf(Dq[t][2 * j], dsn_, Dq[t][2 * j + 1] * dcs_);
                    }
                }
            }
            // the next step's points and d_sigma are taken over HERE, in front of the d_pts stores: loads and stores share one in-order
            // counter (vmcnt) and the stores sit in a branch, so a wait for these loads placed behind them would be a wait for the
            // stores' acknowledgement (a memory round trip with nothing to cover it)
#pragma unroll
            for (int t = 0; t < 2; ++t) dsn[t] = dsn_ok[t] ? dsn_raw[t] : 0.0f;
            unit_points(rn, xu_n);
#pragma unroll
            for (int t = 0; t < 2; ++t) {                                  // (pinned: nothing else keeps this arithmetic in front of the stores)
                asm volatile("" : "+v"(dsn[t]));
#pragma unroll
                for (int d = 0; d < 3; ++d) asm volatile("" : "+v"(xu_n[t][d]));
            }
            if (want_dfeat) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    // a sample's gradient = the sum over its four lanes (c, g = 0..3): the two cross-row swaps of CDNA4
                    float tot[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        typedef unsigned u2v __attribute__((ext_vector_type(2)));
                        const unsigned a0 = __float_as_uint(acc[t][d]);
                        const u2v r1 = __builtin_amdgcn_permlane16_swap(a0, a0, false, false);      // rows (0,1) and (2,3) meet
                        const float s1 = __uint_as_float(r1.x) + __uint_as_float(r1.y);
                        const unsigned b0 = __float_as_uint(s1);
                        const u2v r2 = __builtin_amdgcn_permlane32_swap(b0, b0, false, false);      // the two halves meet
                        tot[d] = __uint_as_float(r2.x) + __uint_as_float(r2.y);
                    }
                    const int64_t m = tile * 32 + 16 * t + c;
                    const float mine = g == 0 ? tot[0] : (g == 1 ? tot[1] : tot[2]);
                    if (have_tile && m < M && g < 3) d_pts[3 * m + g] = 0.5f * sc_up * mine;          // x = (xyz + 1) / 2
                }
            }
        } else
        if (want_dfeat) {
            for (int it = 0; it < nt0; ++it) {
                f32x4 Dq[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
                for (int kb = 0; kb < KBH; ++kb) {
                    const f16x8 a = wt_frag_first(it, kb);
#pragma unroll
                    for (int t = 0; t < 2; ++t) Dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, dz_frag(dzp, kb, t), Dq[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int64_t m = tile * 32 + 16 * t + c;
                    if (have_tile && m < M) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int k = 16 * it + 4 * g + r;
                            if (k < 2 * enc_pairs) st32<float>(dfeat, (uint32_t)k * plane_bytes + (uint32_t)m * 4u, sc_up * Dq[t][r]);
                        }
                    }
                }
            }
        }
        PHASE(10);
        write_x_image_first(x);
        if constexpr (FQ) zero_x(xn);
        {
            const float mxn = wave_max(fmaxf(fabsf(dsn[0]), fabsf(dsn[1])));   // the next step's maximum, exchanged through the other buffer
            if (lane == 0) mx_s[4 * (int)((step + 1) & 1) + wave] = mxn;
        }
        PHASE(11);
        __syncthreads();
        PHASE(12);
        if constexpr (FQ) {
            // the next step's features, one slot behind the MFMAs of each unit of the weight gradient (the rest after it)
            accumulate_dw(std::integral_constant<int, 2 * KT>{}, T{}, acc0, nt0, tr_lane_p, [&](int u) __attribute__((always_inline)) { fq_slot(u, xu_n, xn); });
#pragma unroll
            for (int sl = NU_FIRST; sl < 4 * KT; ++sl) fq_slot(sl, xu_n, xn);
        } else {
            accumulate_dw(std::integral_constant<int, 2 * KT>{}, T{}, acc0, nt0, tr_lane_p, FwdNoSide());
        }
        PHASE(13);
        __syncthreads();                                                   // the images are rewritten by the next step
        PHASE(14);
#pragma unroll
        for (int kb = 0; kb < F16_KB_MAX; ++kb) { x[kb][0] = xn[kb][0]; x[kb][1] = xn[kb][1]; }
        ds[0] = dsn[0]; ds[1] = dsn[1];
        if constexpr (FQ) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int d = 0; d < 3; ++d) xu_c[t][d] = xu_n[t][d];
        }
    }

    // ---- the workgroup's slab (parameter layout, unpadded): every row tile is owned by exactly one wave.  A row tile's 16 rows are one
    // contiguous piece of the slab (16 x in_dim / 16 x H floats): the wave lays it out in its own image area (free after the last
    // barrier; LDS operations of one wave complete in order, so no barrier) and copies it out as whole 256-byte rows of the wave.
    // Written straight from the accumulator layout - 4-byte stores at a row stride per lane, for the fused first layer at scattered
    // feature positions too - the 120 KB slab of the 128 x 2 network took 25 - 30 us per workgroup (profiles/r06_fp16_mlp_phases.txt).
    const int n_mlp = spec.n_mlp_params, in_dim = spec.in_dim;
    const float unit = __uint_as_float(e_unit << 23);
    float* slab = slabs + (size_t)blockIdx.x * n_mlp;
    float* stg = reinterpret_cast<float*>(img);
    static_assert((size_t)B::IMG_WAVE * sizeof(f16) >= 16u * 32u * KT * sizeof(float) && (size_t)B::IMG_WAVE * sizeof(f16) >= 16u * H * sizeof(float), "a row tile fits the wave's image area");
#pragma unroll
    for (int i = 0; i < NO; ++i) {
        const int jt = wave + 4 * i;
        if (jt < HT) {
#pragma unroll
            for (int kt = 0; kt < 2 * KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 16 * kt + c;
                    if constexpr (FQ) {                                     // K position -> the feature evaluated there; the ones-padding columns once
                        const int k = lnr_freq_feature_at(col, spec.n_frequencies);
                        if (k >= 0) stg[(4 * g + r) * in_dim + k] = unit * acc0[i][kt][r];
                        if (kt == 0 && spec.enc_dim + c < in_dim) stg[(4 * g + r) * in_dim + spec.enc_dim + c] = unit * accb[i][r];
                    } else
                    if (col < in_dim) stg[(4 * g + r) * in_dim + col] = unit * (col < spec.enc_dim ? acc0[i][kt][r] : accb[i][r]);
                }
            {
                float* dst = slab + 16 * jt * in_dim;
                for (int j = lane; j < 16 * in_dim; j += 64) dst[j] = stg[j];
            }
            if constexpr (NHID > 0) {
#pragma unroll
                for (int l = 0; l < NHID; ++l) {
#pragma unroll
                    for (int kt = 0; kt < HT; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) stg[(4 * g + r) * H + 16 * kt + c] = unit * acch[l][i][kt][r];
                    float* dst = slab + H * in_dim + l * H * H + 16 * jt * H;
#pragma unroll
                    for (int j = 0; j < 16 * H / 64; ++j) dst[64 * j + lane] = stg[64 * j + lane];
                }
            }
        }
    }
    // output row: every wave has a partial over its own samples; summed in a fixed order.  Rows 1..15 of the padded output matrix: 0.
#pragma unroll
    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = dwo[jt][r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (c == 0) dwo_s[wave * H + 16 * jt + 4 * g + r] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * H; i += blockDim.x)
        slab[H * in_dim + NHID * H * H + i] = i < H ? ((dwo_s[i] + dwo_s[H + i]) + (dwo_s[2 * H + i] + dwo_s[3 * H + i])) : 0.0f;
    PHASE(15);
    PHASE_FLUSH(lnr_f16_bwd_phase_cycles, 0);
}
