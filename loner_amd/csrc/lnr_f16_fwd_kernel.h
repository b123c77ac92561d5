// General fp16-mode forward of the density MLP (any supported width / depth / activation): the kernel and its LDS layout.
// Host dispatch: lnr_density_f16_fwd.hip.  Semantics = oracle/network.py with precision="fp16" (reference: tinycudann FullyFusedMLP
// behind src/models/nerf_tcnn.py:35-38).
//
// Everything that shapes the instruction stream is a template parameter - HT (16-neuron row tiles), NH (hidden layers), KT (32-wide
// K blocks of the first layer, the inputs' blocks rounded up to 2 or 4), CT (16-sample column tiles a wave carries) - because the
// round-3 counters showed what run-time shapes cost here: as a run-time layer loop with run-time first-layer block counts the kernel
// issued 1030 v_mov per 32-sample tile (activations COPIED from one layer's registers to the next, accumulators copied between the
// arms of `kb < kt1` branches) beside 112 MFMAs, and waves sat in s_waitcnt 45 % of their cycles because every row of every layer
// exposed an LDS round trip in front of its MFMAs.
//   * layers are unrolled at compile time: a layer's outputs ARE the next layer's B operands (register renaming, no copies);
//   * the first layer's rows are zero-padded in LDS to KT blocks: every fragment load and MFMA is unconditional;
//   * hidden rows are stored K-PERMUTED (slot 8g+i of a 32-neuron block = neuron 4g+i / 16+4g+i-4: the permutation that makes one
//     product's C layout the next one's B layout), so a hidden fragment is ONE ds_read_b128 like a first-layer one (it was two
//     ds_read_b64 at a 2-way bank conflict each: SQ_LDS_BANK_CONFLICT was 45 % of the LDS cycles);
//   * rows whose length is a multiple of 256 bytes are XOR-swizzled by 16-byte chunk (chunk ^ (row & 15)) instead of padded: the
//     ds_read_b128 lane groups of an A fragment (16 rows, g differing by at most one inside a group) then hit 64 distinct banks;
//   * the constant-one padding of the network input (tinycudann pads the encoding to a multiple of 16 with ones) is a per-neuron
//     bias: the sum of its first-layer weights, kept as fp32 in LDS, is the C operand the first product of a row starts from, and
//     the padded weight columns are stored as zeros;
//   * features come through buffer loads, one descriptor per K block whose num_records ends at the last encoded plane: planes
//     beyond the encoding read as zero in hardware, and a step costs 4 address additions per column tile instead of a 64-bit
//     multiply-add, two compares and two selects per loaded dword;
//   * the fragments of row jt + 1 are requested, and pinned there by a scheduling barrier, BEFORE the MFMAs of row jt; the
//     activations of row jt - 1 are applied after them, in the shadow of the matrix pipe.
#pragma once
#include "lnr_f16_common.h"
#include "lnr_f16_freq.h"

template <int HT, int NH, int KT>
struct FwdLds {
    static constexpr int H = 16 * HT, KBH = (HT + 1) / 2;
    static constexpr int K0 = 32 * KT, KH = 32 * KBH;                       // halves per stored row: first layer / hidden layers
    static constexpr bool SWZ0 = K0 % 128 == 0, SWZH = KH % 128 == 0;       // 256-byte multiples: swizzled, else padded by 16 bytes
    static constexpr int S0 = SWZ0 ? K0 : K0 + 8, SH = SWZH ? KH : KH + 8;  // row strides (halves)
    static constexpr int OFF_H = H * S0, OFF_O = OFF_H + (NH - 1) * H * SH, N_W = (OFF_O + H + 7) & ~7;
    static constexpr size_t BYTES = (size_t)N_W * sizeof(f16) + (size_t)H * sizeof(float);   // + the first layer's bias (fp32)
};

// stored position (halves, within its row) of K slot `col` of row `row`
template <bool SWZ> __device__ __forceinline__ int fwd_slot(int col, int row) {
    return SWZ ? ((((col >> 3) ^ (row & 15)) << 3) | (col & 7)) : col;
}

// fp32 parameters (tinycudann layout: W1 [H][in_dim], hidden [H][H] each, output row [H]) -> the fp16 LDS copy described above.
// The fill is the fixed cost of every launch (each persistent workgroup converts the whole network in front of its first MFMA), and it
// is made of exposed L2 round trips: as `for (i = tid; i < n; i += blockDim.x)` every element was one (128 in a row for the 128 x 2
// network: ~20 us); unrolled by 8 it was 14 rounds (~13 us of the 161 us north-star forward, profiles/r06_fp16_mlp_phases.txt).
// Now a thread takes 2 (first layer: a (sin, cos) pair / two neighbouring features) or 4 (hidden: one lane group's four neurons)
// consecutive parameters per load and the trip counts are compile-time and fully unrolled: every load of the fill is in flight at once.
// FQ (the frequency encoding computed in the kernel, lnr_f16_freq.h): K position `col` of a first-layer row holds the weight of the
// feature the fused kernels evaluate there (nf = n_frequencies), padding positions zero.
template <int HT, int NH, int KT, bool FQ = false>
__device__ __forceinline__ void fwd_fill_weights(f16* Ws, const float* __restrict__ params, int in_dim, int enc_dim, int nf = 0) {
    using L = FwdLds<HT, NH, KT>;
    constexpr int NT = LNR_DENSITY_BLOCK;
    const int n0 = L::H * in_dim, tid = threadIdx.x;
    static_assert((L::H * L::K0) % NT == 0 && (L::H * L::KH) % NT == 0, "whole trips");
    constexpr int T0 = (L::H * L::K0 + 2 * NT - 1) / (2 * NT), TH = (L::H * L::KH + 4 * NT - 1) / (4 * NT);   // vector trips (the last one may be partial)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // vector form: 8-/16-byte loads need even first-layer rows (in_dim, enc_dim) and a 16-byte aligned parameter vector
    const bool vec = ((in_dim | enc_dim) & 1) == 0 && (reinterpret_cast<uintptr_t>(params) & 15u) == 0;
    if (vec) {
        f32x2 w0[T0];
        f32x4 wh[NH > 1 ? NH - 1 : 1][TH];
#pragma unroll
        for (int it = 0; it < T0; ++it) {
            const int i = 2 * (it * NT + tid), row = i / L::K0 % L::H, col = i % L::K0;     // (% H: a partial last trip re-reads row 0, not stored)
            int k = col;
            if constexpr (FQ) k = lnr_freq_feature_at(col, nf);          // (col even: the pair's sine; its cosine is the next parameter)
            const bool live = FQ ? k >= 0 : col < enc_dim;
            w0[it] = *reinterpret_cast<const f32x2*>(params + row * in_dim + (live ? k : 0));
            if (!live) w0[it] = f32x2{0.0f, 0.0f};
        }
        if constexpr (NH > 1) {
#pragma unroll
            for (int l = 0; l < NH - 1; ++l)
#pragma unroll
                for (int it = 0; it < TH; ++it) {
                    const int i = 4 * (it * NT + tid), row = i / L::KH % L::H, p = i % L::KH;
                    const int g = (p >> 3) & 3, ii = p & 7;
                    const int n = (p & ~31) + (ii < 4 ? 4 * g + ii : 16 + 4 * g + ii - 4);   // the neurons K slots p .. p + 3 stand for: n .. n + 3
                    wh[l][it] = *reinterpret_cast<const f32x4*>(params + n0 + (l * L::H + row) * L::H + (n < L::H ? n : 0));
                    if (n >= L::H) wh[l][it] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
        }
#pragma unroll
        for (int it = 0; it < T0; ++it) {
            const int i = 2 * (it * NT + tid), row = i / L::K0, col = i % L::K0;
            if (i < L::H * L::K0) *reinterpret_cast<uint32_t*>(Ws + row * L::S0 + fwd_slot<L::SWZ0>(col, row)) = pack_h2(w0[it][0], w0[it][1]);
        }
        if constexpr (NH > 1) {
#pragma unroll
            for (int l = 0; l < NH - 1; ++l)
#pragma unroll
                for (int it = 0; it < TH; ++it) {
                    const int i = 4 * (it * NT + tid), row = i / L::KH, p = i % L::KH;
                    if (i < L::H * L::KH) *reinterpret_cast<uint2*>(Ws + L::OFF_H + (l * L::H + row) * L::SH + fwd_slot<L::SWZH>(p, row)) =
                        make_uint2(pack_h2(wh[l][it][0], wh[l][it][1]), pack_h2(wh[l][it][2], wh[l][it][3]));
                }
        }
    } else {
#pragma unroll 8
        for (int it = 0; it < L::H * L::K0 / NT; ++it) {
            const int i = it * NT + tid, row = i / L::K0, col = i % L::K0;
            int k = col;
            if constexpr (FQ) k = lnr_freq_feature_at(col, nf);
            const bool live = FQ ? k >= 0 : col < enc_dim;
            const float w = params[row * in_dim + (live ? k : 0)];
            Ws[row * L::S0 + fwd_slot<L::SWZ0>(col, row)] = live ? (f16)w : (f16)0.0f;
        }
        if constexpr (NH > 1) {
            for (int l = 0; l < NH - 1; ++l) {
#pragma unroll 8
                for (int it = 0; it < L::H * L::KH / NT; ++it) {
                    const int i = it * NT + tid, row = i / L::KH, p = i % L::KH;
                    const int g = (p >> 3) & 3, ii = p & 7;
                    const int n = (p & ~31) + (ii < 4 ? 4 * g + ii : 16 + 4 * g + ii - 4);       // the neuron K slot p stands for
                    const float w = params[n0 + (l * L::H + row) * L::H + (n < L::H ? n : 0)];
                    Ws[L::OFF_H + (l * L::H + row) * L::SH + fwd_slot<L::SWZH>(p, row)] = n < L::H ? (f16)w : (f16)0.0f;
                }
            }
        }
    }
    float* bias = reinterpret_cast<float*>(Ws + L::N_W);
    for (int i = tid; i < L::H; i += NT) {
        Ws[L::OFF_O + i] = (f16)params[n0 + (NH - 1) * L::H * L::H + i];
        float b = 0.0f;
        for (int k0 = enc_dim; k0 < in_dim; k0 += 8) {                    // (eight loads in flight; the sum in index order)
            float w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = params[i * in_dim + (k0 + j < in_dim ? k0 + j : k0)];
#pragma unroll
            for (int j = 0; j < 8; ++j) if (k0 + j < in_dim) b += (float)(f16)w[j];
        }
        bias[i] = b;
    }
}

// per-lane LDS offsets (halves) of the A fragments of row tile 0: K block kb of row c, K slots 8g..8g+7
template <int KB, int S, bool SWZ>
__device__ __forceinline__ void fwd_frag_offsets(int c, int g, int (&koff)[F16_KB_MAX]) {
#pragma unroll
    for (int kb = 0; kb < F16_KB_MAX; ++kb) koff[kb] = kb < KB ? c * S + fwd_slot<SWZ>(32 * kb + 8 * g, c) : 0;
}

// Buffer descriptor over the feature planes of K block kb (half2 pairs 16kb .. 16kb+15, planes [pair][sample] of plane_bytes each),
// ending at the last ENCODED plane: anything beyond reads as zero (raw buffer range check; an empty block has zero records).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fwd_block_rsrc(const uint32_t* featp, uint32_t plane_bytes, int kb, int enc_pairs) {
    const int planes = enc_pairs - 16 * kb;
    const uint32_t n = planes <= 0 ? 0u : (uint32_t)(planes < 16 ? planes : 16) * plane_bytes;
    const char* base = reinterpret_cast<const char*>(featp) + (planes <= 0 ? 0 : (size_t)(16 * kb) * plane_bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)n, 0x00020000);
}

// ReLU as one v_max_i32 on the bit pattern (negative floats, -0 included, are negative integers; fmaxf costs two instructions here:
// a canonicalising v_max of the MFMA result in front of the real one).  A NaN with a clear sign bit stays NaN (numpy's maximum does
// the same) and reaches the output's finite_or_clipped guard.
template <int ACT> __device__ __forceinline__ float fwd_act(float v, int kind) {
    if constexpr (ACT == LNR_ACT_RELU) {
        const int b = __builtin_bit_cast(int, v);
        return __builtin_bit_cast(float, b > 0 ? b : 0);
    } else {
        return gact<ACT>(v, kind);
    }
}

// One layer: B operands Bin (KB K blocks x CT column tiles) against the matrix at Wl (row stride S halves, lane offsets koff).
// LAST = false: the activations become Bout, the next layer's B operands.  LAST = true: they are reduced against the output row
// (wo: the lane's entries, rows 16jt + 4g + r) into part[t].
// BIAS: the products of row tile jt start from bias[16jt + 4g .. +3] (bias_lane = the lane's bias + 4g) instead of zero.
// side(k): a piece of independent VALU work issued behind the MFMAs of row tile k - SIDE_BASE (the fused frequency kernels evaluate one
// (sin, cos) slot of the NEXT step's features there: the matrix pipe is busy for ~130 cycles per row tile, and a wave that computes its
// features in one block in front of its MFMAs leaves that time to whatever other wave the SIMD holds)
struct FwdNoSide { __device__ __forceinline__ void operator()(int) const {} };
template <int HT, int ACT, int KB, int S, int CT, bool LAST, bool BIAS, int SIDE_BASE = 0, typename Side = FwdNoSide>
__device__ __forceinline__ void fwd_layer(const f16* Wl, const int (&koff)[F16_KB_MAX], const float* bias_lane, int act, const u32x4 (&Bin)[F16_KB_MAX][CT],
                                          u32x4 (&Bout)[F16_KB_MAX][CT], const float (&wo)[HT][4], float (&part)[CT], Side side = Side()) {
    f16x8 a[2][KB];
    f32x4 Z[2][CT], z0[2];
    auto frags = [&](int jt, f16x8 (&dst)[KB], f32x4& zb) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) dst[kb] = *reinterpret_cast<const f16x8*>(Wl + 16 * jt * S + koff[kb]);
        zb = BIAS ? *reinterpret_cast<const f32x4*>(bias_lane + 16 * jt) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };
    auto finish = [&](int jt, const f32x4 (&z)[CT]) {                    // activations of row tile jt
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            if constexpr (LAST) {
#pragma unroll
                for (int r = 0; r < 4; ++r) part[t] = __builtin_fmaf(wo[jt][r], fwd_act<ACT>(z[t][r], act), part[t]);
                // (pinned to its row: left free, one column tile's whole chain - 32 dependent fmas - was sunk below the layer, behind the
                // other tile's store branch, with no MFMA left to issue beside it)
                asm volatile("" : "+v"(part[t]));
            } else {
                if constexpr (ACT == LNR_ACT_RELU) {
                    Bout[jt >> 1][t][2 * (jt & 1)] = relu_pack_h2(z[t][0], z[t][1]);
                    Bout[jt >> 1][t][2 * (jt & 1) + 1] = relu_pack_h2(z[t][2], z[t][3]);
                } else {
                    Bout[jt >> 1][t][2 * (jt & 1)] = pack_h2(fwd_act<ACT>(z[t][0], act), fwd_act<ACT>(z[t][1], act));
                    Bout[jt >> 1][t][2 * (jt & 1) + 1] = pack_h2(fwd_act<ACT>(z[t][2], act), fwd_act<ACT>(z[t][3], act));
                }
            }
        }
    };
    if constexpr (!LAST && (HT & 1)) {                                    // odd tile count (16 neurons): the upper half of the last block
#pragma unroll
        for (int t = 0; t < CT; ++t) { Bout[HT >> 1][t][2] = 0u; Bout[HT >> 1][t][3] = 0u; }
    }
    frags(0, a[0], z0[0]);
#pragma unroll
    for (int jt = 0; jt < HT; ++jt) {
        if (jt + 1 < HT) frags(jt + 1, a[(jt + 1) & 1], z0[(jt + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);                                // the loads stay in front of this row's MFMAs
#pragma unroll
        for (int t = 0; t < CT; ++t) Z[jt & 1][t] = z0[jt & 1];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int t = 0; t < CT; ++t)
                Z[jt & 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[jt & 1][kb], __builtin_bit_cast(f16x8, Bin[kb][t]), Z[jt & 1][t], 0, 0, 0);
        side(SIDE_BASE + jt);
        if (jt > 0) finish(jt - 1, Z[(jt - 1) & 1]);
    }
    finish(HT - 1, Z[(HT - 1) & 1]);
}

#ifdef LNR_PHASE_TIMING
static __device__ unsigned long long lnr_f16_fwd_phase_cycles[LNR_N_PHASES];
#endif

// FQ: the network's input is the frequency encoding of `src`'s points, evaluated here (lnr_f16_freq.h); featp / m_pad unused.
template <int HT, int ACT, int NH, int KT, int CT, bool FQ = false>
__global__ void __launch_bounds__(LNR_DENSITY_BLOCK, 2)          // two waves per SIMD: <= 256 registers (two workgroups share a CU's LDS)
mlp_forward_f16_gen_kernel(const LnrNetSpec spec, const float* __restrict__ params, const uint32_t* __restrict__ featp, int64_t m_pad,
                           int64_t n_points, const int32_t* __restrict__ n_rays_dev, int n_rays, int n_samples, float* __restrict__ sigma, int32_t* __restrict__ clip_flag,
                           const PointSrc src) {
    extern __shared__ __attribute__((aligned(16))) f16 Ws[];
    using L = FwdLds<HT, NH, KT>;
    static_assert(NH >= 1 && NH <= F16_NH_MAX && KT >= 1 && KT <= F16_KB_MAX, "shape");
    static_assert(L::KBH <= F16_KB_MAX || NH == 1, "256 neurons: one hidden layer");
    constexpr int TS = 16 * CT;                                           // samples per wave step
    PHASE_INIT();
    fwd_fill_weights<HT, NH, KT, FQ>(Ws, params, spec.in_dim, spec.enc_dim, spec.n_frequencies);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int act = spec.activation, enc_pairs = spec.enc_dim / 2;
    const int64_t M = live_samples(n_points, n_rays_dev, n_rays, n_samples);
    const int64_t n_tiles = M > 0 ? (M + TS - 1) / TS : 0;
    const uint32_t plane_bytes = (uint32_t)m_pad * 4u;
    const float* bias_lane = reinterpret_cast<const float*>(Ws + L::N_W) + 4 * g;
    __amdgpu_buffer_rsrc_t rsrc[KT];
#pragma unroll
    for (int kb = 0; kb < KT; ++kb) rsrc[kb] = fwd_block_rsrc(featp, plane_bytes, kb, enc_pairs);
    uint32_t qoff[4];                                                     // byte offsets of the lane's four planes inside a K block, + its column
#pragma unroll
    for (int q = 0; q < 4; ++q) qoff[q] = (uint32_t)(4 * g + q) * plane_bytes + (uint32_t)c * 4u;
    int koff0[F16_KB_MAX], koffh[F16_KB_MAX];
    fwd_frag_offsets<KT, L::S0, L::SWZ0>(c, g, koff0);
    fwd_frag_offsets<(NH > 1 ? L::KBH : 0), L::SH, L::SWZH>(c, g, koffh);
    float wo[HT][4];                                                      // the lane's entries of the output row
#pragma unroll
    for (int jt = 0; jt < HT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) wo[jt][r] = (float)Ws[L::OFF_O + 16 * jt + 4 * g + r];
    // FQ (lnr_f16_freq.h): slot sl of this lane = coordinate sl % 3, frequency 4 (sl / 3) + g; a compile-time number of slots
    constexpr int fq_slots = LNR_FREQ_SLOTS_OF_KT(KT);
    const float fq_pg = __uint_as_float((uint32_t)(127 + g) << 23);      // 2^g
    const bool fq_uni = FQ && ray_uniform(src, (uint32_t)TS) && M % TS == 0;      // (M % TS: the clamp of a ragged last tile would mix two rays)
    const int ns_shift = (src.n_samples > 0 && (src.n_samples & (src.n_samples - 1)) == 0) ? __builtin_ctz((unsigned)src.n_samples) : -1;   // samples per ray a power of two: a shift
    // the features of the NEXT step are in flight while this one goes through the layers
    // (samples past M: their planes are padded to m_pad, whatever they hold only reaches columns that are never stored)
    // FQ: the unit-cube points of a step's samples (x the lane's 2^g) ...
    // The loads and the arithmetic are separate: the main loop requests a step's points one whole step before it turns them into
    // features (phase timers, profiles/r06_fp16_mlp_phases.txt: requested and consumed in one place, the round trip was 35 % of a wave's
    // time with two waves per SIMD to cover it).
    auto fq_request = [&](int64_t tile, RawPoint (&rp)[CT]) __attribute__((always_inline)) {
        uint32_t mm[CT], rr[CT];
        if (fq_uni) {
            // the step's samples lie on ONE ray: its index from wave-uniform operands (a shift or one scalar division instead of a
            // per-lane division for every column tile); nothing to clamp (the sample count is a multiple of the step)
            const uint32_t m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tile * TS));
            const uint32_t ray = ns_shift >= 0 ? m0 >> ns_shift : m0 / (uint32_t)__builtin_amdgcn_readfirstlane(src.n_samples);
#pragma unroll
            for (int t = 0; t < CT; ++t) { mm[t] = m0 + 16u * t + (uint32_t)c; rr[t] = ray; }
        } else {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                int64_t m = tile * TS + 16 * t + c;
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
        for (int t = 0; t < CT; ++t) {
            const uint32_t oo = is_pts ? mm[t] * 12u : rr[t] * (uint32_t)(LNR_RAY_STRIDE * 4);
            const uint32_t od = is_pts ? 0u : oo + 12u, oz = is_pts ? 0u : mm[t] * 4u;
            rp[t].o0 = ld32_at<float>(b_od, oo); rp[t].o1 = ld32_at<float>(b_od, oo + 4u); rp[t].o2 = ld32_at<float>(b_od, oo + 8u);
            rp[t].d0 = ld32_at<float>(b_od, od); rp[t].d1 = ld32_at<float>(b_od, od + 4u); rp[t].d2 = ld32_at<float>(b_od, od + 8u);
            rp[t].z = ld32_at<float>(b_z, oz);
        }
    };
    auto fq_unit = [&](const RawPoint (&rp)[CT], float (&xu)[CT][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            unit_point(src, rp[t], xu[t]);
#pragma unroll
            for (int d = 0; d < 3; ++d) xu[t][d] *= fq_pg;              // the lane's share of the frequency (exact)
        }
    };
    auto fq_points = [&](int64_t tile, float (&xu)[CT][3]) __attribute__((always_inline)) {
        RawPoint rp[CT];
        fq_request(tile, rp);
        fq_unit(rp, xu);
    };
    // ... and ONE slot of their features (both column tiles): the unit of work interleaved with the previous step's MFMAs
    auto fq_slot = [&](int sl, const float (&xu)[CT][3], u32x4 (&x)[F16_KB_MAX][CT]) __attribute__((always_inline)) {
        if (sl < fq_slots) {                                           // (compile-time: sl is a literal of the unrolled callers)
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                float d0, d1;
                // pinned between two empty volatile statements: nothing orders this arithmetic otherwise, and instruction selection
                // gathered the slots of a whole step in front of its first MFMA (the sched_barriers only bind the later scheduler)
                float y = xu[t][sl % 3];
                asm volatile("" : "+v"(y));
                uint32_t v = freq_pair<false>(y * lnr_freq_slot_scale(sl), 0.0f, d0, d1);
                asm volatile("" : "+v"(v));
                x[sl >> 2][t][sl & 3] = v;
            }
        }
    };
    // the same slot from the coordinate's shared phase part (freq_base: three instructions per coordinate instead of per slot)
    auto fq_slot_shared = [&](int sl, const float (&xu)[CT][3], const FreqBase (&fb)[CT][3], u32x4 (&x)[F16_KB_MAX][CT]) __attribute__((always_inline)) {
        if (sl < fq_slots) {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
#if LNR_FREQ_HW_SIN
                float y = xu[t][sl % 3];
                asm volatile("" : "+v"(y));                               // (pinned: see fq_slot)
                uint32_t v = freq_pair_scaled(y, fb[t][sl % 3], lnr_freq_slot_scale(sl));
                asm volatile("" : "+v"(v));
                x[sl >> 2][t][sl & 3] = v;
#else
                (void)fb;
                float d0, d1;
                x[sl >> 2][t][sl & 3] = freq_pair<false>(xu[t][sl % 3] * lnr_freq_slot_scale(sl), 0.0f, d0, d1);
#endif
            }
        } else if (sl < 4 * KT) {                                         // a dead slot of a live K block: zero (its weights are zero too)
#pragma unroll
            for (int t = 0; t < CT; ++t) x[sl >> 2][t][sl & 3] = 0u;
        }
    };
    auto load_tile = [&](int64_t tile, u32x4 (&x)[F16_KB_MAX][CT]) __attribute__((always_inline)) {
        if constexpr (FQ) {
            float xu[CT][3];
            fq_points(tile, xu);
#pragma unroll
            for (int kb = 0; kb < F16_KB_MAX; ++kb)
#pragma unroll
                for (int t = 0; t < CT; ++t) x[kb][t] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int sl = 0; sl < 4 * KT; ++sl) fq_slot(sl, xu, x);
            return;
        }
        const uint32_t m0 = (uint32_t)(tile * TS) * 4u;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
#pragma unroll
            for (int kb = 0; kb < F16_KB_MAX; ++kb) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    x[kb][t][q] = kb < KT ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc[kb < KT ? kb : 0], (int)(qoff[q] + m0) + 64 * t, 0, 0) : 0u;
            }
        }
    };
    // (always inlined: with a run-time activation the body is large enough for the inliner to leave it a FUNCTION, and a call passes
    // the operand arrays through scratch memory - 500 bytes per lane)
    // side: FwdNoSide, or the fused frequency kernels' job - slot k of the NEXT step's features behind the MFMAs of row tile k
    const __amdgpu_buffer_rsrc_t sig_rsrc = __builtin_amdgcn_make_buffer_rsrc(sigma, 0, (int)(M * 4), 0x00020000);   // (M <= 2^25: lnr_density_forward)
    auto run_tile = [&](int64_t tile, const u32x4 (&x)[F16_KB_MAX][CT], auto side) __attribute__((always_inline)) {
        float part[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) part[t] = 0.0f;
        u32x4 B1[F16_KB_MAX][CT], B2[F16_KB_MAX][CT];
        using SideT = decltype(side);
        if constexpr (NH == 1) {
            fwd_layer<HT, ACT, KT, L::S0, CT, true, true, 0, SideT>(Ws, koff0, bias_lane, act, x, B1, wo, part, side);
        } else if constexpr (NH == 2) {
            fwd_layer<HT, ACT, KT, L::S0, CT, false, true, 0, SideT>(Ws, koff0, bias_lane, act, x, B1, wo, part, side);
            PHASE(2);
            fwd_layer<HT, ACT, L::KBH, L::SH, CT, true, false, HT, SideT>(Ws + L::OFF_H, koffh, bias_lane, act, B1, B2, wo, part, side);
            PHASE(3);
        } else {
            fwd_layer<HT, ACT, KT, L::S0, CT, false, true, 0, SideT>(Ws, koff0, bias_lane, act, x, B1, wo, part, side);
            fwd_layer<HT, ACT, L::KBH, L::SH, CT, false, false, HT, SideT>(Ws + L::OFF_H, koffh, bias_lane, act, B1, B2, wo, part, side);
            fwd_layer<HT, ACT, L::KBH, L::SH, CT, true, false, 2 * HT, SideT>(Ws + L::OFF_H + L::H * L::SH, koffh, bias_lane, act, B2, B1, wo, part, side);
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            float v = part[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int64_t m = tile * TS + 16 * t + c;
            // ONE unconditional store per column tile (lanes with nothing to store get an offset beyond the buffer: dropped by the
            // range check).  Loads and stores share one in-order counter (vmcnt); behind a branch the compiler cannot count the store,
            // and the wait for the next step's points - requested a whole step earlier, in front of it - became a wait for everything
            // outstanding, i.e. for this store's acknowledgement: a memory round trip per step with only the SIMD's other wave to
            // cover it (30 % of a wave's time, profiles/r06_fp16_mlp_phases.txt).
            const bool stored = g == 0 && m < M;
            if (stored) v = finite_or_clipped<true>(v, clip_flag);         // (the guard counts stored values only; the STORE sits behind the join)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), sig_rsrc, stored ? (int)((uint32_t)m * 4u) : (int)0x80000000u, 0, 0);
        }
        PHASE(4);
    };
    const int64_t stride = (int64_t)gridDim.x * nw;
    int64_t tile = (int64_t)blockIdx.x * nw + wave;
    u32x4 xa[F16_KB_MAX][CT], xb[F16_KB_MAX][CT];
    if (tile < n_tiles) load_tile(tile, xa);
    PHASE(0);
    if constexpr (FQ && ACT >= 0) {
        // fused frequency encoding: the NEXT step's points are requested in front of this step's layers, and its features are evaluated
        // slot by slot BEHIND the MFMAs of this step's row tiles (HT x NH of them; what does not fit there follows the last layer)
        constexpr int N_SIDE = HT * NH;
        float xu[CT][3];
        FreqBase fb[CT][3];                                               // the shared phase part of every coordinate (lnr_f16_freq.h)
        auto fq_bases = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int d = 0; d < 3; ++d) fb[t][d] = freq_base(xu[t][d]);
        };
        RawPoint rp[CT];                                                  // the points of step tile + stride, requested one step earlier
        if (tile < n_tiles) fq_request(tile + stride < n_tiles ? tile + stride : tile, rp);
        for (; tile < n_tiles; tile += 2 * stride) {
            const int64_t t1 = tile + stride, t2 = tile + 2 * stride, t3 = tile + 3 * stride;
            fq_unit(rp, xu);
            fq_request(t2 < n_tiles ? t2 : tile, rp);                      // a whole step ahead of its use (the output stores in between are countable: run_tile)
            fq_bases();
            PHASE(1);
            // (every dword of the live K blocks is written by its slot - dead slots as zero - so the buffers need no clearing)
            run_tile(tile, xa, [&](int k) __attribute__((always_inline)) { fq_slot_shared(k, xu, fb, xb); });
#pragma unroll
            for (int sl = N_SIDE; sl < 4 * KT; ++sl) fq_slot_shared(sl, xu, fb, xb);
            PHASE(5);
            if (t1 >= n_tiles) break;
            fq_unit(rp, xu);
            fq_request(t3 < n_tiles ? t3 : t1, rp);
            fq_bases();
            PHASE(1);
            run_tile(t1, xb, [&](int k) __attribute__((always_inline)) { fq_slot_shared(k, xu, fb, xa); });
#pragma unroll
            for (int sl = N_SIDE; sl < 4 * KT; ++sl) fq_slot_shared(sl, xu, fb, xa);
            PHASE(5);
        }
    } else if constexpr (ACT >= 0) {
        for (; tile < n_tiles; tile += 2 * stride) {                      // two steps per trip: the feature buffers alternate
            const int64_t t1 = tile + stride, t2 = tile + 2 * stride;
            load_tile(t1 < n_tiles ? t1 : tile, xb);                      // unconditional (clamped): a static number of loads in flight
            run_tile(tile, xa, FwdNoSide());
            if (t1 >= n_tiles) break;
            load_tile(t2 < n_tiles ? t2 : t1, xa);
            run_tile(t1, xb, FwdNoSide());
        }
    } else {
        // run-time activation: ONE copy of the step body (every activation of it carries the whole switch over the libm
        // evaluations: two copies were 90 k instructions per kernel), the prefetched features copied instead of alternated
        for (; tile < n_tiles; tile += stride) {
            const int64_t t1 = tile + stride;
            load_tile(t1 < n_tiles ? t1 : tile, xb);
            run_tile(tile, xa, FwdNoSide());
#pragma unroll
            for (int kb = 0; kb < F16_KB_MAX; ++kb)
#pragma unroll
                for (int t = 0; t < CT; ++t) xa[kb][t] = xb[kb][t];
        }
    }
    PHASE_FLUSH(lnr_f16_fwd_phase_cycles, 0);
}
