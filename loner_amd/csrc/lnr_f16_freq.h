// Frequency encoding computed INSIDE the fp16-mode MLP kernels (lnr_f16_fwd_kernel.h / lnr_f16_bwd_kernel.h with FQ = true): the
// reference hands encoding and MLP to ONE tinycudann module (src/models/nerf_tcnn.py:35-38, called :63-72), and sin/cos of a point is
// register arithmetic - no feature planes and no d_feature planes cross HBM for `otype: Frequency` (rounds 2-5: 72 half2 pair planes
// written by freq_forward_h16_kernel and read back by the MLP kernels, fp32 d_feature planes written by the backward and read back by
// freq_backward_kernel: 0.3 + 0.6 GB per direction at 2.1 M samples).
//
// Who computes what.  The B operand of v_mfma_f32_16x16x32_f16 gives lane (c = lane & 15, g = lane >> 4) the K positions 8g .. 8g+7 of
// a 32-wide K block for sample c: four (sin, cos) pairs per block, q = 0..3, i.e. SLOT sl = 4 kb + q of the lane.  The first layer's
// input order is ours to choose (its weight columns are permuted to match when they are converted into LDS, its gradient is permuted
// back when the slab is written), so the slots are dealt such that every lane evaluates pairs of ONE coordinate wherever it can:
//   lane groups g = 0, 1, 2:  coordinate g, frequencies f = sl          for sl < NSL = ceil(3 nf / 4)
//   lane group  g = 3:        the frequencies NSL .. nf-1 the others leave over, coordinate sl / REM, f = NSL + sl % REM   (REM = nf - NSL,
//                             sl < 3 REM <= NSL)
// nf = 12: nine pairs per lane, three K blocks (96 inputs instead of the 128 the plane kernels pad 72 + 8 to), no idle lane.
// Feature values: exactly those of freq_forward_h16_kernel (one Cody-Waite reduction per pair, the second feature through the
// recovered rounding error of ph + fl(pi/2)): tests/test_frequency_pairs.py is the numpy statement of that arithmetic.
#pragma once
#include "lnr_encoding.h"

__host__ __device__ inline int lnr_freq_nsl(int nf) { return (3 * nf + 3) / 4; }
// K blocks of the fused first layer (0: not supported by the fused kernels - more than 16 slots per lane)
__host__ __device__ inline int lnr_freq_kt(int nf) { const int n = lnr_freq_nsl(nf); return n <= 8 ? 2 : (n <= 12 ? 3 : 0); }
// the feature (tinycudann order [dim][frequency][sin, cos]) that sits at K position p of the fused first layer; -1: padding
__host__ __device__ inline int lnr_freq_feature_at(int p, int nf) {
    const int nsl = lnr_freq_nsl(nf), rem = nf - nsl;
    const int kb = p >> 5, g = (p >> 3) & 3, q = (p >> 1) & 3, h = p & 1, sl = 4 * kb + q;
    int dim, f;
    if (g < 3) {
        if (sl >= nsl) return -1;
        dim = g; f = sl;
    } else {
        if (sl >= 3 * rem) return -1;
        dim = sl / rem; f = nsl + sl % rem;
    }
    return dim * 2 * nf + 2 * f + h;
}

// one lane's view of its slots
struct FreqLane {
    int nsl, rem;          // wave-uniform
    bool g3;               // lane group 3
    int g;
    __device__ __forceinline__ void init(int nf, int g_) {
        nsl = __builtin_amdgcn_readfirstlane(lnr_freq_nsl(nf));
        rem = __builtin_amdgcn_readfirstlane(nf - lnr_freq_nsl(nf));
        g = g_; g3 = g_ == 3;
    }
    // coordinate of slot sl (x: unit-cube point of the lane's sample) and 2^f; a dead slot gets 2^f = 0 (features sin 0, cos 0 against
    // zero weight columns)
    __device__ __forceinline__ void slot(int sl, const float x[3], float xg, float& xs, float& mult) const {
        const int d3 = rem > 0 ? sl / rem : 0, f3 = nsl + (rem > 0 ? sl % rem : 0);          // scalar arithmetic (sl is a literal after unrolling)
        const float x3 = d3 == 0 ? x[0] : (d3 == 1 ? x[1] : x[2]);
        xs = g3 ? x3 : xg;
        const float m012 = __uint_as_float((uint32_t)(127 + sl) << 23), m3 = sl < 3 * rem ? __uint_as_float((uint32_t)(127 + f3) << 23) : 0.0f;
        mult = g3 ? m3 : m012;
    }
    // which of the three coordinates slot sl of THIS lane belongs to (for the input gradient)
    __device__ __forceinline__ int dim_of(int sl) const { return g3 ? (rem > 0 ? sl / rem : 0) : g; }
};

// the (sin, cos-like) feature pair of one slot and, for the backward, d(pair)/d(xs): ds = dph * cos(ph), dc = -dph * (sin(ph) + d cos(ph))
// with dph = 2^f pi (freq_backward_kernel's arithmetic)
template <bool DERIV>
__device__ __forceinline__ uint32_t freq_pair(float xs, float mult, float& ds, float& dc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float ph = lnr_mul_rn(lnr_mul_rn(xs, mult), LNR_PI_F);
    float s, c;
    sincos_f32(ph, &s, &c);
    const float h = lnr_add_rn(ph, LNR_PI_2_F);                              // the reference's second phase; e = its rounding error
    const float bb = lnr_add_rn(h, -ph);
    const float e = lnr_add_rn(lnr_add_rn(ph, -lnr_add_rn(h, -bb)), lnr_add_rn(LNR_PI_2_F, -bb));
    const float d = 4.371139000186243e-8f - e;
    const float c2 = __builtin_fmaf(-d, s, c);
    if constexpr (DERIV) {
        const float dph = mult * LNR_PI_F;
        ds = dph * c;
        dc = -dph * __builtin_fmaf(d, c, s);
    }
    return __builtin_bit_cast(uint32_t, h2{(_Float16)s, (_Float16)c2});
}
