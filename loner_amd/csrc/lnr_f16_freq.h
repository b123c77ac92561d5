// Frequency encoding computed INSIDE the fp16-mode MLP kernels (lnr_f16_fwd_kernel.h / lnr_f16_bwd_kernel.h with FQ = true): the
// reference hands encoding and MLP to ONE tinycudann module (src/models/nerf_tcnn.py:35-38, called :63-72), and sin/cos of a point is
// register arithmetic - no feature planes and no d_feature planes cross HBM for `otype: Frequency` (rounds 2-5: 72 half2 pair planes
// written by freq_forward_h16_kernel and read back by the MLP kernels, fp32 d_feature planes written by the backward and read back by
// freq_backward_kernel: 0.3 + 0.6 GB per direction at 2.1 M samples).
//
// Who computes what.  The B operand of v_mfma_f32_16x16x32_f16 gives lane (c = lane & 15, g = lane >> 4) the K positions 8g .. 8g+7 of
// a 32-wide K block for sample c: four (sin, cos) pairs per block, q = 0..3, i.e. SLOT sl = 4 kb + q of the lane.  The first layer's
// input order is ours to choose (its weight columns are permuted to match when they are converted into LDS, its gradient is permuted
// back when the slab is written), so the slots are dealt such that NOTHING about a slot depends on the lane but one factor:
//   slot sl of lane group g  =  coordinate sl % 3, frequency f = 4 (sl / 3) + g        (sl < 3 J, J = ceil(nf / 4); f < nf)
// i.e. lane group g takes the frequencies = g (mod 4) of all three coordinates: the coordinate and 2^(4 (sl / 3)) are literals of the
// unrolled code, the lane's share is the factor 2^g on its point (exact).  A first version dealt coordinate g to lane group g and the
// left-over frequencies to group 3: six selects per slot and a table of multipliers (437 v_cndmask + 314 lane moves of spilled scalars
// in the forward kernel - slower than reading the planes).  nf = 12: nine pairs per lane, three K blocks (96 inputs instead of the 128
// the plane kernels pad 72 + 8 to), no idle lane.  Slots whose frequency is >= nf (nf not a multiple of 4) are evaluated like the
// others - finite values against zero weight columns; their gradient rows are zero.
// Feature values: those of freq_forward_h16_kernel to ~2e-7 (one reduction per pair, the second feature through the recovered rounding
// error of ph + fl(pi/2)): tests/test_frequency_pairs.py is the numpy statement of both routes' arithmetic.
#pragma once
#include "lnr_encoding.h"

__host__ __device__ inline int lnr_freq_slots(int nf) { return 3 * ((nf + 3) / 4); }          // live slots per lane
// K blocks of the fused first layer: 2 (six slots per lane evaluated, nf <= 8) or 3 (nine, nf <= 12); 0: not fused (nf > 12: the
// plane kernels - above 2^11 the first-order corrections of freq_pair no longer hold fp32 accuracy)
__host__ __device__ inline int lnr_freq_kt(int nf) { const int n = lnr_freq_slots(nf); return n <= 6 ? 2 : (n <= 9 ? 3 : 0); }
#define LNR_FREQ_SLOTS_OF_KT(KT) ((KT) == 3 ? 9 : 6)          /* slots a kernel with KT blocks evaluates (dead ones meet zero weights) */
// the feature (tinycudann order [dim][frequency][sin, cos]) that sits at K position p of the fused first layer; -1: padding
__host__ __device__ inline int lnr_freq_feature_at(int p, int nf) {
    const int kb = p >> 5, g = (p >> 3) & 3, q = (p >> 1) & 3, h = p & 1, sl = 4 * kb + q;
    const int dim = sl % 3, f = 4 * (sl / 3) + g;
    if (sl >= lnr_freq_slots(nf) || f >= nf) return -1;
    return dim * 2 * nf + 2 * f + h;
}
// 2^(4 (sl / 3)): the literal part of slot sl's frequency
__device__ __forceinline__ float lnr_freq_slot_scale(int sl) { return __uint_as_float((uint32_t)(127 + 4 * (sl / 3)) << 23); }

// LNR_FREQ_HW_SIN = 1 (default): sin / cos of the phase from the hardware's v_sin_f32 / v_cos_f32.  Measured on MI355X
// (tools/valu_rate.hip, profiles/r06_valu_rate.txt): max abs error 1.25e-7 over [-1, 1] revolutions - as good as the minimax
// polynomials - at 7.2 cycles per wave instruction, against ~25 instructions (~70 cycles, four of them selects) for the Cody-Waite
// route.  The instructions take REVOLUTIONS: with y = x 2^f (exact) the reference's phase ph = rn(y fl(pi)) is
//   ph = pi y + dl,   dl = y (fl(pi) - pi) - e1,   e1 = y fl(pi) - ph  (the product's rounding error, exact from one fma),
// so sin(ph) = sin(pi y) cos(dl) + cos(pi y) sin(dl) with sin(pi y) = v_sin(fract(y / 2)) - the reduction is an exact v_fract, no
// Cody-Waite - and |dl| < 4.2e-4 for f <= 11 (first order; the fused kernels take nf <= 12).  0: the polynomial route of sincos_f32 (A/B, parity checks).
#ifndef LNR_FREQ_HW_SIN
#define LNR_FREQ_HW_SIN 1
#endif

// The slot-independent part of a phase.  Slot sl of a lane evaluates y = xg S with xg = x 2^g (the lane's share) and S = 2^(4 (sl / 3)) a
// literal; scaling by a power of two commutes with every rounding below, so ph(y) = S ph(xg) and dl(y) = S dl(xg) EXACTLY: the three
// instructions that produce them are shared by the three slots of a coordinate, a slot scales them (nothing at all for S = 1).
struct FreqBase { float ph0, dl0; };
__device__ __forceinline__ FreqBase freq_base(float xg) {
    FreqBase b;
    b.ph0 = lnr_mul_rn(xg, LNR_PI_F);
    const float e1 = __builtin_fmaf(xg, LNR_PI_F, -b.ph0);
    b.dl0 = __builtin_fmaf(xg, 8.742278000372485e-8f, -e1);
    return b;
}
// freq_pair<false>(xg S, ...) from the shared part (bit-identical; LNR_FREQ_HW_SIN route only)
__device__ __forceinline__ uint32_t freq_pair_scaled(float xg, const FreqBase& b, float S) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float ph = b.ph0 * S, dl = b.dl0 * S;                                // (exact)
    const float r = __builtin_amdgcn_fractf(xg * (0.5f * S));
    const float Sn = __builtin_amdgcn_sinf(r), Cs = __builtin_amdgcn_cosf(r);
    const float s = __builtin_fmaf(Cs, dl, Sn);
    const float c = __builtin_fmaf(-Sn, dl, Cs);
    const float h = lnr_add_rn(ph, LNR_PI_2_F);
    const float e = lnr_add_rn(LNR_PI_2_F, -lnr_add_rn(h, -ph));
    const float d = 4.371139000186243e-8f - e;
    const float c2 = __builtin_fmaf(-d, s, c);
    return __builtin_bit_cast(uint32_t, h2{(_Float16)s, (_Float16)c2});
}

// the (sin, cos-like) feature pair of one slot from y = x 2^f (exact, f <= 11) and, for the backward, d(pair)/dx: ds = dph cos(ph),
// dc = -dph (sin(ph) + d cos(ph)) with dph = 2^f pi (freq_backward_kernel's arithmetic)
template <bool DERIV>
__device__ __forceinline__ uint32_t freq_pair(float y, float dph, float& ds, float& dc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float ph = lnr_mul_rn(y, LNR_PI_F);
    float s, c;
#if LNR_FREQ_HW_SIN
    const float e1 = __builtin_fmaf(y, LNR_PI_F, -ph);
    const float dl = __builtin_fmaf(y, 8.742278000372485e-8f, -e1);          // ph - pi y: |dl| < 4.2e-4, the dropped dl^2 / 2 < 9e-8
    const float r = __builtin_amdgcn_fractf(y * 0.5f);
    const float S = __builtin_amdgcn_sinf(r), C = __builtin_amdgcn_cosf(r);
    s = __builtin_fmaf(C, dl, S);
    c = __builtin_fmaf(-S, dl, C);
#else
    sincos_f32(ph, &s, &c);
#endif
    // the reference's second phase h = rn(ph + fl(pi/2)) and e = its rounding error, by Fast2Sum (three operations): exact whenever
    // |ph| >= fl(pi/2), and otherwise (h < pi: half an ulp is 1.2e-7) off by at most that - the size of v_sin_f32's own error; Knuth's
    // branch-free TwoSum, exact everywhere, is six (freq_forward_h16_kernel, whose planes nf > 12 still use)
    const float h = lnr_add_rn(ph, LNR_PI_2_F);
    const float e = lnr_add_rn(LNR_PI_2_F, -lnr_add_rn(h, -ph));
    const float d = 4.371139000186243e-8f - e;
    const float c2 = __builtin_fmaf(-d, s, c);                               // cos(ph + d), |d| < 2.5e-4
    if constexpr (DERIV) {
        ds = dph * c;
        dc = -dph * __builtin_fmaf(d, c, s);
    }
    return __builtin_bit_cast(uint32_t, h2{(_Float16)s, (_Float16)c2});
}
