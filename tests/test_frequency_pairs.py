"""The argument behind freq_forward_h16_kernel / freq_backward_kernel (loner_amd/csrc/lnr_encode.hip), restated in float32 numpy: ONE range
reduction per (sin, cos) pair of the frequency encoding reproduces the reference's two features sin(ph), sin(rn(ph + pi/2)) with
ph = rn(rn(x 2^f) pi) (oracle/encoding.py; tinycudann frequency.h) to ~1e-7, so that their fp16 roundings agree except on a rounding
boundary.  No GPU."""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    return f32(np.float64(a) * np.float64(b) + np.float64(c))          # (double rounding is negligible at this scale)


def sincos_f32(ph):
    """sincos_f32 of lnr_encode.hip: k = rint(ph 2/pi), three-term Cody-Waite, minimax polynomials on [-pi/4, pi/4], quadrant"""
    k = np.rint(ph * f32(0.636619772367581343)).astype(f32)
    r = fma(-k, f32(1.57079637050628662109375), ph)
    r = fma(-k, f32(-4.37113900018624283e-8), r)
    r = fma(-k, f32(-1.7151245100059e-15), r).astype(f32)
    z = (r * r).astype(f32)
    sp = fma(z, f32(-1.9515295891e-4), f32(8.3321608736e-3)); sp = fma(z, sp, f32(-1.6666654611e-1))
    s = fma((z * r).astype(f32), sp, r)
    cp = fma(z, f32(2.443315711809948e-5), f32(-1.388731625493765e-3)); cp = fma(z, cp, f32(4.166664568298827e-2))
    c = fma((z * z).astype(f32), cp, fma(z, f32(-0.5), f32(1.0)))
    q = k.astype(np.int64)
    a = np.where(q & 1, c, s); b = np.where(q & 1, s, c)
    return np.where(q & 2, -a, a).astype(f32), np.where((q + 1) & 2, -b, b).astype(f32)


def test_pair_reduction_reproduces_both_features():
    rng = np.random.default_rng(0)
    x = rng.random(100000).astype(f32)
    PI, H = f32(np.pi), f32(np.pi / 2)
    worst_s = worst_c = 0.0
    differ = total = 0
    for f in range(12):
        ph = ((x * f32(2 ** f)).astype(f32) * PI).astype(f32)
        s, c = sincos_f32(ph)
        h = (ph + H).astype(f32)                                         # the reference's second phase
        bb = (h - ph).astype(f32)
        e = ((ph - (h - bb).astype(f32)).astype(f32) + (H - bb).astype(f32)).astype(f32)      # rounding error of ph + fl(pi/2), exact (TwoSum)
        d = (f32(4.371139000186243e-8) - e).astype(f32)                  # fl(pi/2) - pi/2 = +4.37e-8
        c2 = fma(-d, s, c)                                               # cos(ph + d) to first order
        ref_s, ref_c = np.sin(ph.astype(np.float64)), np.sin(h.astype(np.float64))
        worst_s = max(worst_s, float(np.abs(s - ref_s).max())); worst_c = max(worst_c, float(np.abs(c2 - ref_c).max()))
        differ += int((s.astype(np.float16) != ref_s.astype(f32).astype(np.float16)).sum() +
                      (c2.astype(np.float16) != ref_c.astype(f32).astype(np.float16)).sum())
        total += 2 * len(x)
    assert worst_s < 2e-7 and worst_c < 2e-7, (worst_s, worst_c)
    assert differ < 2e-4 * total, (differ, total)                        # fp16 features: the same value except on a rounding boundary


def test_hardware_sine_route_reproduces_both_features():
    """freq_pair with LNR_FREQ_HW_SIN (lnr_f16_freq.h): sin / cos(pi y) from v_sin_f32 / v_cos_f32 of fract(y / 2) (revolutions; modelled
    here as the exact function + the 1.25e-7 the instructions were measured at, profiles/r06_valu_rate.txt) and the first-order correction
    for dl = ph - pi y, recovered exactly from one fma - against the reference's sin(ph), sin(rn(ph + pi/2))."""
    rng = np.random.default_rng(2)
    x = rng.random(100000).astype(f32)
    PI, H = f32(np.pi), f32(np.pi / 2)
    worst_s = worst_c = 0.0
    for f in range(12):                                                              # (the fused kernels take n_frequencies <= 12)
        y = (x * f32(2 ** f)).astype(f32)
        ph = (y * PI).astype(f32)
        e1 = (np.float64(y) * np.float64(PI) - np.float64(ph)).astype(f32)            # fma(y, PI, -ph): exact
        assert np.array_equal(np.float64(e1), np.float64(y) * np.float64(PI) - np.float64(ph))
        dl = fma(y, f32(8.742278000372485e-8), -e1)
        r = ((y * f32(0.5)) - np.floor(y * f32(0.5))).astype(f32)                       # v_fract_f32: exact
        noise = rng.uniform(-1.25e-7, 1.25e-7, size=(2, len(x)))
        S = (np.sin(2 * np.pi * np.float64(r)) + noise[0]).astype(f32)
        C = (np.cos(2 * np.pi * np.float64(r)) + noise[1]).astype(f32)
        s, c = fma(C, dl, S), fma(-S, dl, C)
        h = (ph + H).astype(f32)
        e = (H - (h - ph).astype(f32)).astype(f32)                                       # Fast2Sum: exact for ph >= fl(pi/2), else within 1.2e-7
        d = (f32(4.371139000186243e-8) - e).astype(f32)
        c2 = fma(-d, s, c)
        worst_s = max(worst_s, float(np.abs(s - np.sin(ph.astype(np.float64))).max()))
        worst_c = max(worst_c, float(np.abs(c2 - np.sin(h.astype(np.float64))).max()))
    assert worst_s < 4e-7 and worst_c < 4e-7, (worst_s, worst_c)


def test_sampler_prefix_sums_are_exact_in_float64():
    """sample_occ_kernel's parallel cumsum (lnr_sampler.hip) relies on the float64 prefix sums of the pdf being exact, hence independent of
    the order of the additions: a sequential float64 cumsum (torch.cumsum on the CPU) equals a chunked one bit for bit."""
    rng = np.random.default_rng(1)
    for K in (62, 254, 1022, 4094):
        w = (rng.random(K).astype(f32) ** 8).astype(f32) + f32(1e-5)     # many nearly empty bins, like a trained occupancy grid
        pdf = (w / f32(np.float32(w.astype(np.float64).sum()))).astype(f32)
        seq = np.cumsum(pdf.astype(np.float64))
        chunk = (K + 63) // 64
        out = np.empty(K)
        sums = [pdf[i:i + chunk].astype(np.float64).sum() for i in range(0, K, chunk)]
        pre = np.concatenate([[0.0], np.cumsum(sums)[:-1]])
        for j, i in enumerate(range(0, K, chunk)):
            out[i:i + chunk] = pre[j] + np.cumsum(pdf[i:i + chunk].astype(np.float64))
        assert np.array_equal(seq.astype(f32), out.astype(f32)) and np.array_equal(seq, out)


def test_shared_phase_part_scales_exactly():
    """freq_base / freq_pair_scaled (lnr_f16_freq.h): the three slots of a coordinate evaluate y = xg S with S = 1, 2^4, 2^8 (a literal
    power of two).  Scaling by a power of two commutes with every rounding of the phase part, so ph(y) = S ph(xg), dl(y) = S dl(xg)
    and the argument of v_fract_f32 is the same number - the per-coordinate form is the per-slot form bit for bit (no overflow or
    underflow: xg in [0, 8), S <= 2^8).  The fma is modelled in exact rational arithmetic here (double is not enough to tell)."""
    from fractions import Fraction
    rng = np.random.default_rng(7)
    PI, DPI = f32(np.pi), f32(8.742278000372485e-8)

    def rn(q):                                                            # round a Fraction to float32, to nearest even
        v = f32(float(q))                                                 # (float(q) is correctly rounded to double; a second rounding to
        lo, hi = np.nextafter(v, f32(-np.inf)), np.nextafter(v, f32(np.inf))   # float32 may be off by one ulp: pick the nearest of three)
        best = min((lo, v, hi), key=lambda c: (abs(Fraction(float(c)) - q), int(np.float32(c).view(np.uint32)) & 1))
        return f32(best)

    for _ in range(400):
        xg = f32(rng.random() * 2 ** rng.integers(0, 4))
        ph0 = f32(xg * PI)
        e10 = rn(Fraction(float(xg)) * Fraction(float(PI)) - Fraction(float(ph0)))
        dl0 = rn(Fraction(float(xg)) * Fraction(float(DPI)) - Fraction(float(e10)))
        for j in (0, 1, 2):
            S = f32(2 ** (4 * j))
            y = f32(xg * S)
            ph = f32(y * PI)
            e1 = rn(Fraction(float(y)) * Fraction(float(PI)) - Fraction(float(ph)))
            dl = rn(Fraction(float(y)) * Fraction(float(DPI)) - Fraction(float(e1)))
            assert ph == f32(ph0 * S) and dl == f32(dl0 * S)
            assert f32(y * f32(0.5)) == f32(xg * f32(0.5 * float(S)))
