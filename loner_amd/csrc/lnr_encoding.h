// Input encodings of the density network (multiresolution hash grid / frequency), device helpers shared by the
// level-major encode kernels (lnr_encode.hip).  Semantics = oracle/network.py (tinycudann's published HashGrid /
// Frequency encodings, which the reference configures at src/models/nerf_tcnn.py:35-38).
#pragma once
#include "lnr_common.h"
#include "lnr_density_api.h"

#define PRIME_Y 2654435761u
#define PRIME_Z 805459861u
#define LNR_PI_F 3.14159265358979323846f
#define LNR_PI_2_F 1.57079632679489661923f


__device__ __forceinline__ int64_t live_points(const PointSrc& s) {
    if (s.pts) return s.n_points;
    return (int64_t)lnr_live_rays(s.n_rays, s.n_rays_dev) * s.n_samples;
}

// Point m as loaded (no arithmetic yet, so the loads can be issued a whole iteration ahead of their use).
// pts mode: o = xyz; rays mode: o = origin, d = direction, z = depth.  Scalar fields, all always assigned: an array
// written differently by the two modes ends up in scratch memory.
struct RawPoint { float o0, o1, o2, d0, d1, d2, z; };

__device__ __forceinline__ void load_raw_point(const PointSrc& s, int64_t m, RawPoint& r) {
    if (s.pts) {
        r.o0 = s.pts[3 * m + 0]; r.o1 = s.pts[3 * m + 1]; r.o2 = s.pts[3 * m + 2];
        r.d0 = 0.0f; r.d1 = 0.0f; r.d2 = 0.0f; r.z = 0.0f;
    } else {
        const float* ray = s.rays + (m / s.n_samples) * LNR_RAY_STRIDE;
        r.o0 = ray[0]; r.o1 = ray[1]; r.o2 = ray[2]; r.d0 = ray[3]; r.d1 = ray[4]; r.d2 = ray[5];
        r.z = s.z[m];
    }
}

// unit-cube coordinates
__device__ __forceinline__ void unit_point(const PointSrc& s, const RawPoint& r, float x[3]) {
    float p[3] = {r.o0, r.o1, r.o2};
    if (!s.pts) {                    // o + d*z, as the reference rounds it
        p[0] = lnr_add_rn(r.o0, lnr_mul_rn(r.d0, r.z));
        p[1] = lnr_add_rn(r.o1, lnr_mul_rn(r.d1, r.z));
        p[2] = lnr_add_rn(r.o2, lnr_mul_rn(r.d2, r.z));
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = lnr_mul_rn(lnr_add_rn(p[d], 1.0f), 0.5f);   // (xyz+1)/2 rounded like the reference (no fma)
}

__device__ __forceinline__ void load_unit_point(const PointSrc& s, int64_t m, float x[3]) {
    RawPoint r;
    load_raw_point(s, m, r);
    unit_point(s, r, x);
}

// ------------------------------------------------------------------------------------------------
// multiresolution hash grid: one level, one point
// ------------------------------------------------------------------------------------------------
struct LevelCell {
    uint32_t base[3];   // integer cell
    float frac[3];
    float scale;
    uint32_t res, size, offset, hashed;
};

// Per-level geometry lives in LDS (5 x 32 words at the start of the dynamic LDS block): the level a
// lane works on depends on its lane group, and indexing kernel arguments by a VGPR would force the
// whole spec struct into scratch memory.
#define LNR_LV_WORDS (5 * LNR_MAX_LEVELS)
__device__ __forceinline__ void stage_level_tables(const LnrNetSpec& spec, float* lds) {
    uint32_t* u = reinterpret_cast<uint32_t*>(lds);
    for (int i = threadIdx.x; i < LNR_MAX_LEVELS; i += blockDim.x) {
        lds[i] = spec.level_scale[i];
        u[LNR_MAX_LEVELS + i] = spec.level_res[i];
        u[2 * LNR_MAX_LEVELS + i] = spec.level_size[i];
        u[3 * LNR_MAX_LEVELS + i] = spec.level_offset[i];
        const uint32_t sz = spec.level_size[i];
        u[4 * LNR_MAX_LEVELS + i] = (spec.level_hashed[i] & 1u) | ((sz != 0u && (sz & (sz - 1u)) == 0u) ? 2u : 0u);   // bit0 hashed, bit1 size is 2^k
    }
}

__device__ __forceinline__ LevelCell level_cell(const float* lvt, int lv, const float x[3]) {
    LevelCell c;
    const uint32_t* u = reinterpret_cast<const uint32_t*>(lvt);
    c.scale = lvt[lv];
    c.res = u[LNR_MAX_LEVELS + lv];
    c.size = u[2 * LNR_MAX_LEVELS + lv];
    c.offset = u[3 * LNR_MAX_LEVELS + lv];
    c.hashed = u[4 * LNR_MAX_LEVELS + lv];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float pos = lnr_add_rn(lnr_mul_rn(x[d], c.scale), 0.5f);
        float fl = floorf(pos);
        c.frac[d] = pos - fl;
        c.base[d] = (uint32_t)(int32_t)fl;
    }
    return c;
}

// kept out of line so that the common power-of-two case really skips the division sequence
__device__ __noinline__ uint32_t lnr_slow_mod(uint32_t a, uint32_t b) { return a % b; }

__device__ __forceinline__ uint32_t cell_entry(const LevelCell& c, int corner) {
    uint32_t cx = c.base[0] + (corner & 1), cy = c.base[1] + ((corner >> 1) & 1), cz = c.base[2] + ((corner >> 2) & 1);
    uint32_t idx = (c.hashed & 1u) ? (cx ^ (cy * PRIME_Y) ^ (cz * PRIME_Z)) : (cx + cy * c.res + cz * c.res * c.res);
    // table sizes are powers of two for every level of the usual configurations: mask instead of a ~40-instruction modulo
    if (c.hashed & 2u) idx &= (c.size - 1u); else idx = lnr_slow_mod(idx, c.size);
    return c.offset + idx;
}

__device__ __forceinline__ float corner_weight(const LevelCell& c, int corner) {
    float wx = (corner & 1) ? c.frac[0] : 1.0f - c.frac[0];
    float wy = (corner & 2) ? c.frac[1] : 1.0f - c.frac[1];
    float wz = (corner & 4) ? c.frac[2] : 1.0f - c.frac[2];
    return wx * wy * wz;
}

// Frequency encoding: feature k = sin(x[dim]*2^freq*pi + (k&1)*pi/2), order [dim][freq][sin,cos]
__device__ __forceinline__ float freq_phase(const LnrNetSpec& spec, const float x[3], int k, float* dphase_dx, int* dim_out) {
    const int per_dim = 2 * spec.n_frequencies;
    const int dim = k / per_dim;
    const int rem = k - dim * per_dim;
    const float mult = (float)(1u << (rem >> 1));     // exact power of two (exp2f is not exact on the GPU)
    float xv = dim == 0 ? x[0] : (dim == 1 ? x[1] : x[2]);
    float ph = lnr_mul_rn(lnr_mul_rn(xv, mult), LNR_PI_F);
    if (rem & 1) ph = lnr_add_rn(ph, LNR_PI_2_F);
    *dphase_dx = mult * LNR_PI_F;
    *dim_out = dim;
    return ph;
}

__device__ __forceinline__ void freq_features4(const LnrNetSpec& spec, const float x[3], int k0, float out[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + r;
        if (k < spec.enc_dim) {
            float d; int dim;
            out[r] = sinf(freq_phase(spec, x, k, &d, &dim));
        } else {
            out[r] = 1.0f;
        }
    }
}

__device__ __forceinline__ void freq_features4_bwd(const LnrNetSpec& spec, const float x[3], int k0,
                                                   const float d_out[4], float dx[3]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + r;
        if (k < spec.enc_dim && d_out[r] != 0.0f) {
            float d; int dim;
            float ph = freq_phase(spec, x, k, &d, &dim);
            float v = d_out[r] * cosf(ph) * d;
            if (dim == 0) dx[0] += v; else if (dim == 1) dx[1] += v; else dx[2] += v;
        }
    }
}
