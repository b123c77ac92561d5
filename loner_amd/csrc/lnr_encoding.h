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

// 32-bit byte offsets from wave-uniform base pointers (global_load saddr + voffset): per-lane 64-bit pointer
// arithmetic costs 3-4 VALU instructions per access, and these kernels are bound by VALU issue, not by memory.
// The host checks n_points < 2^28 and table bytes < 2^32.
template <typename T>
__device__ __forceinline__ T ld32(const void* base, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)byte_off);
}
template <typename T>
__device__ __forceinline__ void st32(void* base, uint32_t byte_off, T v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (size_t)byte_off) = v;
}

// Streaming variants (nt cache policy): data that is written once and read once by a later kernel (gradient records) or read
// once per launch (the d_feature planes) should not evict the table lines the gathers live on from the 4 MB L2 of an XCD.
template <typename T>
__device__ __forceinline__ T ld32_stream(const void* base, uint32_t byte_off) {
    return __builtin_nontemporal_load(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)byte_off));
}
template <typename T>
__device__ __forceinline__ void st32_stream(void* base, uint32_t byte_off, T v) {
    __builtin_nontemporal_store(v, reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (size_t)byte_off));
}
__device__ __forceinline__ void store_stream_b96(uint64_t global_addr, uint32_t a, uint32_t b, uint32_t c) {
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    const u32x3 v = {a, b, c};
    asm volatile("global_store_dwordx3 %0, %1, off nt" :: "v"(global_addr), "v"(v) : "memory");
}
__device__ __forceinline__ void store_stream_b64(uint64_t global_addr, uint2 v) {
    asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(global_addr), "v"(v) : "memory");
}

// Sample index of a thread that walks m, m+step, m+2*step, ...: the ray index m / n_samples is kept incrementally
// (one division at the start instead of one per sample).
struct SampleCursor {
    uint32_t m, ray, rem;          // m = ray * n_samples + rem
    uint32_t step, dq, dr, S;      // step = dq * S + dr
    __device__ __forceinline__ void init(uint32_t m0, uint32_t step_, uint32_t n_samples) {
        S = n_samples; step = step_; m = m0;
        ray = m0 / S; rem = m0 - ray * S;
        dq = step_ / S; dr = step_ - dq * S;
    }
    __device__ __forceinline__ void advance() {
        m += step; ray += dq; rem += dr;
        if (rem >= S) { rem -= S; ++ray; }
    }
};

// Point as loaded (no arithmetic yet, so the loads can be issued a whole iteration ahead of their use).
// pts mode: o = xyz; rays mode: o = origin, d = direction, z = depth.  Scalar fields, all always assigned: an array
// written differently by the two modes ends up in scratch memory.
struct RawPoint { float o0, o1, o2, d0, d1, d2, z; };

// Ray records through the SCALAR cache when every lane of the wave sits on the same ray (rays form, samples per ray a multiple of the
// wave's sample count: ray_uniform below).  A vector load costs the CU's address path ~16 clocks per wave instruction even when all 64
// lanes read one dword (tools/gather_bench.hip: 3.3 lanes per clock and CU at best) - six of them per step were a tenth of the fine
// levels' gather time and most of the coarse levels'; the scalar unit fetches the same six floats with one or two s_load instructions
// that never touch that path.  The rays are read-only for the whole launch (constant address space: always selected as SMEM).
typedef const float __attribute__((address_space(4))) lnr_cfloat;
__device__ __forceinline__ bool ray_uniform(const PointSrc& s, uint32_t samples_per_wave) {
    return s.pts == nullptr && ((uint32_t)s.n_samples % samples_per_wave) == 0u;
}

// uni: ray_uniform() holds for the calling wave (wave-uniform flag); `ray` of the first active lane then stands for all of them
__device__ __forceinline__ void load_raw_point(const PointSrc& s, uint32_t m, uint32_t ray, RawPoint& r, bool uni = false) {
    if (s.pts) {
        r.o0 = ld32<float>(s.pts, m * 12u); r.o1 = ld32<float>(s.pts, m * 12u + 4u); r.o2 = ld32<float>(s.pts, m * 12u + 8u);
        r.d0 = 0.0f; r.d1 = 0.0f; r.d2 = 0.0f; r.z = 0.0f;
    } else if (uni) {
        const uint32_t ru = (uint32_t)__builtin_amdgcn_readfirstlane((int)ray);
        const lnr_cfloat* rp = (const lnr_cfloat*)(uintptr_t)(s.rays + (size_t)ru * LNR_RAY_STRIDE);
        r.o0 = rp[0]; r.o1 = rp[1]; r.o2 = rp[2]; r.d0 = rp[3]; r.d1 = rp[4]; r.d2 = rp[5];
        r.z = ld32<float>(s.z, m * 4u);
    } else {
        const uint32_t ro = ray * (uint32_t)(LNR_RAY_STRIDE * 4);
        r.o0 = ld32<float>(s.rays, ro); r.o1 = ld32<float>(s.rays, ro + 4u); r.o2 = ld32<float>(s.rays, ro + 8u);
        r.d0 = ld32<float>(s.rays, ro + 12u); r.d1 = ld32<float>(s.rays, ro + 16u); r.d2 = ld32<float>(s.rays, ro + 20u);
        r.z = ld32<float>(s.z, m * 4u);
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

// The same with the kind of source a compile-time constant (encode_forward_kernel's specialised loops: no uniform branch per step).
enum : int { LNR_SRC_PTS = 0, LNR_SRC_RAY_UNIFORM = 1, LNR_SRC_RAY = 2 };
__device__ __forceinline__ int point_source_kind(const PointSrc& s, uint32_t samples_per_wave) {
    return s.pts ? LNR_SRC_PTS : (ray_uniform(s, samples_per_wave) ? LNR_SRC_RAY_UNIFORM : LNR_SRC_RAY);
}
// the loads of a point (issued one step ahead of their use by the forward loops) ...
template <int SK>
__device__ __forceinline__ void load_raw_point_of(const PointSrc& s, uint32_t m, uint32_t ray, RawPoint& r) {
    if constexpr (SK == LNR_SRC_PTS) {
        r.o0 = ld32<float>(s.pts, m * 12u); r.o1 = ld32<float>(s.pts, m * 12u + 4u); r.o2 = ld32<float>(s.pts, m * 12u + 8u);
        r.d0 = 0.0f; r.d1 = 0.0f; r.d2 = 0.0f; r.z = 0.0f;
    } else if constexpr (SK == LNR_SRC_RAY_UNIFORM) {
        const uint32_t ru = (uint32_t)__builtin_amdgcn_readfirstlane((int)ray);
        const lnr_cfloat* rp = (const lnr_cfloat*)(uintptr_t)(s.rays + (size_t)ru * LNR_RAY_STRIDE);
        r.o0 = rp[0]; r.o1 = rp[1]; r.o2 = rp[2]; r.d0 = rp[3]; r.d1 = rp[4]; r.d2 = rp[5];
        r.z = ld32<float>(s.z, m * 4u);
    } else {
        const uint32_t ro = ray * (uint32_t)(LNR_RAY_STRIDE * 4);
        r.o0 = ld32<float>(s.rays, ro); r.o1 = ld32<float>(s.rays, ro + 4u); r.o2 = ld32<float>(s.rays, ro + 8u);
        r.d0 = ld32<float>(s.rays, ro + 12u); r.d1 = ld32<float>(s.rays, ro + 16u); r.d2 = ld32<float>(s.rays, ro + 20u);
        r.z = ld32<float>(s.z, m * 4u);
    }
}
// ... and its unit-cube coordinates
template <int SK>
__device__ __forceinline__ void unit_point_of(const RawPoint& r, float x[3]) {
    float p[3] = {r.o0, r.o1, r.o2};
    if constexpr (SK != LNR_SRC_PTS) {                   // o + d*z, as the reference rounds it
        p[0] = lnr_add_rn(r.o0, lnr_mul_rn(r.d0, r.z));
        p[1] = lnr_add_rn(r.o1, lnr_mul_rn(r.d1, r.z));
        p[2] = lnr_add_rn(r.o2, lnr_mul_rn(r.d2, r.z));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = lnr_mul_rn(lnr_add_rn(p[k], 1.0f), 0.5f);
}

__device__ __forceinline__ void load_unit_point(const PointSrc& s, int64_t m, float x[3]) {
    RawPoint r;
    load_raw_point(s, (uint32_t)m, s.pts ? 0u : (uint32_t)(m / s.n_samples), r);
    unit_point(s, r, x);
}

// ------------------------------------------------------------------------------------------------
// multiresolution hash grid: one level (wave-uniform, its geometry lives in SGPRs), one point per lane
// ------------------------------------------------------------------------------------------------
struct LevelInfo {
    float scale;
    uint32_t res, size, offset;     // entries
    bool hashed, pow2;
    bool pos_fma;                   // grid position rounded once (tiny-cuda-nn's fmaf) or as separate multiply and add
};

__device__ __forceinline__ LevelInfo level_info(const LnrNetSpec& spec, int lv) {
    LevelInfo L;
    L.scale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(spec.level_scale[lv])));
    L.res = __builtin_amdgcn_readfirstlane(spec.level_res[lv]);
    L.size = __builtin_amdgcn_readfirstlane(spec.level_size[lv]);
    L.offset = __builtin_amdgcn_readfirstlane(spec.level_offset[lv]);
    L.hashed = (__builtin_amdgcn_readfirstlane(spec.level_hashed[lv]) & 1u) != 0u;
    L.pow2 = (L.size & (L.size - 1u)) == 0u;
    L.pos_fma = __builtin_amdgcn_readfirstlane(spec.pos_rounding) == LNR_POS_FMA;
    return L;
}

struct Cell {
    uint32_t b[3];      // integer cell
    float frac[3];
};

__device__ __forceinline__ Cell cell_of(const LevelInfo& L, const float x[3]) {
    Cell c;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float pos = L.pos_fma ? __builtin_fmaf(x[d], L.scale, 0.5f) : lnr_add_rn(lnr_mul_rn(x[d], L.scale), 0.5f);
        const float fl = floorf(pos);
        c.frac[d] = pos - fl;
        c.b[d] = (uint32_t)(int32_t)fl;
    }
    return c;
}

// Table entries (absolute, in entries) of the 8 corners; corner bit 0 = +x, bit 1 = +y, bit 2 = +z.  The per-axis
// terms are shared between corners: 12 xors (or 8 adds) instead of 8 full hash evaluations.
__device__ __forceinline__ void cell_entries(const LevelInfo& L, const Cell& c, uint32_t e[8]) {
    if (L.hashed) {
        const uint32_t hx0 = c.b[0], hx1 = c.b[0] + 1u;
        const uint32_t hy0 = c.b[1] * PRIME_Y, hy1 = hy0 + PRIME_Y;
        const uint32_t hz0 = c.b[2] * PRIME_Z, hz1 = hz0 + PRIME_Z;
        const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = ((k & 1) ? hx1 : hx0) ^ yz[k >> 1];
    } else {
        const uint32_t sy = L.res, sz = L.res * L.res;
        const uint32_t i0 = c.b[0] + c.b[1] * sy + c.b[2] * sz;
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = i0 + (uint32_t)(k & 1) + ((k & 2) ? sy : 0u) + ((k & 4) ? sz : 0u);
    }
    if (L.pow2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = (e[k] & (L.size - 1u)) + L.offset;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = e[k] % L.size + L.offset;
    }
}

// trilinear weights, associated as (wx*wy)*wz
__device__ __forceinline__ void cell_weights(const Cell& c, float w[8]) {
    const float wx[2] = {1.0f - c.frac[0], c.frac[0]}, wy[2] = {1.0f - c.frac[1], c.frac[1]}, wz[2] = {1.0f - c.frac[2], c.frac[2]};
    const float wxy[4] = {wx[0] * wy[0], wx[1] * wy[0], wx[0] * wy[1], wx[1] * wy[1]};
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = wxy[k & 3] * wz[k >> 2];
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

// sin and cos of one phase from ONE range reduction (the comment at freq_forward_h16_kernel, lnr_encode.hip, derives the pair form):
//   k = rint(ph 2/pi), r = ph - k pi/2 (three-term Cody-Waite with fma), minimax sin / cos of r on [-pi/4, pi/4], quadrant from k & 3
__device__ __forceinline__ void sincos_f32(float ph, float* s_out, float* c_out) {
    const float k = __builtin_rintf(ph * 0.636619772367581343f);
    float r = __builtin_fmaf(-k, 1.57079637050628662109375f, ph);
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

