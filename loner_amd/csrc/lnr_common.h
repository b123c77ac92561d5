// Shared device/host helpers for libloner_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/loner_hip.h"

#define LNR_WAVE 64

void lnr_set_error(const char* fmt, ...);

#define LNR_REQUIRE(cond, ...)                        \
    do {                                              \
        if (!(cond)) {                                \
            lnr_set_error(__VA_ARGS__);               \
            return LNR_ERR_INVALID_ARG;               \
        }                                             \
    } while (0)

#define LNR_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            lnr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return LNR_ERR_LAUNCH;                                                    \
        }                                                                             \
    } while (0)

// per-kernel timing hooks (lnr_core.hip); a scope records one HIP event before and one after the launches it encloses
int lnr_profile_begin(const char* name, hipStream_t st);
void lnr_profile_end(int span, hipStream_t st);
struct LnrProfScope {
    int span; hipStream_t st;
    LnrProfScope(const char* name, hipStream_t s) : span(lnr_profile_begin(name, s)), st(s) {}
    ~LnrProfScope() { lnr_profile_end(span, st); }
};

// lnr_rng_draws(LNR_DRAW_NOISE): the N(0,1) values lnr_render_* / lnr_los_loss_fused draw, from their own translation unit (lnr_render.hip)
int lnr_render_noise_draws(uint64_t seed, int n_rays, int n_per_ray, float* out, hipStream_t st);

static inline int lnr_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Separately rounded float multiply / add.  __fmul_rn/__fadd_rn inline to plain fmul/fadd, which the backend may
// still contract into an fma under -ffp-contract=fast; the empty asm pins the intermediate in a VGPR so that the
// product is rounded on its own, as the reference's torch ops do.
__device__ __forceinline__ float lnr_mul_rn(float a, float b) { float r = a * b; asm volatile("" : "+v"(r)); return r; }
__device__ __forceinline__ float lnr_add_rn(float a, float b) { float r = a + b; asm volatile("" : "+v"(r)); return r; }

// Shifts within a 16-lane row as DPP modifiers (no LDS crossbar traffic, unlike __shfl_* with width 16,
// which lowers to ds_bpermute).  row_up<N>: lane i reads lane i-N of its row; row_down<N>: lane i reads lane i+N.
// Lanes whose source falls outside the row read 0 (bound_ctrl) - callers mask those lanes anyway.
template <int N> __device__ __forceinline__ int row_up_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xF, 0xF, true); }
template <int N> __device__ __forceinline__ int row_down_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x100 + N, 0xF, 0xF, true); }
template <int N> __device__ __forceinline__ float row_down_f(float v) { return __int_as_float(row_down_i<N>(__float_as_int(v))); }

// ---------------------------------------------------------------------------------------------
// wave-level primitives (64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// Sum over the 64 lanes with DPP only (quad swaps, half-row and row mirrors: plain VALU operands) plus four readlanes
// for the rows; __shfl_xor lowers to ds_bpermute (an LDS round trip per step).  The result is wave-uniform.  Must be
// called with all lanes active.  Association differs from wave_sum (mirror tree instead of xor butterfly).
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    const int vi = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(vi, 0)) + __int_as_float(__builtin_amdgcn_readlane(vi, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(vi, 32)) + __int_as_float(__builtin_amdgcn_readlane(vi, 48)));
}

// exclusive prefix product / inclusive helpers over 64 lanes
__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(inc, o, 64);
        if (lane >= o) inc *= t;
    }
    float ex = __shfl_up(inc, 1, 64);
    return lane == 0 ? 1.0f : ex;
}
__device__ __forceinline__ float wave_excl_sum(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    float ex = __shfl_up(inc, 1, 64);
    return lane == 0 ? 0.0f : ex;
}

// ---------------------------------------------------------------------------------------------
// counter-based RNG (Philox4x32-10), used when the caller passes no random tensors
// ---------------------------------------------------------------------------------------------
struct Philox4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ uint32_t lnr_mulhi32(uint32_t a, uint32_t b) {
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

__host__ __device__ inline Philox4 philox4x32_10(uint64_t counter_lo, uint64_t counter_hi, uint64_t key) {
    uint32_t c0 = (uint32_t)counter_lo, c1 = (uint32_t)(counter_lo >> 32);
    uint32_t c2 = (uint32_t)counter_hi, c3 = (uint32_t)(counter_hi >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        uint32_t hi0 = lnr_mulhi32(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = lnr_mulhi32(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}
// uniform in [0,1) with 24 random bits (same mapping torch uses for float32)
__host__ __device__ __forceinline__ float lnr_u01(uint32_t bits) {
    return (float)(bits & 0x00FFFFFFu) * (1.0f / 16777216.0f);
}
// stream ids keep the three per-iteration draws independent
#define LNR_STREAM_JITTER 0x11ull
#define LNR_STREAM_PDF 0x22ull
#define LNR_STREAM_NOISE 0x33ull

__device__ __forceinline__ float lnr_rand_uniform(uint64_t seed, uint64_t stream_id, uint64_t ray, uint32_t idx) {
    Philox4 p = philox4x32_10(((uint64_t)idx >> 2) | (ray << 20), stream_id, seed);
    uint32_t v = (idx & 3) == 0 ? p.x : (idx & 3) == 1 ? p.y : (idx & 3) == 2 ? p.z : p.w;
    return lnr_u01(v);
}
__device__ __forceinline__ float lnr_rand_normal(uint64_t seed, uint64_t ray, uint32_t idx) {
    Philox4 p = philox4x32_10(((uint64_t)idx >> 1) | (ray << 20), LNR_STREAM_NOISE, seed);
    uint32_t a = (idx & 1) ? p.z : p.x, b = (idx & 1) ? p.w : p.y;
    float u1 = 1.0f - lnr_u01(a);   // (0,1]
    float u2 = lnr_u01(b);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

__device__ __forceinline__ int lnr_live_rays(int n_rays, const int32_t* n_rays_dev) {
    if (n_rays_dev == nullptr) return n_rays;
    int v = *n_rays_dev;
    return v < n_rays ? v : n_rays;
}

// -DLNR_PHASE_TIMING (development builds only, LNR_EXTRA_HIPCC_FLAGS): shader-clock cycles per phase of a kernel, summed over its
// waves into a __device__ array; the launcher prints and clears them when LNR_PHASE_TIMING is set in the environment.
#ifdef LNR_PHASE_TIMING
#include <cstdio>
#define LNR_N_PHASES 16
#define PHASE_INIT() unsigned long long ph_acc[LNR_N_PHASES] = {}; unsigned long long ph_last = __builtin_amdgcn_s_memtime()
#define PHASE(k) do { const unsigned long long ph_now = __builtin_amdgcn_s_memtime(); ph_acc[k] += ph_now - ph_last; ph_last = ph_now; } while (0)
#define PHASE_FLUSH(sym, base) do { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < LNR_N_PHASES; ++k_) if (ph_acc[k_]) atomicAdd(&sym[(base) + k_], ph_acc[k_]); } while (0)
static inline bool lnr_phase_fetch(const void* symbol, unsigned long long* h, int n, hipStream_t st) {
    if (hipStreamSynchronize(st) != hipSuccess || hipMemcpyFromSymbol(h, symbol, n * sizeof(unsigned long long)) != hipSuccess) return false;
    unsigned long long zero[4 * LNR_N_PHASES] = {};
    return hipMemcpyToSymbol(symbol, zero, n * sizeof(unsigned long long)) == hipSuccess;
}
static inline void lnr_phase_print(const char* kernel, const char* const* names, const unsigned long long* h) {
    unsigned long long sum = 0;
    for (int k = 0; k < LNR_N_PHASES; ++k) sum += h[k];
    for (int k = 0; k < LNR_N_PHASES; ++k)
        if (h[k]) fprintf(stderr, "[lnr phases] %-24s %-28s %6.2f %%  (%llu)\n", kernel, names[k], 100.0 * (double)h[k] / (double)sum, h[k]);
}
#else
#define PHASE_INIT()
#define PHASE(k)
#define PHASE_FLUSH(sym, base)
#endif
