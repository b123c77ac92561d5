// Level-major input encoding (forward and backward) for the density network (gfx950).
//
// Why level-major: the default hash grid is 29.7 MB, the L2 of an XCD 4 MB.  A kernel that walks all 16 levels per
// sample thrashes L2 (PMC: 3.5 GB of memory-side fetches per forward launch for 0.13 GB of algorithmic bytes).
// Here the grid is ordered level by level (blockIdx / blocks_per_group = level), so at any moment the chip works on
// one or two levels whose tables (<= 2 MB each) stay resident in every XCD's L2, and each gather is an L2 hit.
// Features travel to the MLP kernels as [feature plane][sample] arrays in HBM (coalesced both ways).
//
// With the gathers served by L2 these kernels are bound by VALU issue, so the per-sample code is kept lean: level
// geometry in SGPRs (the level is wave-uniform), corner hashes/weights from shared per-axis terms, 32-bit offsets
// from uniform base pointers, the ray index advanced incrementally.
//
// Backward: one thread per (sample, level) re-derives the cell, turns d_feature into table-gradient records
// (run-length combined over consecutive samples of a ray on coarse levels), and writes its contribution to d/dx as
// one plane per level; records are reduced by table_grad_reduce2_kernel (lnr_density.hip).
#include "lnr_encoding.h"
#include <cstdio>
#include <cstdlib>

#define ENC_BLOCK 256

// Which level group ("slot") and which chunk of it a workgroup works on.
// Plain order: slot = blockIdx / bpg - the chip walks the levels one after the other, and EVERY XCD pulls every level's table (<= 2 MB)
// into its own L2: 8 x 30 MB of fabric reads per launch whatever the batch size (the per-launch floor of a one-keyframe shard:
// profiles/r03_trace_one_keyframe_iteration.txt, 86 us for 262 k samples against 0.46 ms for 2.1 M).
// XCD-affine order (n_slots a multiple of 8): the hardware hands workgroup b to XCD b % 8 (MI355X_MICROARCH.md; used for speed only,
// nothing depends on it), so slot = (b % 8) + 8 * ((b / 8) / bpg) gives every XCD its own levels - x, x + 8, ... one after the other:
// a table is read into ONE L2, and a coarse and a fine level per XCD keep the eight XCDs' work even (default network: 16 levels).
__device__ __forceinline__ bool level_slot(int bpg, int n_slots, bool xcd_affine, int& slot, int& chunk) {
    if (xcd_affine) {
        const int r = (int)(blockIdx.x >> 3);
        slot = (int)(blockIdx.x & 7u) + 8 * (r / bpg);
        chunk = r % bpg;
    } else {
        slot = (int)blockIdx.x / bpg;
        chunk = (int)blockIdx.x % bpg;
    }
    return slot < n_slots;
}
// Measured (profiles/r04_call3_xcd_affine_ab.log): the forward of a one-keyframe shard (262 k samples) 113 -> 77 us, of the bench window
// (2.1 M samples) 0.43 -> 0.50 ms - with eight XCDs each walking its own two levels the chip is never on ONE level's working set, and a
// long launch is dispatched less regularly than b % 8; the backward's partition kernels lose at both sizes.  Hence: the forward only,
// and only for launches small enough that the table fills dominate.
#ifndef LNR_XCD_AFFINE_MAX_POINTS
#define LNR_XCD_AFFINE_MAX_POINTS (1 << 19)
#endif
static inline bool lnr_xcd_affine(int n_slots, int64_t n_points, bool forward) {
    bool on = forward && n_points <= LNR_XCD_AFFINE_MAX_POINTS;
#ifdef LNR_ABLATE
    static const char* env = getenv("LNR_X_XCD");                 // development builds: 0 / 1 force it off / on for every encode kernel
    if (env) on = atoi(env) != 0;
#endif
    return on && n_slots > 0 && n_slots % 8 == 0;
}

// the n-th (0-based) set bit of mask: the level a workgroup group works on (level_mask = all levels outside ablation builds)
__device__ __forceinline__ int nth_level(uint32_t mask, int n) {
    for (int i = 0; i < n; ++i) mask &= mask - 1u;
    return __builtin_ctz(mask);
}


// ------------------------------------------------------------------------------------------------ forward
template <int F>
__device__ __forceinline__ void gather_entries(const float* __restrict__ table, const uint32_t e[8], float tv[8][F]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t off = e[k] * (uint32_t)(F * 4);
        if constexpr (F == 1) tv[k][0] = ld32<float>(table, off);
        else if constexpr (F == 2) { const float2 t2 = ld32<float2>(table, off); tv[k][0] = t2.x; tv[k][1] = t2.y; }
        else {
#pragma unroll
            for (int q = 0; q < F / 4; ++q) {
                const float4 t4 = ld32<float4>(table, off + 16u * q);
                tv[k][4 * q] = t4.x; tv[k][4 * q + 1] = t4.y; tv[k][4 * q + 2] = t4.z; tv[k][4 * q + 3] = t4.w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ paired lanes
// (helpers of encode_dx_pair_kernel, the -DLNR_SPLIT_DX=1 experiment; the forward ran in this form for part of round 4 - see
// forward_level_loop for what replaced it)
// Two lanes per sample: lane 2i takes the four corners of the cell with x = b0, lane 2i+1 those with x = b0 + 1.
// Why: a random table gather costs one L2 LINE per distinct line an instruction touches (tools/gather_bench.hip: 0.5 lines/clk/CU
// whatever the access width), and the two x-neighbours of a corner pair sit in the same 64-byte line most of the time - dense levels:
// e1 = e0 + 1; hashed levels: e1 = e0 ^ (2^(t+1) - 1), t = trailing one bits of x, i.e. inside one aligned group of 8 entries for
// 7 of 8 cells.  With one lane per sample the two neighbours are fetched by DIFFERENT instructions (corner k and k + 1) and the line
// is paid twice; with the pair in adjacent lanes of ONE instruction it is paid once: ~1.1 instead of 2 lines per corner pair on the
// fine levels, where no two samples share a cell.  The price is the cell arithmetic done twice - VALU these kernels have to spare
// (the gathers of 64 samples occupy a CU's L1 path for ~1000 clocks, their arithmetic the four SIMDs for ~40).
__device__ __forceinline__ float pair_partner_f(float v) {       // the other lane of the pair (quad_perm [1,0,3,2]); all lanes active
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}

// entries of the 4 (y, z) rows of a cell at x = b0 + hx: row r = (y bit) + 2 (z bit) is corner k = hx + 2 r of cell_entries
__device__ __forceinline__ void cell_entries_x(const LevelInfo& L, const Cell& c, uint32_t hx, uint32_t e[4]) {
    if (L.hashed) {
        const uint32_t x = c.b[0] + hx;
        const uint32_t hy0 = c.b[1] * PRIME_Y, hy1 = hy0 + PRIME_Y;
        const uint32_t hz0 = c.b[2] * PRIME_Z, hz1 = hz0 + PRIME_Z;
        e[0] = x ^ (hy0 ^ hz0); e[1] = x ^ (hy1 ^ hz0); e[2] = x ^ (hy0 ^ hz1); e[3] = x ^ (hy1 ^ hz1);
    } else {
        const uint32_t sy = L.res, sz = L.res * L.res;
        const uint32_t i0 = c.b[0] + hx + c.b[1] * sy + c.b[2] * sz;
        e[0] = i0; e[1] = i0 + sy; e[2] = i0 + sz; e[3] = i0 + sy + sz;
    }
    if (L.pow2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = (e[r] & (L.size - 1u)) + L.offset;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = e[r] % L.size + L.offset;
    }
}

template <int F>
__device__ __forceinline__ void gather_entries4(const float* __restrict__ table, const uint32_t e[4], float tv[4][F]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t off = e[r] * (uint32_t)(F * 4);
        if constexpr (F == 1) tv[r][0] = ld32<float>(table, off);
        else if constexpr (F == 2) { const float2 t2 = ld32<float2>(table, off); tv[r][0] = t2.x; tv[r][1] = t2.y; }
        else {
#pragma unroll
            for (int q = 0; q < F / 4; ++q) {
                const float4 t4 = ld32<float4>(table, off + 16u * q);
                tv[r][4 * q] = t4.x; tv[r][4 * q + 1] = t4.y; tv[r][4 * q + 2] = t4.z; tv[r][4 * q + 3] = t4.w;
            }
        }
    }
}

// ---- the forward kernel: one workgroup row per level, the per-step loop specialised at compile time
// Level kinds: what the entry arithmetic of a level needs (wave-uniform; decided once per workgroup, not once per step - the round-4
// first form carried ~20 uniform branches and the general modulo through every step: SQ_INSTS_SALU 43 M against 127 M VALU per launch).
enum : int { LK_HASH_POW2 = 0,      // hashed, table size a power of two (every hashed level tiny-cuda-nn builds): xor + mask
             LK_DENSE = 1,          // not hashed: index arithmetic; the modulo only if an index reaches the table size (points outside
                                    // the unit cube - the integer result is the same either way)
             LK_GENERIC = 2 };      // anything else (hashed with another size): flags read at run time

template <int LK, int N>
__device__ __forceinline__ void wrap_entries(const LevelInfo& L, uint32_t e[N]) {
    if constexpr (LK == LK_HASH_POW2) {
#pragma unroll
        for (int k = 0; k < N; ++k) e[k] = (e[k] & (L.size - 1u)) + L.offset;
    } else if constexpr (LK == LK_DENSE) {
        uint32_t any = e[0];
#pragma unroll
        for (int k = 1; k < N; ++k) any |= e[k];           // >= every entry: below the size = no entry needs the modulo
        if (any >= L.size) {
#pragma unroll
            for (int k = 0; k < N; ++k) e[k] %= L.size;
        }
#pragma unroll
        for (int k = 0; k < N; ++k) e[k] += L.offset;
    } else {
        if (L.pow2) {
#pragma unroll
            for (int k = 0; k < N; ++k) e[k] = (e[k] & (L.size - 1u)) + L.offset;
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) e[k] = e[k] % L.size + L.offset;
        }
    }
}

// entries of a cell before the wrap: all 8 corners (corner bit 0 = +x, 1 = +y, 2 = +z)
__device__ __forceinline__ void raw_entries8(const LevelInfo& L, const Cell& c, uint32_t e[8]) {
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
}
// One level's samples, chunk `chunk` of `bpg`: one lane per (sample, level), 8 gathers.
// H16: the features leave as half2 pairs (plane p holds features 2p, 2p+1: one dword per sample; F/2 planes per level) for the
// fp16 MLP kernels (lnr_density_f16.hip), rounded to nearest even; the interpolation itself stays fp32.
//
// Software pipeline.  A step as one dependent chain - depth load (HBM) -> cell -> gathers (L2) -> interpolation -> stores - and, gfx9
// having ONE in-order counter for vector-memory loads and stores, a wait for the next depth that also waits for the stores just
// issued, took ~8000 clocks with 8 waves per SIMD to cover it, however few lines the gathers touched (profiles/r04_forward_levels.txt:
// 0.107 ms per coarse level and 16.8 M samples with one lane per sample, with two, with the gathers replaced by scalar loads).
// Hence, per step i: the point loads of step i + 1 and the STORES of step i - 1 go out together with the gathers of i, and only then
// the wait for the gathers and the interpolation.  The body is straight-line code (no branch on "is this sample live": the waits the
// compiler places where divergent branches join are full ones): steps past the last sample - the ragged tail of the last tile - load
// the last sample's point again and store zeros; the first step's "previous" store writes zeros to the slot the next step fills.
//
// Measured and dropped once the loop was pipelined (profiles/r04_forward_levels.txt):
//   * two lanes per sample (lane 2i the four corners with x = b0, lane 2i+1 those with x = b0 + 1: x-neighbours share a 64-byte line,
//     one instruction pays it once) - the round-4 first form, 5 % faster than the un-pipelined one-lane loop; against the pipelined
//     one 0.409 : 0.379 ms on the training window, 3.01 : 2.96 ms on an inference launch, slower on every level of both;
//   * a wave whose 64 samples sit in <= 4 cells (the coarse levels of the inference path) fetching each cell once through the scalar
//     cache (v_readlane + 8 s_load per distinct cell instead of 8 vector gathers): no faster on any level, 73 registers against 46.
template <int F, bool H16, int LK, bool FMA, int SK>
__device__ __forceinline__ void forward_level_loop(LevelInfo L, const float* __restrict__ table, const PointSrc& src, float* __restrict__ planes,
                                                   uint32_t plane_bytes, uint32_t M, uint32_t Mt, int chunk, int bpg) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    L.pos_fma = FMA;
    if constexpr (LK == LK_HASH_POW2) { L.hashed = true; L.pow2 = true; }
    if constexpr (LK == LK_DENSE) L.hashed = false;
    constexpr int NV = H16 ? F / 2 : F;                                // dwords a lane stores per step, one per plane
    SampleCursor cur;
    const uint32_t per_ray = SK == LNR_SRC_PTS ? 1u : (uint32_t)src.n_samples;
    cur.init((uint32_t)chunk * ENC_BLOCK + threadIdx.x, (uint32_t)bpg * ENC_BLOCK, per_ray);
    if (cur.m >= Mt) return;
    const uint32_t m_last = M - 1u, ray_last = m_last / per_ray;      // (M >= 1 here)
    RawPoint ahead;
    load_raw_point_of<SK>(src, cur.m < M ? cur.m : m_last, cur.m < M ? cur.ray : ray_last, ahead);
    uint32_t held[NV], m_held = cur.m;                 // the step before's results, stored while this step's gathers are on their way
#pragma unroll
    for (int v = 0; v < NV; ++v) held[v] = 0u;
    while (cur.m < Mt) {
        const uint32_t m = cur.m;
        const bool live = m < M;
        const RawPoint rp = ahead;
        cur.advance();
        const bool more = cur.m < M;
        uint32_t m_next = more ? cur.m : m_last, ray_next = more ? cur.ray : ray_last;
        // (opaque to the optimiser: seen as "the next step's cur.m", the loads below are merged with the loop's first ones and moved
        // to the top of the next step wherever the body has inner control flow - un-pipelining the loop)
        asm volatile("" : "+v"(m_next), "+v"(ray_next));
        float x[3];
        unit_point_of<SK>(rp, x);
        const Cell c = cell_of(L, x);
        uint32_t e[8]; float tv[8][F], w[8];
        raw_entries8(L, c, e);
        if constexpr (LK == LK_DENSE && F == 2) {
            // Dense levels, two features: the x-neighbour of corner 2r is the NEXT entry (e + 1: 16 contiguous bytes), so the eight
            // corners are four 16-byte gathers - what a coarse level waits for is the CU's address path, ~16 clocks per vector-memory
            // instruction however well it coalesces, 11 instructions per step before (8 gathers, the depth, 2 stores), 7 now.  Lanes
            // whose indices reach the table size (points outside the unit cube: the reference's modulo applies) re-fetch their
            // corners one by one.
            const uint32_t any = e[0] | e[1] | e[2] | e[3] | e[4] | e[5] | e[6] | e[7];
            const bool wrapped = any >= L.size;
            wrap_entries<LK, 8>(L, e);
            typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
            // (a wrapped index may be the level's LAST entry: the 16-byte gather starts one entry earlier at most, so that it never
            // reads past the level - past the parameter buffer, when the level is the last of an all-dense network - and a wrapped
            // lane re-fetches all eight corners)
            const uint32_t last_pair = L.offset + L.size - 2u;       // (wrap_entries made the indices absolute: + L.offset)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t ea = e[2 * r] < last_pair ? e[2 * r] : last_pair;
                const f4u q = *reinterpret_cast<const f4u*>(reinterpret_cast<const char*>(table) + (size_t)(ea * 8u));
                tv[2 * r][0] = q.x; tv[2 * r][1] = q.y; tv[2 * r + 1][0] = q.z; tv[2 * r + 1][1] = q.w;
            }
            if (wrapped) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float2 t2 = ld32<float2>(table, e[k] * 8u);
                    tv[k][0] = t2.x; tv[k][1] = t2.y;
                }
            }
        } else {
            wrap_entries<LK, 8>(L, e);
            gather_entries<F>(table, e, tv);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) st32<uint32_t>(planes, (uint32_t)v * plane_bytes + m_held * 4u, held[v]);
        load_raw_point_of<SK>(src, m_next, ray_next, ahead);
        __builtin_amdgcn_sched_barrier(0);              // (the scheduler otherwise sinks these loads below the interpolation)
        cell_weights(c, w);
        // corner after corner like the reference's loop (oracle/network.py: tiny-cuda-nn accumulates with fmaf)
        float out[F];
#pragma unroll
        for (int f = 0; f < F; ++f) out[f] = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int f = 0; f < F; ++f) out[f] = __builtin_fmaf(w[k], tv[k][f], out[f]);
        if (!live) {
#pragma unroll
            for (int f = 0; f < F; ++f) out[f] = 0.0f;
        }
        if constexpr (H16) {
#pragma unroll
            for (int q = 0; q < F / 2; ++q) held[q] = __builtin_bit_cast(uint32_t, h2{(_Float16)out[2 * q], (_Float16)out[2 * q + 1]});
        } else {
#pragma unroll
            for (int f = 0; f < F; ++f) held[f] = __float_as_uint(out[f]);
        }
        m_held = m;
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) st32<uint32_t>(planes, (uint32_t)v * plane_bytes + m_held * 4u, held[v]);
}

template <int F, bool H16, int LK, bool FMA>
__device__ __forceinline__ void forward_level_by_source(int sk, const LevelInfo& L, const float* __restrict__ table, const PointSrc& src,
                                                        float* __restrict__ planes, uint32_t plane_bytes, uint32_t M, uint32_t Mt, int chunk, int bpg) {
    if (sk == LNR_SRC_RAY_UNIFORM) forward_level_loop<F, H16, LK, FMA, LNR_SRC_RAY_UNIFORM>(L, table, src, planes, plane_bytes, M, Mt, chunk, bpg);
    else if (sk == LNR_SRC_PTS) forward_level_loop<F, H16, LK, FMA, LNR_SRC_PTS>(L, table, src, planes, plane_bytes, M, Mt, chunk, bpg);
    else forward_level_loop<F, H16, LK, FMA, LNR_SRC_RAY>(L, table, src, planes, plane_bytes, M, Mt, chunk, bpg);
}
template <int F, bool H16>
__device__ __forceinline__ void forward_level(const LevelInfo& L, const float* __restrict__ table, const PointSrc& src, float* __restrict__ planes,
                                              uint32_t plane_bytes, uint32_t M, uint32_t Mt, int chunk, int bpg) {
    const int sk = point_source_kind(src, 64u);
#define LNR_FWD_KIND(LK) do { if (L.pos_fma) forward_level_by_source<F, H16, LK, true>(sk, L, table, src, planes, plane_bytes, M, Mt, chunk, bpg); \
                              else forward_level_by_source<F, H16, LK, false>(sk, L, table, src, planes, plane_bytes, M, Mt, chunk, bpg); } while (0)
    if (L.hashed && L.pow2) LNR_FWD_KIND(LK_HASH_POW2);
    else if (!L.hashed) LNR_FWD_KIND(LK_DENSE);
    else LNR_FWD_KIND(LK_GENERIC);
#undef LNR_FWD_KIND
}

template <int F, bool H16>
__global__ void __launch_bounds__(ENC_BLOCK)
encode_forward_kernel(const LnrNetSpec spec, const float* __restrict__ table, const PointSrc src, float* __restrict__ feat,
                      int64_t m_pad, int bpg, uint32_t level_mask, int xcd_affine) {
    static_assert(!H16 || F >= 2, "half2 planes pair up the features of a level");
    int slot, chunk;
    if (!level_slot(bpg, __builtin_popcount(level_mask), xcd_affine != 0, slot, chunk)) return;
    const int lv = nth_level(level_mask, slot);
    const LevelInfo L = level_info(spec, lv);
    const uint32_t M = (uint32_t)live_points(src);
    // the MLP kernels read whole tiles of 16 (exact-chain fp32) / 32 (split-bf16 fp32 backward, fp16) samples: zero the ragged tail
    const uint32_t Mt = (M + 31u) / 32u * 32u;
    float* planes = feat + (size_t)(H16 ? lv * (F / 2) : lv * F) * m_pad;
    const uint32_t plane_bytes = (uint32_t)m_pad * 4u;
    forward_level<F, H16>(L, table, src, planes, plane_bytes, M, Mt, chunk, bpg);
}

// fp32 feature planes of the frequency encoding: four features per thread (the fp16 mode's pair planes: freq_forward_h16_kernel)
__global__ void __launch_bounds__(ENC_BLOCK)
freq_forward_kernel(const LnrNetSpec spec, const PointSrc src, float* __restrict__ feat, int64_t m_pad, int bpg) {
    const int group = blockIdx.x / bpg, chunk = blockIdx.x % bpg;       // 4 features per group
    const int64_t M = live_points(src);
    const int64_t M16 = (M + 15) / 16 * 16;
    for (int64_t m = (int64_t)chunk * ENC_BLOCK + threadIdx.x; m < M16; m += (int64_t)bpg * ENC_BLOCK) {
        float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (m < M) {
            float x[3];
            load_unit_point(src, m, x);
            freq_features4(spec, x, 4 * group, out);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * group + r < spec.enc_dim) feat[(size_t)(4 * group + r) * m_pad + m] = out[r];
    }
}

// Half2 pair planes of the frequency encoding (fp16 mode): one thread per (sample, dimension), every frequency's (sin, cos) pair
// from ONE range reduction.  The reference's features are sin(ph) and sin(rn(ph + pi/2)) with ph = rn(rn(x 2^f) pi)
// (oracle/encoding.py, tinycudann frequency.h); as two libm sinf calls per pair, four features per thread and the point re-derived by
// each of the 18 feature groups, this kernel took longer than the 128 x 2 MLP it feeds (165 vs 133 us at 2.1 M samples).  Here
//   k = rint(ph 2/pi), r = ph - k pi/2 (three-term Cody-Waite with fma), minimax sin / cos of r on [-pi/4, pi/4], quadrant from k & 3
// gives sin(ph), cos(ph) to ~1e-7, and the second feature is cos(ph + d) = cos(ph) - d sin(ph), d = (fl(pi/2) - pi/2) - e, where e is
// the rounding error of the fp32 addition ph + fl(pi/2) recovered exactly (TwoSum): 1.2e-7 from sin of the rounded sum (numpy check
// in docs/HISTORY.md 4.6), i.e. the same fp16 value except within 1e-7 of a rounding boundary (0.004 % of the features).
__global__ void __launch_bounds__(ENC_BLOCK)
freq_forward_h16_kernel(const LnrNetSpec spec, const PointSrc src, uint32_t* __restrict__ pairs, int64_t m_pad, int bpg) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const int dim = blockIdx.x / bpg, chunk = blockIdx.x % bpg;
    const int nf = spec.n_frequencies;
    const int64_t M = live_points(src);
    const int64_t M32 = (M + 31) / 32 * 32;
    for (int64_t m = (int64_t)chunk * ENC_BLOCK + threadIdx.x; m < M32; m += (int64_t)bpg * ENC_BLOCK) {
        float xv = 0.0f;
        const bool live = m < M;
        if (live) {
            float x[3];
            load_unit_point(src, m, x);
            xv = dim == 0 ? x[0] : (dim == 1 ? x[1] : x[2]);
        }
        for (int f = 0; f < nf; ++f) {
            const float mult = __uint_as_float((uint32_t)(127 + f) << 23);          // 2^f
            const float ph = lnr_mul_rn(lnr_mul_rn(xv, mult), LNR_PI_F);
            float s, c;
            sincos_f32(ph, &s, &c);
            const float h = lnr_add_rn(ph, LNR_PI_2_F);                              // the reference's second phase; e = its rounding error
            const float bb = lnr_add_rn(h, -ph);
            const float e = lnr_add_rn(lnr_add_rn(ph, -lnr_add_rn(h, -bb)), lnr_add_rn(LNR_PI_2_F, -bb));
            const float d = 4.371139000186243e-8f - e;
            const float c2 = __builtin_fmaf(-d, s, c);
            const uint32_t v = live ? __builtin_bit_cast(uint32_t, h2{(_Float16)s, (_Float16)c2}) : 0u;
            pairs[(size_t)(dim * nf + f) * (size_t)m_pad + (size_t)m] = v;                       // (64-bit index: up to 2^28 points per call)
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// Table-gradient records.  Float atomics to global memory top out at ~21 G/s on MI355X whatever the scope, locality
// or table size (profiles/r01_atomic_throughput.txt), and scattered 16-byte stores at ~87 G lines/s
// (profiles/r01_scatter_transactions.txt) - both far too slow for the 270 M corner updates of one default iteration.
// Instead each workgroup radix-partitions the records of a batch of 256 samples by owner slice IN LDS (histogram ->
// scan -> scatter into a staging buffer) and then copies the staging buffer out linearly: records of one owner are
// contiguous both in LDS and in that owner's (workgroup, owner) region in HBM, so the stores coalesce into full lines.
// table_grad_reduce2_kernel (lnr_density.hip) sums each owner's regions in LDS.
struct EncSink {
    float* grad_table;      // fallback target for records beyond a region's capacity (float atomics) ...
    long long* ovf;         // the levels' 64-bit overflow accumulators (LevelList::slab_off)
    void* regions;          // record regions, laid out by `plan` (lnr_density_api.h)
    RegionPlan plan;
    int* counts;            // [level][maxo][chunk]
    int maxo, shift;
    int xcd_affine;         // level_slot(): the launch's levels one per XCD (list.n a multiple of 8)
    int* ovf_flag;          // [level] workspace status words: a kernel that adds to a level's overflow accumulators stores `epoch` there, and
    int epoch;              // only then does the reduce read (and zero) them - the accumulators are all-zero between calls (lnr_density.hip)
#ifdef LNR_ABLATE
    int dbg;                // ablation bits (LNR_X_DBG), development builds only
#endif
    float combine_scale_max;
};

// Run-length combining on coarse levels: consecutive samples of a ray fall into the same cell there, so their 8 corner
// updates are summed before they become records.  Runs are confined to the 16-lane DPP rows of a wave: row shifts are
// plain VALU operands, whereas a 64-lane segmented sum needs ds_bpermute (LDS crossbar) for every step.
struct RunMask { bool take1, take2, take4, take8; };

// Run sums of N values at once, step by step (v += left neighbour where the run continues): one fused multiply-add per value and step (v += neighbour * {1, 0}: the DPP read folds
// into v_fmac_f32_dpp; fma(t, 1, v) rounds like t + v, fma(t, 0, v) = v), and consecutive instructions belong to different
// values, so that the DPP read-after-write wait states are filled with work.  The select form cost an add, a v_cndmask and two idle
// states per value and step.
struct RunMul { float m1, m2, m4, m8; };
__device__ __forceinline__ RunMul run_multipliers(const RunMask& k) {
    return RunMul{k.take1 ? 1.0f : 0.0f, k.take2 ? 1.0f : 0.0f, k.take4 ? 1.0f : 0.0f, k.take8 ? 1.0f : 0.0f};
}
template <int N>
__device__ __forceinline__ void row_run_sum_n(float (&v)[N], const RunMul& m) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_fmaf(row_down_f<1>(v[i]), m.m1, v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_fmaf(row_down_f<2>(v[i]), m.m2, v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_fmaf(row_down_f<4>(v[i]), m.m4, v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __builtin_fmaf(row_down_f<8>(v[i]), m.m8, v[i]);
}

// run masks of one 16-lane row: which lanes share the cell of their left neighbour
__device__ __forceinline__ void cell_runs(const Cell& c, int lane, bool& head, RunMask& run) {
    const int c16 = lane & 15;
    const int k1 = (int)(c.b[0] | (c.b[1] << 16)), k2 = (int)c.b[2];
    // the DPP reads must run with all lanes active: evaluate them before (not inside) any short-circuit logic
    const int p1 = row_up_i<1>(k1), p2 = row_up_i<1>(k2);
    head = (c16 == 0) | (p1 != k1) | (p2 != k2);
    int seg = head ? 1 : 0;
    int t;
    t = row_up_i<1>(seg); if (c16 >= 1) seg += t;
    t = row_up_i<2>(seg); if (c16 >= 2) seg += t;
    t = row_up_i<4>(seg); if (c16 >= 4) seg += t;
    t = row_up_i<8>(seg); if (c16 >= 8) seg += t;
    const int s1 = row_down_i<1>(seg), s2 = row_down_i<2>(seg), s4 = row_down_i<4>(seg), s8 = row_down_i<8>(seg);
    run.take1 = (c16 + 1 < 16) & (s1 == seg);
    run.take2 = (c16 + 2 < 16) & (s2 == seg);
    run.take4 = (c16 + 4 < 16) & (s4 == seg);
    run.take8 = (c16 + 8 < 16) & (s8 == seg);
}

// d(level features . g)/dx of one sample from its 8 corner entries
template <int F>
__device__ __forceinline__ void dx_from_entries(const LevelInfo& L, const Cell& c, const float g[F], const float tv[8][F], float dx[3]) {
    float dot[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        dot[k] = 0.0f;
#pragma unroll
        for (int f = 0; f < F; ++f) dot[k] += g[f] * tv[k][f];
    }
    const float fx = c.frac[0], fy = c.frac[1], fz = c.frac[2];
    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
    // differences along one axis, interpolated along the other two
    const float dxv = ((dot[1] - dot[0]) * gy + (dot[3] - dot[2]) * fy) * gz + ((dot[5] - dot[4]) * gy + (dot[7] - dot[6]) * fy) * fz;
    const float dyv = ((dot[2] - dot[0]) * gx + (dot[3] - dot[1]) * fx) * gz + ((dot[6] - dot[4]) * gx + (dot[7] - dot[5]) * fx) * fz;
    const float dzv = ((dot[4] - dot[0]) * gx + (dot[5] - dot[1]) * fx) * gy + ((dot[6] - dot[2]) * gx + (dot[7] - dot[3]) * fx) * fy;
    dx[0] = dxv * L.scale; dx[1] = dyv * L.scale; dx[2] = dzv * L.scale;
}

// d/dx modes of the backward kernels
#define ENC_DX_NONE 0
#define ENC_DX_PLANES 1       // one d/dx plane per level, summed later by sum_dx_planes_kernel (any point source)
#define ENC_DX_RAYS 2         // rays form with n_samples % 64 == 0: a wave's 64 samples lie on one ray, so its d/dx sum goes
                              // straight into that ray's record gradient (origin += dx/2, direction += z dx/2): six atomics per
                              // wave instead of three plane stores per thread, a 400 MB plane sum and the d_pts round trip

// all lanes active; dx = 0 on lanes without a contribution.
// Six wave sums at once: every value is first summed over its 16-lane row with four DPP butterfly steps (all lanes of a row end
// up with the row's sum), lane k < 6 of every row then keeps value k, and the four rows are added with the two cross-row
// swaps of CDNA4 (v_permlane16_swap / v_permlane32_swap): 24 + 5 + 6 VALU instructions instead of six independent
// reductions with four v_readlane each (71).  Fixed association, the same in every launch.
__device__ __forceinline__ float row_sum_all(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}
__device__ __forceinline__ float rows_sum_all(float v) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const unsigned a = __float_as_uint(v);
    const u2 r = __builtin_amdgcn_permlane16_swap(a, a, false, false);          // rows (0,1) and (2,3) meet
    const float s = __uint_as_float(r.x) + __uint_as_float(r.y);
    const unsigned b = __float_as_uint(s);
    const u2 q = __builtin_amdgcn_permlane32_swap(b, b, false, false);          // the two halves meet
    return __uint_as_float(q.x) + __uint_as_float(q.y);
}
__device__ __forceinline__ void ray_accumulate_dx(float* __restrict__ ray_acc_f, uint32_t ray, float z, const float dx[3], int lane, int n_rays_cap) {
    long long* ray_acc = reinterpret_cast<long long*>(ray_acc_f);        // [n_rays][6] fixed-point sums (passed through the float* d/dx argument)
    const int c16 = lane & 15;
    float v = row_sum_all(dx[0]);
    { const float t = row_sum_all(dx[1]); if (c16 == 1) v = t; }
    { const float t = row_sum_all(dx[2]); if (c16 == 2) v = t; }
    { const float t = row_sum_all(z * dx[0]); if (c16 == 3) v = t; }
    { const float t = row_sum_all(z * dx[1]); if (c16 == 4) v = t; }
    { const float t = row_sum_all(z * dx[2]); if (c16 == 5) v = t; }
    v = rows_sum_all(v);
    if (lane < 6) {
        // 64-bit fixed point: integer atomics add exactly, so the ray gradient does not depend on the order of the waves
        if (__builtin_expect(__builtin_isfinite(v), 1)) {
            if (v != 0.0f) atomicAdd(reinterpret_cast<unsigned long long*>(ray_acc) + (size_t)ray * 6 + lane,
                                     (unsigned long long)__float2ll_rn(0.5f * v * LNR_FIX_SCALE));         // x = (xyz + 1) / 2
        } else {
            // an integer sum cannot carry inf / NaN: the ray's word behind the sums says that one occurred, and ray_grad_apply_kernel
            // turns THAT ray's gradient into NaN - what a float sum would have produced, and what the pose check of the reference looks
            // for (optimizer.py:368-370); the other rays keep their gradients, so d_rays still says which keyframe failed
            ray_acc[(size_t)n_rays_cap * 6 + ray] = 1ll;
        }
    }
}

// d_rays[ray, 0:6] += the per-ray sums of ray_accumulate_dx (ray_acc[n_rays * 6 + ray] != 0: a non-finite term occurred on that ray -> NaN)
__global__ void __launch_bounds__(ENC_BLOCK)
ray_grad_apply_kernel(const long long* __restrict__ ray_acc, int n_rays, const int32_t* __restrict__ n_rays_dev, float* __restrict__ d_rays) {
    const int i = blockIdx.x * ENC_BLOCK + threadIdx.x;
    if (i >= lnr_live_rays(n_rays, n_rays_dev) * 6) return;
    const long long q = ray_acc[i];
    if (__builtin_expect(ray_acc[(size_t)n_rays * 6 + i / 6] != 0ll, 0)) d_rays[(size_t)(i / 6) * LNR_RAY_STRIDE + (i % 6)] = __builtin_nanf("");
    else if (q != 0ll) d_rays[(size_t)(i / 6) * LNR_RAY_STRIDE + (i % 6)] += (float)((double)q * (1.0 / (double)LNR_FIX_SCALE));
}

// The input gradient d(level features . g)/dx as a kernel of its own, two lanes per sample (see encode_forward_pair_kernel): level-major
// like the forward, no LDS, no barriers, 8 waves per SIMD.  Inside the partition kernel the term's gathers (the 8 corner entries of
// every live sample, one line each on the fine levels) sat in a wave's dependent chain between three workgroup barriers at 4 waves per
// SIMD, and the parts of that kernel ADDED (profiles/r03_ablate_binned_partition.txt); here they are all a wave does.
//   lane hx holds the dots d_r = g . entry(x = b0 + hx, row r) of the cell's four (y, z) rows;
//   d/dx from the differences to the partner lane's dots (even lanes only), d/dy and d/dz from differences of the lane's own dots,
//   weighted with the lane's x weight: per-lane PARTIAL sums that the per-ray reduction (or the pair add of the planes form) completes.
template <int F, int DXM>
__global__ void __launch_bounds__(ENC_BLOCK)
encode_dx_pair_kernel(const LnrNetSpec spec, const float* __restrict__ table, const PointSrc src, const float* __restrict__ dfeat,
                      float* __restrict__ dxl, int64_t m_pad, int bpg, uint32_t level_mask, int xcd_affine) {
    static_assert(DXM != ENC_DX_NONE, "nothing to compute");
    int slot, chunk;
    if (!level_slot(bpg, __builtin_popcount(level_mask), xcd_affine != 0, slot, chunk)) return;
    const int lv = nth_level(level_mask, slot);
    const LevelInfo L = level_info(spec, lv);
    const int lane = threadIdx.x & 63;
    const uint32_t hx = threadIdx.x & 1u;
    const uint32_t M = (uint32_t)live_points(src);
    if (M == 0u) return;
    const uint32_t step = (uint32_t)bpg * (ENC_BLOCK / 2);
    const uint32_t n_iter = (M + step - 1u) / step;                     // wave-uniform trip count: the body uses DPP
    const float* gplanes = dfeat + (size_t)(lv * F) * m_pad;
    float* dxplanes = dxl + (size_t)(lv * 3) * m_pad;
    const uint32_t plane_bytes = (uint32_t)m_pad * 4u;
    SampleCursor cur;
    cur.init((uint32_t)chunk * (ENC_BLOCK / 2) + (threadIdx.x >> 1), step, src.pts ? 1u : (uint32_t)src.n_samples);
    const uint32_t last_ray = src.pts ? 0u : (M - 1u) / cur.S;
    const bool uni = ray_uniform(src, 32u);
    for (uint32_t it = 0; it < n_iter; ++it) {
        const uint32_t m = cur.m;
        const bool live = m < M;
        const uint32_t mc = live ? m : M - 1u;                           // unconditional (clamped) loads
        float g[F];
        bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float v = ld32_stream<float>(gplanes, (uint32_t)f * plane_bytes + mc * 4u);
            g[f] = live ? v : 0.0f;
            any |= (g[f] != 0.0f);
        }
        RawPoint p;
        load_raw_point(src, mc, live ? cur.ray : last_ray, p, uni);
        const uint32_t ray_cur = cur.ray;
        cur.advance();
        const bool wave_any = __ballot(any) != 0ull;
        float dx[3] = {0.0f, 0.0f, 0.0f};
        if (wave_any) {                                                  // wave-uniform: all lanes active inside
            float x[3];
            unit_point(src, p, x);
            const Cell c = cell_of(L, x);
            uint32_t e[4];
            cell_entries_x(L, c, hx, e);
            float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (any) {                                                   // (a gather costs per ACTIVE lane and line: dead samples must not gather)
                float tv[4][F];
                gather_entries4<F>(table, e, tv);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int f = 0; f < F; ++f) d[r] += g[f] * tv[r][f];
            }
            float dd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) dd[r] = pair_partner_f(d[r]) - d[r];          // even lanes: dot(x + 1) - dot(x) of row r
            const float fx = c.frac[0], fy = c.frac[1], fz = c.frac[2];
            const float gy = 1.0f - fy, gz = 1.0f - fz;
            const float wxl = hx ? fx : 1.0f - fx;
            // differences along one axis, interpolated along the other two (rows: 0 = y0 z0, 1 = y1 z0, 2 = y0 z1, 3 = y1 z1)
            const float dxv = (dd[0] * gy + dd[1] * fy) * gz + (dd[2] * gy + dd[3] * fy) * fz;
            const float dyv = ((d[1] - d[0]) * gz + (d[3] - d[2]) * fz) * wxl;
            const float dzv = ((d[2] - d[0]) * gy + (d[3] - d[1]) * fy) * wxl;
            dx[0] = hx ? 0.0f : dxv * L.scale; dx[1] = dyv * L.scale; dx[2] = dzv * L.scale;
        }
        if constexpr (DXM == ENC_DX_RAYS) {
            // a wave's 32 samples lie on one ray (n_samples % 64 == 0, checked by the caller); lane 0 holds the wave's first sample
            if (wave_any) ray_accumulate_dx(dxl, (uint32_t)__builtin_amdgcn_readfirstlane((int)ray_cur), p.z, dx, lane, src.n_rays);
        } else {
            float s[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) s[k] = dx[k] + pair_partner_f(dx[k]);
            if (live && hx == 0u) {
#pragma unroll
                for (int k = 0; k < 3; ++k) st32<float>(dxplanes, (uint32_t)k * plane_bytes + m * 4u, s[k]);
            }
        }
    }
}

__device__ __forceinline__ void store_stream_b128(uint64_t global_addr, uint4 v) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(global_addr), "v"(q) : "memory");
}

#define ENC_BWD_BLOCK LNR_ENC_BWD_BLOCK
#define ENC_STAGE_RECORDS (ENC_BWD_BLOCK * 8)

// what the copy-out phase needs to know about one owner of the current batch: one 16-byte LDS read per record
struct OwnerSlot {
    uint32_t ptr_lo, ptr_hi;    // address of the owner's next free record in its region
    int scan;                   // first staged record of this owner
    int room;                   // records the region can still take
};

// An x-pair (lnr_density_api.h) that cannot become a record - its region is full, or its two corners straddle owner slices - is
// added to the level's 64-bit overflow accumulators with exactly the arithmetic the reduce applies to a record.
__device__ __forceinline__ void xpair_overflow(long long* ovf_level, uint32_t fi0_in_level, uint32_t fi1_in_level, uint32_t t, float a0, float a1, float fx) {
    uint32_t pi, tt; float q0, q1, qx;
    lnr_unpack_xpair(lnr_pack_xpair(0u, t, a0, a1, fx), pi, tt, q0, q1, qx);
    long long q[4];
    lnr_xpair_fix(q0, q1, qx, q);
    unsigned long long* o = reinterpret_cast<unsigned long long*>(ovf_level);
    atomicAdd(o + fi0_in_level, (unsigned long long)q[0]);
    atomicAdd(o + fi0_in_level + 1, (unsigned long long)q[1]);
    atomicAdd(o + fi1_in_level, (unsigned long long)q[2]);
    atomicAdd(o + fi1_in_level + 1, (unsigned long long)q[3]);
}

#ifdef LNR_PHASE_TIMING
__device__ unsigned long long lnr_phase_cycles[2 * LNR_N_PHASES];          // [8-byte record levels | x-pair levels]
#endif
// -DLNR_ABLATE (development builds only): LNR_X_DBG bits switch parts of encode_backward_kernel off, to time what is left
#ifdef LNR_ABLATE
#define DBG_SKIP(bit) ((sink.dbg & (bit)) != 0)
#else
#define DBG_SKIP(bit) false
#endif

// dynamic LDS: int cnt[maxo4], gcur[maxo4]; OwnerSlot slot[maxo] (16-byte aligned); then the staging buffer
// XP: the launch's levels take x-pair records (compile-time: the two record formats share little code, and a kernel that carries both
// spills ~60 SGPRs into VGPR lanes inside the batch loop)
template <int F, int DXM, bool XP>
#ifndef LNR_ENC_BWD_WAVES
#define LNR_ENC_BWD_WAVES 4
#endif
// 1: the d/dx term and the next batch's inputs are taken over in front of the record copy-out (see the batch loop): encode_backward
// 0.832 -> 0.783 ms, the iteration 1.918 -> 1.867 ms (profiles/r06_encode_backward_dx_before_copyout.txt); 0: the d/dx term after the passes
#ifndef LNR_ENC_DX_BEFORE_COPYOUT
#define LNR_ENC_DX_BEFORE_COPYOUT 1
#endif
__global__ void __launch_bounds__(ENC_BWD_BLOCK, LNR_ENC_BWD_WAVES)   // (max threads, min waves per SIMD): 4 = 128 VGPRs
encode_backward_kernel(const LnrNetSpec spec, const float* __restrict__ table, const PointSrc src, const float* __restrict__ dfeat,
                       float* __restrict__ dxl, int64_t m_pad, int bpg, const LevelList list, const EncSink sink) {
    constexpr bool WANT_DX = DXM != ENC_DX_NONE;          // dxl: the d/dx planes (ENC_DX_PLANES) or d_rays [n_rays,13] (ENC_DX_RAYS)
    extern __shared__ __attribute__((aligned(16))) int dyn[];
    __shared__ int s_total;
    constexpr bool PAIR = F >= 2;
    constexpr int NPASS = PAIR ? F / 2 : 1;
    const int maxo = sink.maxo;
    const int maxo4 = (maxo + 3) & ~3;
    int* cnt = dyn;
    int* gcur = dyn + maxo4;
    OwnerSlot* oslot = reinterpret_cast<OwnerSlot*>(dyn + 2 * maxo4);
    void* stage = reinterpret_cast<void*>(dyn + 2 * maxo4 + 4 * maxo);
    for (int i = threadIdx.x; i < maxo; i += ENC_BWD_BLOCK) { cnt[i] = 0; gcur[i] = 0; }
    __syncthreads();
    int slot, chunk;
    if (!level_slot(bpg, list.n, sink.xcd_affine != 0, slot, chunk)) return;          // (workgroup-uniform)
    const int lv = list.lv[slot];
    const LevelInfo L = level_info(spec, lv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t M = (uint32_t)live_points(src);
    const int first_owner = (int)(((uint64_t)L.offset * F) >> sink.shift);
    // x-pair records (12 bytes for the two x-neighbours of a corner pair) on the hashed power-of-two levels from LNR_XPAIR_SCALE_MIN up;
    // 8-byte records, run-length combined along the rays, on the coarser ones
    constexpr bool xp = XP && F == 2;
    const bool combine = !xp && L.scale < sink.combine_scale_max;
    const uint32_t rec_bytes = xp ? 12u : 8u;
    const uint32_t region_bytes = sink.plan.bytes[lv];
    const int cap_rec = (int)(region_bytes / rec_bytes);                   // a region's capacity in records of this level's format
    const char* level_regions = reinterpret_cast<const char*>(sink.regions) + sink.plan.off[lv];
    // region of (level, owner o, chunk): [level][owner][chunk] - the reduce of one owner streams its chunks' regions back to back
    const int ovf_off = list.slab_off[slot];
    long long* ovf = ovf_off >= 0 ? sink.ovf + ovf_off : nullptr;                // indexed by float index inside the level
    const uint32_t level_base = L.offset * F;
    const size_t region0 = (size_t)lv * maxo * bpg + chunk;
    const size_t region_step = (size_t)bpg;          // between consecutive owners
    const uint32_t step = (uint32_t)bpg * ENC_BWD_BLOCK;
    const uint32_t n_iter = (M + step - 1u) / step;          // workgroup-uniform trip count (the loop body has barriers)
    const float* gplanes = dfeat + (size_t)(lv * F) * m_pad;
    float* dxplanes = dxl + (size_t)(lv * 3) * m_pad;
    const uint32_t plane_bytes = (uint32_t)m_pad * 4u;
    const bool emit = sink.regions != nullptr;                // false: parameters frozen, only the input gradient is wanted
    if (M == 0u) {                                            // workgroup-uniform
        if (emit) for (int i = threadIdx.x; i < maxo; i += ENC_BWD_BLOCK) sink.counts[region0 + i * region_step] = 0;
        return;
    }

    // Software pipeline: the inputs of iteration it+1 (d_feature values, the point) are loaded while iteration it goes
    // through its three barriers, and the table gathers of the d/dx term are issued before them and consumed after.
    constexpr bool EARLY_DX = WANT_DX && F <= 2;
    SampleCursor cur;
    cur.init((uint32_t)chunk * ENC_BWD_BLOCK + threadIdx.x, step, src.pts ? 1u : (uint32_t)src.n_samples);
    const uint32_t last_ray = src.pts ? 0u : (M - 1u) / cur.S;
    const bool uni = ray_uniform(src, 64u);          // (workgroups of 256 / 512 threads: a wave's 64 samples start at a multiple of 64)
    float g_next[F];
    RawPoint p_next;
    {
        const bool in = cur.m < M;
        const uint32_t mc = in ? cur.m : M - 1u;
#pragma unroll
        for (int f = 0; f < F; ++f) g_next[f] = ld32_stream<float>(gplanes, (uint32_t)f * plane_bytes + mc * 4u);
        load_raw_point(src, mc, in ? cur.ray : last_ray, p_next, uni);
    }
    PHASE_INIT();
    for (uint32_t it = 0; it < n_iter; ++it) {
        PHASE(11);
        const uint32_t m = cur.m;
        const bool live = m < M;
        float g[F];
        bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) { g[f] = live ? g_next[f] : 0.0f; any |= (g[f] != 0.0f); }
        const RawPoint p_cur = p_next;
        const uint32_t ray_cur = cur.ray;
        cur.advance();
        {
            const bool in = cur.m < M;
            const uint32_t mc = in ? cur.m : M - 1u;          // unconditional (clamped) loads: a static number in flight
#pragma unroll
            for (int f = 0; f < F; ++f) g_next[f] = DBG_SKIP(32) ? ld32<float>(gplanes, (uint32_t)f * plane_bytes + mc * 4u) : ld32_stream<float>(gplanes, (uint32_t)f * plane_bytes + mc * 4u);
            load_raw_point(src, mc, in ? cur.ray : last_ray, p_next, uni);
        }
        const bool wave_any = __ballot(any) != 0ull;
        PHASE(0);
        Cell c;
        uint32_t e[8];
        float w[8];
        bool head = true;
        RunMask run = {false, false, false, false};
        float tv[EARLY_DX ? 8 : 1][F];
        if (wave_any) {
            float x[3];
            unit_point(src, p_cur, x);
            c = cell_of(L, x);
            cell_entries(L, c, e);
            // (a gather costs per ACTIVE lane and line - tools/gather_bench.hip - and on the fine levels no two lanes share a line:
            // the 48 % of the samples without a gradient must not gather)
            if constexpr (EARLY_DX) { if (any && !DBG_SKIP(4)) gather_entries<F>(table, e, tv); }
            cell_weights(c, w);
            // runs = consecutive samples (lanes of one 16-lane row) in the same CELL, not merely the same hashed entry
            if (combine) cell_runs(c, lane, head, run);
        }
        PHASE(1);
#pragma unroll
        for (int pass = 0; pass < (emit && !DBG_SKIP(16) ? NPASS : 0); ++pass) {
            // ---- A: this thread's records of the batch; rank within the owner's bucket from an LDS histogram
            float rv0[8], rv1[8]; int rrank[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) rrank[k] = -1;
            uint32_t xt = 0u;                       // x-pair levels: trailing one bits of the cell's x (e(x+1) = e(x) ^ (2^(t+1) - 1))
            if (wave_any && xp) {
                xt = (uint32_t)__builtin_ctz(~c.b[0]);
                const float wy[2] = {1.0f - c.frac[1], c.frac[1]}, wz[2] = {1.0f - c.frac[2], c.frac[2]};
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const float wyz = wy[k2 & 1] * wz[k2 >> 1];
                    const float a0 = wyz * g[0], a1 = wyz * g[F >= 2 ? 1 : 0];
                    rv0[k2] = a0; rv1[k2] = a1;
                    if ((a0 != 0.0f) | (a1 != 0.0f)) {
                        const uint32_t fi = e[2 * k2] * F;
                        if (xt < 12u) rrank[k2] = atomicAdd(&cnt[(int)(fi >> sink.shift) - first_owner], 1);
                        else {
                            // x ends in >= 12 one bits (2^-12 of the cells): e(x + 1) = e(x) ^ (2^(t+1) - 1) lies in ANOTHER owner slice.  Two
                            // single-corner records instead of one x-pair - each an x-pair record with t = 0, fx = 0, whose one live corner
                            // carries the x weight already (the reduce adds a to the record's entry and 0 to its neighbour) - so that these
                            // cells no longer go through the overflow accumulators in every launch (round 2-3: ~8 k of them per launch kept
                            // all 59 MB of accumulators in play)
                            const uint32_t fj = e[2 * k2 + 1] * F;
                            const float gx = 1.0f - c.frac[0];
                            rv0[k2] = gx * a0; rv1[k2] = gx * a1;
                            rv0[4 + k2] = c.frac[0] * a0; rv1[4 + k2] = c.frac[0] * a1;
                            rrank[k2] = atomicAdd(&cnt[(int)(fi >> sink.shift) - first_owner], 1);
                            rrank[4 + k2] = atomicAdd(&cnt[(int)(fj >> sink.shift) - first_owner], 1);
                        }
                    }
                }
            } else if (wave_any) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { rv0[k] = w[k] * g[PAIR ? 2 * pass : 0]; rv1[k] = PAIR ? w[k] * g[2 * pass + 1] : 0.0f; }
                if (combine) {                       // (all corners of a step together: one v_fmac_f32_dpp per value and step, no idle states)
                    const RunMul rm = run_multipliers(run);
                    row_run_sum_n<8>(rv0, rm);
                    if (PAIR) row_run_sum_n<8>(rv1, rm);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float v0 = rv0[k], v1 = rv1[k];
                    if (head & ((v0 != 0.0f) | (v1 != 0.0f))) {
                        const uint32_t fi = e[k] * F + (PAIR ? 2 * pass : 0);
                        const int local = (int)(fi >> sink.shift) - first_owner;
                        if (local >= 0 && local < maxo) rrank[k] = atomicAdd(&cnt[local], 1);
                        else { atomicAdd(sink.grad_table + fi, v0); if (PAIR) atomicAdd(sink.grad_table + fi + 1, v1); }
                    }
                }
            }
            PHASE(2);
            __syncthreads();
            PHASE(3);
            // ---- B: exclusive scan of the histogram; reserve the slots in the regions
            if (wave == 0) {
                int running = 0;
                for (int o0 = 0; o0 < maxo; o0 += 64) {
                    const int o = o0 + lane;
                    const int n = o < maxo ? cnt[o] : 0;
                    int incl = n;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
                    if (o < maxo) {
                        const int have = gcur[o];
                        const uint64_t p = reinterpret_cast<uint64_t>(level_regions) + ((size_t)o * bpg + chunk) * (size_t)region_bytes + (size_t)have * rec_bytes;
                        OwnerSlot os;
                        os.ptr_lo = (uint32_t)p; os.ptr_hi = (uint32_t)(p >> 32);
                        os.scan = running + incl - n;
                        os.room = cap_rec - have;
                        oslot[o] = os;
                        gcur[o] = have + n;
                        cnt[o] = 0;
                    }
                    running += __shfl(incl, 63, 64);
                }
                if (lane == 0) s_total = running;
            }
            PHASE(4);
            __syncthreads();
            PHASE(5);
            // ---- C: scatter into the staging buffer, grouped by owner
            if (xp) {
                // x-pair: (float index of the x corner, a0, a1, fx rounded to 28 bits with t in the freed low nibble)
                const bool straddle = xt >= 12u;                 // single-corner records: t = 0, fx = 0
                const uint32_t fxt = straddle ? 0u : ((((__float_as_uint(c.frac[0]) + 8u) & ~0xFu)) | xt);
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    if (rrank[k2] >= 0) {
                        const uint32_t fi = e[2 * k2] * F;
                        const int at = oslot[(int)(fi >> sink.shift) - first_owner].scan + rrank[k2];
                        reinterpret_cast<uint4*>(stage)[at] = make_uint4(fi, __float_as_uint(rv0[k2]), __float_as_uint(rv1[k2]), fxt);
                    }
                }
                if (straddle) {
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) {
                        if (rrank[4 + k2] >= 0) {
                            const uint32_t fj = e[2 * k2 + 1] * F;
                            const int at = oslot[(int)(fj >> sink.shift) - first_owner].scan + rrank[4 + k2];
                            reinterpret_cast<uint4*>(stage)[at] = make_uint4(fj, __float_as_uint(rv0[4 + k2]), __float_as_uint(rv1[4 + k2]), 0u);
                        }
                    }
                }
            } else
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (rrank[k] >= 0) {
                    const uint32_t fi = e[k] * F + (PAIR ? 2 * pass : 0);
                    const int local = (int)(fi >> sink.shift) - first_owner;
                    const int at = oslot[local].scan + rrank[k];
                    if (PAIR) reinterpret_cast<uint4*>(stage)[at] = make_uint4(fi, __float_as_uint(rv0[k]), __float_as_uint(rv1[k]), 0u);
                    else reinterpret_cast<uint2*>(stage)[at] = make_uint2(fi, __float_as_uint(rv0[k]));
                }
            }
            PHASE(6);
            __syncthreads();
            PHASE(7);
#if LNR_ENC_DX_BEFORE_COPYOUT
            // Everything this batch still WAITS for from memory is taken over here, in front of the copy-out: the next batch's inputs
            // (requested at the top) and the gathers of the d/dx term.  Loads and stores share one in-order counter (vmcnt) and the record
            // stores below are invisible to the compiler (inline asm, data-dependent trip count): a wait placed behind them - the d/dx
            // term after the passes, the inputs at the loop's end - is a wait for the stores' acknowledgement as well.
            if (pass == NPASS - 1) {
#pragma unroll
                for (int f = 0; f < F; ++f) asm volatile("" : "+v"(g_next[f]));
                asm volatile("" : "+v"(p_next.z));
                if constexpr (WANT_DX) {
                    float dx[3] = {0.0f, 0.0f, 0.0f};
                    if (any && !DBG_SKIP(2)) {
                        if constexpr (EARLY_DX) dx_from_entries<F>(L, c, g, tv, dx);
                        else { float tl[8][F]; gather_entries<F>(table, e, tl); dx_from_entries<F>(L, c, g, tl, dx); }
                    }
                    if constexpr (DXM == ENC_DX_RAYS) {
                        if (wave_any && !DBG_SKIP(1))            // wave-uniform; lane 0 holds the wave's first (live) sample
                            ray_accumulate_dx(dxl, (uint32_t)__builtin_amdgcn_readfirstlane((int)ray_cur), p_cur.z, dx, lane, src.n_rays);
                    } else if (live) {
#pragma unroll
                        for (int d = 0; d < 3; ++d) st32<float>(dxplanes, (uint32_t)d * plane_bytes + m * 4u, dx[d]);
                    }
                }
            }
#endif
            // ---- D: linear copy-out; neighbouring lanes write neighbouring records of the same region
            const int total = s_total;
            if (xp) {
                for (int i = threadIdx.x; i < total; i += ENC_BWD_BLOCK) {
                    const uint4 r4 = reinterpret_cast<const uint4*>(stage)[i];
                    const uint32_t idx = r4.x, t = r4.w & 0xFu;
                    const float a0 = __uint_as_float(r4.y), a1 = __uint_as_float(r4.z), fx = __uint_as_float(r4.w & ~0xFu);
                    const OwnerSlot os = oslot[(int)(idx >> sink.shift) - first_owner];
                    const int k = i - os.scan;
                    if (DBG_SKIP(8)) continue;
                    if (k < os.room) {
                        const LnrXRec rec = lnr_pack_xpair((idx & ((1u << LNR_SLICE_SHIFT) - 1u)) >> 1, t, a0, a1, fx);
                        store_stream_b96((((uint64_t)os.ptr_hi << 32) | os.ptr_lo) + (uint64_t)k * 12u, rec.a, rec.b, rec.c);
                    } else {
                        const uint32_t in_level = idx - level_base;                       // e1 = e0 ^ (2^(t+1) - 1): float index ^ (mask << 1)
                        xpair_overflow(ovf, in_level, in_level ^ (((2u << t) - 1u) << 1), t, a0, a1, fx);
                        sink.ovf_flag[lv] = sink.epoch;
                    }
                }
            } else
            for (int i = threadIdx.x; i < total; i += ENC_BWD_BLOCK) {
                uint32_t idx; float v0, v1 = 0.0f;
                uint2 r2;
                if (PAIR) { const uint4 r4 = reinterpret_cast<const uint4*>(stage)[i]; idx = r4.x; v0 = __uint_as_float(r4.y); v1 = __uint_as_float(r4.z); }
                else { r2 = reinterpret_cast<const uint2*>(stage)[i]; idx = r2.x; v0 = __uint_as_float(r2.y); }
                const int local = (int)(idx >> sink.shift) - first_owner;
                const OwnerSlot os = oslot[local];
                const int k = i - os.scan;
                if (DBG_SKIP(8)) continue;
                if (k < os.room) {
                    if (PAIR) r2 = lnr_pack_pair((idx & ((1u << LNR_SLICE_SHIFT) - 1u)) >> 1, v0, v1);
                    store_stream_b64((((uint64_t)os.ptr_hi << 32) | os.ptr_lo) + (uint64_t)k * 8u, r2);
                } else if (ovf) {
                    // same 26-bit rounding as a packed record: which records overflow depends on arrival order, the sum must not
                    sink.ovf_flag[lv] = sink.epoch;
                    const float q0 = PAIR ? __uint_as_float(lnr_pack26(v0) << 6) : v0, q1 = PAIR ? __uint_as_float(lnr_pack26(v1) << 6) : 0.0f;
                    if (q0 != 0.0f) atomicAdd(reinterpret_cast<unsigned long long*>(ovf + (idx - level_base)), (unsigned long long)lnr_to_fix(q0));
                    if (PAIR && q1 != 0.0f) atomicAdd(reinterpret_cast<unsigned long long*>(ovf + (idx - level_base) + 1), (unsigned long long)lnr_to_fix(q1));
                } else {
                    atomicAdd(sink.grad_table + idx, v0);
                    if (PAIR) atomicAdd(sink.grad_table + idx + 1, v1);
                }
            }
            // no barrier here: the next histogram only touches cnt[], and its first barrier orders D before the next B/C
            PHASE(8);
        }
#if LNR_ENC_DX_BEFORE_COPYOUT
        if (!(emit && !DBG_SKIP(16))) {                       // (no passes ran: parameters frozen)
            if constexpr (WANT_DX) {
                float dx[3] = {0.0f, 0.0f, 0.0f};
                if (any && !DBG_SKIP(2)) {
                    if constexpr (EARLY_DX) dx_from_entries<F>(L, c, g, tv, dx);
                    else { float tl[8][F]; gather_entries<F>(table, e, tl); dx_from_entries<F>(L, c, g, tl, dx); }
                }
                if constexpr (DXM == ENC_DX_RAYS) {
                    if (wave_any && !DBG_SKIP(1))            // wave-uniform; lane 0 holds the wave's first (live) sample
                        ray_accumulate_dx(dxl, (uint32_t)__builtin_amdgcn_readfirstlane((int)ray_cur), p_cur.z, dx, lane, src.n_rays);
                } else if (live) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) st32<float>(dxplanes, (uint32_t)d * plane_bytes + m * 4u, dx[d]);
                }
            }
        }
#else
        if constexpr (WANT_DX) {
            float dx[3] = {0.0f, 0.0f, 0.0f};
            if (any && !DBG_SKIP(2)) {
                if constexpr (EARLY_DX) dx_from_entries<F>(L, c, g, tv, dx);
                else { float tl[8][F]; gather_entries<F>(table, e, tl); dx_from_entries<F>(L, c, g, tl, dx); }
            }
            if constexpr (DXM == ENC_DX_RAYS) {
                if (wave_any && !DBG_SKIP(1))            // wave-uniform; lane 0 holds the wave's first (live) sample
                    ray_accumulate_dx(dxl, (uint32_t)__builtin_amdgcn_readfirstlane((int)ray_cur), p_cur.z, dx, lane, src.n_rays);
            } else if (live) {
#pragma unroll
                for (int d = 0; d < 3; ++d) st32<float>(dxplanes, (uint32_t)d * plane_bytes + m * 4u, dx[d]);
            }
        }
#endif
        PHASE(9);
    }
    PHASE_FLUSH(lnr_phase_cycles, xp ? LNR_N_PHASES : 0);
    __syncthreads();
    if (emit)
        for (int i = threadIdx.x; i < maxo; i += ENC_BWD_BLOCK)
            sink.counts[region0 + i * region_step] = gcur[i] < cap_rec ? gcur[i] : cap_rec;
}

// ------------------------------------------------------------------------------------------------ binned partition
// The same partition for HASHED levels, where the records of a batch spread evenly over the owners: every owner has a fixed BIN
// of LDS, a record takes its place with one returning atomic on the bin's fill (no histogram pass, no scan, no OwnerSlot lookups:
// two barriers per batch instead of three and no single-wave phase), and the bin IS the staging buffer - packed records, appended
// in arrival order.  What leaves the workgroup is whole 128-byte lines: after the barrier wave w copies the full lines of its
// eight owners' bins out (one ds_read_b128 + one global_store_dwordx4 per owner, region address and line count on the scalar
// unit) and moves the unfinished line - the tail, < 128 bytes - to the front of the bin, where the next batch appends.  A region is
// therefore a dense byte stream of records written line by line; only the last line of a region, flushed when the kernel ends, is
// partial (round 2 measured the partial-line appends of the per-batch copy-out at 0.25-0.38 of 0.94 ms against 0.12 ms for the
// same bytes as whole lines, and the per-owner bookkeeping of a first whole-line attempt as costlier than that gain: here the
// bookkeeping is a lane-resident {position, tail} pair per owner and nothing is looked up per record).
// A record that finds its bin full (BIN_BYTES covers > 5 sigma of a batch's records per owner plus the tail), or whose region is
// closed (no room left for another worst-case batch), takes the overflow path with the rounding of a packed record, as before.
#define ENC_BIN_BYTES LNR_BIN_BYTES
#define ENC_BIN_CLOSED 0x40000000
static_assert(LNR_BIN_BYTES == 64 * 16, "a wave copies a whole bin with one 16-byte load per lane");

// NW = waves per workgroup (8: 512 threads, 1 KB bins; 4: 256 threads, 512-byte bins - twice as many independent barrier domains per CU
// at the same LDS and wave count; a region only ever receives whole lines, so the batch size does not change what reaches HBM)
template <int F, int DXM, bool XP, int NW>
__global__ void __launch_bounds__(64 * NW, 4)
encode_backward_binned_kernel(const LnrNetSpec spec, const float* __restrict__ table, const PointSrc src, const float* __restrict__ dfeat,
                              float* __restrict__ dxl, int64_t m_pad, int bpg, const LevelList list, const EncSink sink) {
    constexpr bool WANT_DX = DXM != ENC_DX_NONE;
    static_assert(F >= 2, "binned records are pair records");
    static_assert(NW == 8 || NW == 4, "8 waves x 8 owners or 4 waves x 16 owners");
    constexpr int BLK = 64 * NW;                                  // threads = samples of a batch
    constexpr int BIN = 128 * NW;                                 // bytes of an owner's bin: tail (< 128) + a batch's records with > 5 sigma to spare
    constexpr int OPW = 64 / NW;                                  // owners served by a wave
    constexpr int G = NW;                                         // lanes that hold the state of one owner (G x 128 / G bytes move its tail)
    constexpr int LPB = BIN / 16;                                 // lanes that copy one bin (16 bytes each)
    constexpr int BPI = 64 / LPB;                                 // bins per copy instruction
    constexpr int TQ = 128 / G / 16;                              // 16-byte pieces of the tail per lane
    extern __shared__ __attribute__((aligned(16))) int dyn[];
    constexpr int NPASS = F / 2;
    constexpr bool xp = XP && F == 2;
    constexpr uint32_t REC = xp ? 12u : 8u;
    int* fill = dyn;                                             // [64] bytes in the owner's bin (tail + this batch's records)
    char* stage = reinterpret_cast<char*>(dyn + 64);             // [64][BIN]
    const int maxo = sink.maxo;                                  // <= 64 (host)
    int slot, chunk;
    if (!level_slot(bpg, list.n, sink.xcd_affine != 0, slot, chunk)) return;          // (workgroup-uniform)
    const int lv = list.lv[slot];
    const LevelInfo L = level_info(spec, lv);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t M = (uint32_t)live_points(src);
    const int first_owner = (int)(((uint64_t)L.offset * F) >> sink.shift);
    const bool combine = !xp && L.scale < sink.combine_scale_max;
    const uint32_t region_bytes = sink.plan.bytes[lv];
    char* level_regions = reinterpret_cast<char*>(sink.regions) + sink.plan.off[lv];
    const int ovf_off = list.slab_off[slot];
    long long* ovf = sink.ovf + ovf_off;
    const uint32_t level_base = L.offset * F;
    const size_t region0 = (size_t)lv * maxo * bpg + chunk;
    const size_t region_step = (size_t)bpg;
    const uint32_t step = (uint32_t)bpg * BLK;
    const uint32_t n_iter = (M + step - 1u) / step;
    const float* gplanes = dfeat + (size_t)(lv * F) * m_pad;
    float* dxplanes = dxl + (size_t)(lv * 3) * m_pad;
    const uint32_t plane_bytes = (uint32_t)m_pad * 4u;
    const bool emit = sink.regions != nullptr;
    // owner bookkeeping: lanes G k .. G k + G - 1 of wave w all hold the state of owner OPW w + k
    const int my_owner = wave * OPW + lane / G;
    const bool own = my_owner < maxo;
    uint32_t gpos = 0u;                                           // bytes of the owner's region already written (multiple of 128)
    uint32_t tail = 0u;                                           // bytes of the unfinished line at the front of the bin
    // a region too small for one worst-case batch never opens (LNR_BWD_TABLE_ATOMICS: region_bytes == 0)
    bool closed = !own || region_bytes < (uint32_t)(BIN + 128);
    if (threadIdx.x < 64) fill[threadIdx.x] = 0;
    __syncthreads();
    if (lane % G == 0 && own && closed) fill[my_owner] = ENC_BIN_CLOSED;
    __syncthreads();
    if (M == 0u) {
        if (emit) for (int i = threadIdx.x; i < maxo; i += BLK) sink.counts[region0 + i * region_step] = 0;
        return;
    }
    constexpr bool EARLY_DX = WANT_DX && F <= 2;
    SampleCursor cur;
    cur.init((uint32_t)chunk * BLK + threadIdx.x, step, src.pts ? 1u : (uint32_t)src.n_samples);
    const uint32_t last_ray = src.pts ? 0u : (M - 1u) / cur.S;
    const bool uni = ray_uniform(src, 64u);          // (workgroups of 256 / 512 threads: a wave's 64 samples start at a multiple of 64)
    float g_next[F];
    RawPoint p_next;
    {
        const bool in = cur.m < M;
        const uint32_t mc = in ? cur.m : M - 1u;
#pragma unroll
        for (int f = 0; f < F; ++f) g_next[f] = ld32_stream<float>(gplanes, (uint32_t)f * plane_bytes + mc * 4u);
        load_raw_point(src, mc, in ? cur.ray : last_ray, p_next, uni);
    }
    for (uint32_t it = 0; it < n_iter; ++it) {
        const uint32_t m = cur.m;
        const bool live = m < M;
        float g[F];
        bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) { g[f] = live ? g_next[f] : 0.0f; any |= (g[f] != 0.0f); }
        const RawPoint p_cur = p_next;
        const uint32_t ray_cur = cur.ray;
        cur.advance();
        {
            const bool in = cur.m < M;
            const uint32_t mc = in ? cur.m : M - 1u;
#pragma unroll
            for (int f = 0; f < F; ++f) g_next[f] = ld32_stream<float>(gplanes, (uint32_t)f * plane_bytes + mc * 4u);
            load_raw_point(src, mc, in ? cur.ray : last_ray, p_next, uni);
        }
        const bool wave_any = __ballot(any) != 0ull;
        Cell c;
        uint32_t e[8];
        float w[8];
        bool head = true;
        RunMask run = {false, false, false, false};
        float tv[EARLY_DX ? 8 : 1][F];
        if (wave_any) {
            float x[3];
            unit_point(src, p_cur, x);
            c = cell_of(L, x);
            cell_entries(L, c, e);
            if constexpr (EARLY_DX) { if (any && !DBG_SKIP(4)) gather_entries<F>(table, e, tv); }
            cell_weights(c, w);
            if (combine) cell_runs(c, lane, head, run);
        }
#pragma unroll
        for (int pass = 0; pass < (emit && !DBG_SKIP(16) ? NPASS : 0); ++pass) {
            // ---- A: every record is packed and put into its owner's bin at the offset a returning atomic hands out.  All of a
            // thread's atomics are issued before the first offset is used: one LDS round trip per batch instead of one per record
            // (measured: the rank atomics cost 0.11 of the x-pair levels' 0.54 ms when each was waited for in turn).
            constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;
            if (wave_any && xp) {
                const uint32_t xt = (uint32_t)__builtin_ctz(~c.b[0]);
                const float wy[2] = {1.0f - c.frac[1], c.frac[1]}, wz[2] = {1.0f - c.frac[2], c.frac[2]};
                float a0[4], a1[4];
                uint32_t at[4];
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const float wyz = wy[k2 & 1] * wz[k2 >> 1];
                    a0[k2] = wyz * g[0]; a1[k2] = wyz * g[1];
                    at[k2] = NO_SLOT;
                    if (((a0[k2] != 0.0f) | (a1[k2] != 0.0f)) & (xt < 12u)) {
                        const int o = (int)((e[2 * k2] * F) >> LNR_SLICE_SHIFT) - first_owner;
                        at[k2] = DBG_SKIP(64) ? (uint32_t)((threadIdx.x * 4 + k2) & 31) * REC : (uint32_t)atomicAdd(&fill[o], (int)REC);
                    }
                }
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    if ((a0[k2] != 0.0f) | (a1[k2] != 0.0f)) {
                        const uint32_t fi = e[2 * k2] * F;
                        if (at[k2] <= (uint32_t)BIN - REC) {
                            if (!DBG_SKIP(128)) {
                                const int o = (int)(fi >> LNR_SLICE_SHIFT) - first_owner;
                                const LnrXRec rec = lnr_pack_xpair((fi & ((1u << LNR_SLICE_SHIFT) - 1u)) >> 1, xt, a0[k2], a1[k2], c.frac[0]);
                                uint32_t* dst = reinterpret_cast<uint32_t*>(stage + (uint32_t)o * BIN + at[k2]);
                                dst[0] = rec.a; dst[1] = rec.b; dst[2] = rec.c;
                            }
                        } else {
                            // bin full or region closed: same rounding as a packed record.  Corners that straddle owner slices (xt >= 12): as
                            // the scan partition's two single-corner records (t = 0, fx = 0, the x weight applied here), so that the two
                            // partitions agree to the bit
                            if (xt < 12u) xpair_overflow(ovf, fi - level_base, e[2 * k2 + 1] * F - level_base, xt, a0[k2], a1[k2], c.frac[0]);
                            else {
                                const uint32_t i0 = fi - level_base, i1 = e[2 * k2 + 1] * F - level_base;
                                const float gx = 1.0f - c.frac[0];
                                xpair_overflow(ovf, i0, i0 ^ 2u, 0u, gx * a0[k2], gx * a1[k2], 0.0f);
                                xpair_overflow(ovf, i1, i1 ^ 2u, 0u, c.frac[0] * a0[k2], c.frac[0] * a1[k2], 0.0f);
                            }
                            sink.ovf_flag[lv] = sink.epoch;
                        }
                    }
                }
            } else if (wave_any) {
                float v0[8], v1[8];
                uint32_t at[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { v0[k] = w[k] * g[2 * pass]; v1[k] = w[k] * g[2 * pass + 1]; }
                if (combine) {
                    const RunMul rm = run_multipliers(run);
                    row_run_sum_n<8>(v0, rm);
                    row_run_sum_n<8>(v1, rm);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    at[k] = NO_SLOT;
                    if (head & ((v0[k] != 0.0f) | (v1[k] != 0.0f))) {
                        const int o = (int)((e[k] * F + 2 * pass) >> LNR_SLICE_SHIFT) - first_owner;
                        at[k] = (uint32_t)atomicAdd(&fill[o], (int)REC);
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (at[k] != NO_SLOT) {
                        const uint32_t fi = e[k] * F + 2 * pass;
                        if (at[k] <= (uint32_t)BIN - REC) {
                            const int o = (int)(fi >> LNR_SLICE_SHIFT) - first_owner;
                            *reinterpret_cast<uint2*>(stage + (uint32_t)o * BIN + at[k]) = lnr_pack_pair((fi & ((1u << LNR_SLICE_SHIFT) - 1u)) >> 1, v0[k], v1[k]);
                        } else {
                            sink.ovf_flag[lv] = sink.epoch;
                            const float q0 = __uint_as_float(lnr_pack26(v0[k]) << 6), q1 = __uint_as_float(lnr_pack26(v1[k]) << 6);
                            if (q0 != 0.0f) atomicAdd(reinterpret_cast<unsigned long long*>(ovf + (fi - level_base)), (unsigned long long)lnr_to_fix(q0));
                            if (q1 != 0.0f) atomicAdd(reinterpret_cast<unsigned long long*>(ovf + (fi - level_base) + 1), (unsigned long long)lnr_to_fix(q1));
                        }
                    }
                }
            }
            if (!DBG_SKIP(256)) __syncthreads();
            // ---- B: whole lines out, tail to the front
            if (!DBG_SKIP(32)) {
                const uint32_t f_raw = own ? (uint32_t)fill[my_owner] : 0u;
                uint32_t valid = 0u;
                if (!closed) {
                    const uint32_t n_new = (f_raw - tail) / REC, n_room = ((uint32_t)BIN - tail) / REC;
                    valid = tail + (n_new < n_room ? n_new : n_room) * REC;
                }
                const uint32_t nlines = valid >> 7, rem = valid & 127u;
                const char* bins = stage + (uint32_t)(wave * OPW) * BIN;
                // tail: the G lanes of an owner move the (< 128) bytes behind its last whole line to the front of its bin
                // (LDS operations of a wave execute in order: the line reads below come first, the tail's read next, its write last)
                const uint32_t sub = (uint32_t)(lane % G) * (128u / G);
                const bool mv = nlines > 0u && sub < rem;
                char* my_bin = stage + (uint32_t)my_owner * BIN;
                const int half = lane / LPB, piece = lane % LPB;                                            // which bin of a copy instruction, which 16 bytes of it
#pragma unroll
                for (int j0 = 0; j0 < OPW / BPI; j0 += 4) {
                    uint4 v[4];
                    uint32_t n16[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // line count of the bin this lane copies from: wave-uniform per bin, selected per lane when an instruction covers two bins
                        n16[j] = (uint32_t)__builtin_amdgcn_readlane((int)nlines, ((j0 + j) * BPI) * G) * 8u;
                        if constexpr (BPI == 2) { const uint32_t n1 = (uint32_t)__builtin_amdgcn_readlane((int)nlines, ((j0 + j) * BPI + 1) * G) * 8u; if (half) n16[j] = n1; }
                        if ((uint32_t)piece < n16[j]) v[j] = *reinterpret_cast<const uint4*>(bins + (j0 + j) * (BPI * BIN) + lane * 16);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int o0 = wave * OPW + (j0 + j) * BPI;
                        uint64_t dst = reinterpret_cast<uint64_t>(level_regions) + ((size_t)o0 * bpg + chunk) * (size_t)region_bytes +
                                       (uint32_t)__builtin_amdgcn_readlane((int)gpos, ((j0 + j) * BPI) * G);
                        if constexpr (BPI == 2) {
                            const uint64_t d1 = reinterpret_cast<uint64_t>(level_regions) + ((size_t)(o0 + 1) * bpg + chunk) * (size_t)region_bytes +
                                                (uint32_t)__builtin_amdgcn_readlane((int)gpos, ((j0 + j) * BPI + 1) * G);
                            if (half) dst = d1;
                        }
                        if ((uint32_t)piece < n16[j] && !DBG_SKIP(8)) store_stream_b128(dst + (uint32_t)piece * 16u, v[j]);
                    }
                }
                if (mv) {
                    uint4 t4[TQ];
#pragma unroll
                    for (int q = 0; q < TQ; ++q) if (sub + 16u * q < rem) t4[q] = *reinterpret_cast<const uint4*>(my_bin + (nlines << 7) + sub + 16u * q);
#pragma unroll
                    for (int q = 0; q < TQ; ++q) if (sub + 16u * q < rem) *reinterpret_cast<uint4*>(my_bin + sub + 16u * q) = t4[q];
                }
                gpos += nlines << 7;
                tail = rem;
                if (!closed && gpos + (uint32_t)(BIN + 128) > region_bytes) {
                    // no room for another worst-case batch: the tail - now at the front of the bin - goes out as the region's last
                    // (partial) line and the owner closes; whatever is addressed to it from here on takes the overflow path
                    char* dst = level_regions + ((size_t)my_owner * bpg + chunk) * (size_t)region_bytes + gpos;
#pragma unroll
                    for (int q = 0; q < 32 / G; ++q) {
                        const uint32_t off = sub + 4u * q;
                        if (off < rem) *reinterpret_cast<uint32_t*>(dst + off) = *reinterpret_cast<const uint32_t*>(my_bin + off);
                    }
                    gpos += rem;
                    tail = 0u;
                    closed = true;
                }
                if (lane % G == 0 && own) fill[my_owner] = closed ? ENC_BIN_CLOSED : (int)tail;
            }
            __syncthreads();
        }
        if constexpr (WANT_DX) {
            float dx[3] = {0.0f, 0.0f, 0.0f};
            if (any && !DBG_SKIP(2)) {
                if constexpr (EARLY_DX) dx_from_entries<F>(L, c, g, tv, dx);
                else { float tl[8][F]; gather_entries<F>(table, e, tl); dx_from_entries<F>(L, c, g, tl, dx); }
            }
            if constexpr (DXM == ENC_DX_RAYS) {
                if (wave_any && !DBG_SKIP(1)) ray_accumulate_dx(dxl, (uint32_t)__builtin_amdgcn_readfirstlane((int)ray_cur), p_cur.z, dx, lane, src.n_rays);
            } else if (live) {
#pragma unroll
                for (int d = 0; d < 3; ++d) st32<float>(dxplanes, (uint32_t)d * plane_bytes + m * 4u, dx[d]);
            }
        }
    }
    if (emit && own) {
        // the unfinished line of every open owner: the only partial-line write of a region
        if (!closed && tail > 0u) {
            const char* my_bin = stage + (uint32_t)my_owner * BIN;
            char* dst = level_regions + ((size_t)my_owner * bpg + chunk) * (size_t)region_bytes + gpos;
            const uint32_t sub = (uint32_t)(lane % G) * (128u / G);
#pragma unroll
            for (int q = 0; q < 32 / G; ++q) {
                const uint32_t off = sub + 4u * q;
                if (off < tail) *reinterpret_cast<uint32_t*>(dst + off) = *reinterpret_cast<const uint32_t*>(my_bin + off);
            }
        }
        if (lane % G == 0) sink.counts[region0 + (size_t)my_owner * region_step] = (int)((gpos + (closed ? 0u : tail)) / REC);
    }
}

// Frequency encoding has no table: backward is only the input gradient, one plane group for all features.
// One range reduction per (sin, cos) pair as in freq_forward_h16_kernel: d/dph of the pair's features sin(ph) and sin(rn(ph + pi/2))
// is cos(ph) and cos(ph + pi/2 + d) = -(sin(ph) + d cos(ph)); as 72 libm cosf calls a sample this kernel cost 236 us at 2.1 M samples
// beside the 128 x 2 MLP's 800.
__global__ void __launch_bounds__(ENC_BLOCK)
freq_backward_kernel(const LnrNetSpec spec, const PointSrc src, const float* __restrict__ dfeat, float* __restrict__ dxl, int64_t m_pad) {
    const int64_t M = live_points(src);
    const int nf = spec.n_frequencies;
    for (int64_t m = (int64_t)blockIdx.x * ENC_BLOCK + threadIdx.x; m < M; m += (int64_t)gridDim.x * ENC_BLOCK) {
        float x[3];
        load_unit_point(src, m, x);
        float dx[3];
#pragma unroll
        for (int dim = 0; dim < 3; ++dim) {
            float acc = 0.0f;
            for (int f = 0; f < nf; ++f) {
                const uint32_t k = (uint32_t)(2 * (dim * nf + f));
                const float d0 = dfeat[(size_t)k * (size_t)m_pad + (size_t)m], d1 = dfeat[(size_t)(k + 1u) * (size_t)m_pad + (size_t)m];   // (64-bit index)
                const float mult = __uint_as_float((uint32_t)(127 + f) << 23);      // 2^f
                const float ph = lnr_mul_rn(lnr_mul_rn(x[dim], mult), LNR_PI_F);
                float sn, cs;
                sincos_f32(ph, &sn, &cs);
                const float h = lnr_add_rn(ph, LNR_PI_2_F);
                const float bb = lnr_add_rn(h, -ph);
                const float e = lnr_add_rn(lnr_add_rn(ph, -lnr_add_rn(h, -bb)), lnr_add_rn(LNR_PI_2_F, -bb));
                const float dl = 4.371139000186243e-8f - e;
                acc += (d0 * cs - d1 * __builtin_fmaf(dl, cs, sn)) * (mult * LNR_PI_F);
            }
            dx[dim] = acc;
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) dxl[(size_t)d * m_pad + m] = dx[d];
    }
}

// d_pts[m] = 0.5 * sum over groups of dxl[group][:, m]     (x = (xyz+1)/2)
__global__ void __launch_bounds__(ENC_BLOCK)
sum_dx_planes_kernel(const float* __restrict__ dxl, int n_groups, int64_t m_pad, const PointSrc src, float* __restrict__ d_pts) {
    const int64_t M = live_points(src);
    for (int64_t m = (int64_t)blockIdx.x * ENC_BLOCK + threadIdx.x; m < M; m += (int64_t)gridDim.x * ENC_BLOCK) {
        float a[3] = {0.0f, 0.0f, 0.0f};
        for (int gI = 0; gI < n_groups; ++gI) {
#pragma unroll
            for (int d = 0; d < 3; ++d) a[d] += dxl[(size_t)(gI * 3 + d) * m_pad + m];
        }
        d_pts[3 * m] = 0.5f * a[0]; d_pts[3 * m + 1] = 0.5f * a[1]; d_pts[3 * m + 2] = 0.5f * a[2];
    }
}

// ------------------------------------------------------------------------------------------------ host launchers
int lnr_encode_forward(const LnrNetSpec* spec, const float* params, const PointSrc* src, int64_t cap_points, float* feat,
                       int64_t m_pad, bool half_planes, hipStream_t st) {
    const float* table = params + spec->n_mlp_params;
    const bool hash = spec->encoding == LNR_ENC_HASHGRID;
    const int n_groups = hash ? spec->n_levels : (spec->enc_dim + 3) / 4;
    int64_t bpg = (cap_points + ENC_BLOCK * 4 - 1) / (ENC_BLOCK * 4);      // ~4 samples per thread
    if (bpg < 1) bpg = 1;
    if (bpg > 2048) bpg = 2048;
    const dim3 grid((unsigned)(n_groups * bpg)), block(ENC_BLOCK);
    if (!hash) {
        if (half_planes) {
            int64_t b1 = (cap_points + ENC_BLOCK - 1) / ENC_BLOCK;         // one sample per thread, one block row per dimension
            if (b1 < 1) b1 = 1;
            if (b1 > 16384) b1 = 16384;
            hipLaunchKernelGGL(freq_forward_h16_kernel, dim3((unsigned)(3 * b1)), block, 0, st, *spec, *src, reinterpret_cast<uint32_t*>(feat), m_pad, (int)b1);
        } else {
            hipLaunchKernelGGL(freq_forward_kernel, grid, block, 0, st, *spec, *src, feat, m_pad, (int)bpg);
        }
        return LNR_OK;
    }
    uint32_t level_mask = spec->n_levels >= 32 ? 0xFFFFFFFFu : (1u << spec->n_levels) - 1u;
#ifdef LNR_ABLATE
    // development builds, read at every call (tools/probe_forward_levels.py walks them in one process): time a subset of the levels
    if (const char* v = getenv("LNR_X_FWD_LEVELS")) level_mask &= (uint32_t)strtol(v, nullptr, 0);
    if (level_mask == 0u) return LNR_OK;
#endif
    const dim3 mgrid((unsigned)(__builtin_popcount(level_mask) * bpg));
    const int xcd = lnr_xcd_affine(__builtin_popcount(level_mask), cap_points, true) ? 1 : 0;
#define LNR_FWD(F, H) hipLaunchKernelGGL((encode_forward_kernel<F, H>), mgrid, block, 0, st, *spec, table, *src, feat, m_pad, (int)bpg, level_mask, xcd)
    if (half_planes) {
        switch (spec->n_features) {
            case 2: LNR_FWD(2, true); break;
            case 4: LNR_FWD(4, true); break;
            case 8: LNR_FWD(8, true); break;
            default: lnr_set_error("fp16 feature planes pair up features: n_features_per_level must be even"); return LNR_ERR_UNSUPPORTED;
        }
    } else {
        switch (spec->n_features) {
            case 1: LNR_FWD(1, false); break;
            case 2: LNR_FWD(2, false); break;
            case 4: LNR_FWD(4, false); break;
            default: LNR_FWD(8, false); break;
        }
    }
#undef LNR_FWD
    return LNR_OK;
}

int lnr_encode_backward(const LnrNetSpec* spec, const float* params, const PointSrc* src, int64_t cap_points, const float* dfeat,
                        float* dxl, int64_t m_pad, float* grad_table, void* regions, const RegionPlan* plan, int* counts, int bpg,
                        int maxo, int shift, long long* ovf, int* ovf_flag, int epoch, float* d_pts, float* d_rays_acc, long long* ray_acc, bool bins_w8,
                        int parts, hipStream_t st) {
    const float* table = params + spec->n_mlp_params;
    const bool hash = spec->encoding == LNR_ENC_HASHGRID;
    // d/dx mode: d_rays_acc (rays form, n_samples % 64 == 0, checked by the caller) > d_pts (planes) > none
    const int dxm = d_rays_acc ? ENC_DX_RAYS : (d_pts ? ENC_DX_PLANES : ENC_DX_NONE);
    float* dx_out = d_rays_acc ? reinterpret_cast<float*>(ray_acc) : dxl;
    // parts (LNR_ENC_PART_*): the input gradient and the table-gradient partition are separate launches of a hash grid's backward
    // (LNR_SPLIT_DX), so that the caller can hand the input gradient on before the partition starts; the partition kernels then carry no d/dx
    const bool do_dx = (parts & LNR_ENC_PART_DX) != 0, do_part = (parts & LNR_ENC_PART_RECORDS) != 0;
#if LNR_SPLIT_DX
    const int dxm_part = hash ? ENC_DX_NONE : dxm;
#else
    const int dxm_part = dxm;
    LNR_REQUIRE(do_dx && do_part, "lnr_encode_backward: this build computes d/dx inside the partition kernels");
#endif
    // six sums + one non-finite word per ray, rounded up to 16 bytes (one fill kernel instead of an aligned part and a tail; the workspace has the room)
    if (do_dx && d_rays_acc && hipMemsetAsync(ray_acc, 0, ((((size_t)src->n_rays * 7) * sizeof(long long)) + 15) & ~(size_t)15, st) != hipSuccess) {
        lnr_set_error("lnr_density_backward: hipMemsetAsync failed");
        return LNR_ERR_LAUNCH;
    }
    int n_groups = 1;
#if LNR_SPLIT_DX
    if (hash && do_dx && dxm != ENC_DX_NONE) {
        uint32_t level_mask = spec->n_levels >= 32 ? 0xFFFFFFFFu : (1u << spec->n_levels) - 1u;
#ifdef LNR_ABLATE
        static const long dx_levels = getenv("LNR_X_LEVELS") ? strtol(getenv("LNR_X_LEVELS"), nullptr, 0) : -1;
        level_mask &= (uint32_t)dx_levels;
#endif
        int64_t b = (cap_points + ENC_BLOCK * 4 - 1) / (ENC_BLOCK * 4);          // ~8 samples per lane pair
        if (b < 1) b = 1;
        if (b > 2048) b = 2048;
        const dim3 dgrid((unsigned)(__builtin_popcount(level_mask) * b)), dblock(ENC_BLOCK);
        const int xcd = lnr_xcd_affine(__builtin_popcount(level_mask), cap_points, false) ? 1 : 0;
        LnrProfScope prof("encode_dx", st);
#define LNR_LAUNCH_DXP(F)                                                                                                                   \
        do {                                                                                                                               \
            if (dxm == ENC_DX_RAYS) hipLaunchKernelGGL((encode_dx_pair_kernel<F, ENC_DX_RAYS>), dgrid, dblock, 0, st, *spec, table, *src, dfeat, dx_out, m_pad, (int)b, level_mask, xcd); \
            else hipLaunchKernelGGL((encode_dx_pair_kernel<F, ENC_DX_PLANES>), dgrid, dblock, 0, st, *spec, table, *src, dfeat, dx_out, m_pad, (int)b, level_mask, xcd);               \
        } while (0)
        if (level_mask != 0u) switch (spec->n_features) {
            case 1: LNR_LAUNCH_DXP(1); break;
            case 2: LNR_LAUNCH_DXP(2); break;
            case 4: LNR_LAUNCH_DXP(4); break;
            default: LNR_LAUNCH_DXP(8); break;
        }
#undef LNR_LAUNCH_DXP
    }
#endif
    if (hash) n_groups = spec->n_levels;
    if (hash && do_part && (regions != nullptr || dxm_part != ENC_DX_NONE)) {
        LevelList rec_levels, xp_levels, brec_levels, bxp_levels;          // 8-byte record levels, x-pair record levels, and their binned forms: one launch each
        rec_levels.n = xp_levels.n = brec_levels.n = bxp_levels.n = 0;
        const bool allow_binned = regions != nullptr && maxo <= LNR_BIN_MAX_OWNERS;
        int ovf_total = 0;
        for (int l = 0; l < spec->n_levels; ++l) {
            const int nfl = (int)spec->level_size[l] * spec->n_features;
#ifdef LNR_ABLATE
            static const int lmask = getenv("LNR_X_LEVELS") ? (int)strtol(getenv("LNR_X_LEVELS"), nullptr, 0) : -1;
            if (!((lmask >> l) & 1)) { ovf_total += nfl; continue; }
#endif
            const bool is_xp = spec->n_features == 2 && plan->xp[l] != 0;
            const bool is_binned = allow_binned && plan->binned[l] != 0 && plan->bytes[l] != 0;
            LevelList& ll = is_xp ? (is_binned ? bxp_levels : xp_levels) : (is_binned ? brec_levels : rec_levels);
            ll.lv[ll.n] = l; ll.slab_off[ll.n] = ovf_total; ll.n++;
            ovf_total += nfl;
        }
        const dim3 block(ENC_BWD_BLOCK);
#define LNR_LAUNCH_DXM(KERNEL, F, XP, ...)                                                                                    \
        do {                                                                                                                  \
            hipError_t e_ = hipSuccess;                                                                                       \
            const void* fn_ = dxm_part == ENC_DX_RAYS ? reinterpret_cast<const void*>(KERNEL<F, ENC_DX_RAYS, XP>)                  \
                            : dxm_part == ENC_DX_PLANES ? reinterpret_cast<const void*>(KERNEL<F, ENC_DX_PLANES, XP>)              \
                                                   : reinterpret_cast<const void*>(KERNEL<F, ENC_DX_NONE, XP>);               \
            e_ = hipFuncSetAttribute(fn_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
            if (e_ != hipSuccess) { lnr_set_error("lnr_density_backward: hipFuncSetAttribute(%zu) failed", lds); return LNR_ERR_LAUNCH; } \
            if (dxm_part == ENC_DX_RAYS) hipLaunchKernelGGL((KERNEL<F, ENC_DX_RAYS, XP>), grid, block, lds, st, __VA_ARGS__);   \
            else if (dxm_part == ENC_DX_PLANES) hipLaunchKernelGGL((KERNEL<F, ENC_DX_PLANES, XP>), grid, block, lds, st, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<F, ENC_DX_NONE, XP>), grid, block, lds, st, __VA_ARGS__);                      \
        } while (0)
#define LNR_LAUNCH_F(KERNEL, ...)                                                   \
        switch (spec->n_features) {                                                 \
            case 1: LNR_LAUNCH_DXM(KERNEL, 1, false, __VA_ARGS__); break;           \
            case 2: LNR_LAUNCH_DXM(KERNEL, 2, false, __VA_ARGS__); break;           \
            case 4: LNR_LAUNCH_DXM(KERNEL, 4, false, __VA_ARGS__); break;           \
            default: LNR_LAUNCH_DXM(KERNEL, 8, false, __VA_ARGS__); break;          \
        }
        if (rec_levels.n + xp_levels.n + brec_levels.n + bxp_levels.n > 0) {
            EncSink sink;
            // (the overflow accumulators are all-zero here: lnr_density_backward keeps them so between calls, lnr_density.hip)
            sink.ovf_flag = ovf_flag; sink.epoch = epoch;
            sink.grad_table = grad_table; sink.ovf = ovf; sink.regions = regions; sink.plan = *plan; sink.counts = counts; sink.maxo = maxo; sink.shift = shift;
            sink.combine_scale_max = LNR_COMBINE_SCALE_MAX;
#ifdef LNR_ABLATE
            sink.dbg = getenv("LNR_X_DBG") ? atoi(getenv("LNR_X_DBG")) : 0;
#endif
            LnrProfScope prof("encode_backward", st);
            const int maxo4 = (maxo + 3) & ~3;
            size_t lds = (size_t)(2 * maxo4 + 4 * maxo) * sizeof(int) + (size_t)ENC_STAGE_RECORDS * (spec->n_features >= 2 ? 16 : 8);
            // development probe (-DLNR_DEV_PROBES builds only; profiles/r06_occupancy_probes.txt): unused LDS behind the staging buffer, so that ONE workgroup fits a CU
            // (two waves per SIMD instead of four) - does the pair wait for latencies more waves would cover, or for throughput?
#ifdef LNR_DEV_PROBES
            static const int lds_pad = getenv("LNR_ENC_BWD_LDS_PAD") ? atoi(getenv("LNR_ENC_BWD_LDS_PAD")) : 0;
#else
            const int lds_pad = 0;
#endif
            lds += (size_t)lds_pad;
            if (rec_levels.n > 0) {
                const dim3 grid((unsigned)(rec_levels.n * bpg));
                sink.xcd_affine = lnr_xcd_affine(rec_levels.n, cap_points, false) ? 1 : 0;
                LNR_LAUNCH_F(encode_backward_kernel, *spec, table, *src, dfeat, dx_out, m_pad, bpg, rec_levels, sink);
            }
            if (xp_levels.n > 0) {                                          // n_features == 2
                // development probe: a smaller staging buffer for the x-pair launch (4 records per sample + straddling cells), so that
                // three workgroups fit a CU when the kernel is built for six waves per SIMD (-DLNR_ENC_BWD_WAVES=6)
#ifdef LNR_DEV_PROBES
                static const int xp_stage = getenv("LNR_ENC_BWD_XP_STAGE") ? atoi(getenv("LNR_ENC_BWD_XP_STAGE")) : 0;
                if (xp_stage > 0) lds = (size_t)(2 * maxo4 + 4 * maxo) * sizeof(int) + (size_t)xp_stage * 16 + (size_t)lds_pad;
#endif
                const dim3 grid((unsigned)(xp_levels.n * bpg));
                sink.xcd_affine = lnr_xcd_affine(xp_levels.n, cap_points, false) ? 1 : 0;
                LNR_LAUNCH_DXM(encode_backward_kernel, 2, true, *spec, table, *src, dfeat, dx_out, m_pad, bpg, xp_levels, sink);
            }
            {
                // binned partition: NW = 4 (256-thread workgroups, 512-byte bins) or 8 (512 threads, 1 KB bins)
                const int nw = bins_w8 ? 8 : 4;
                const size_t lds_b = 64 * sizeof(int) + (size_t)LNR_BIN_MAX_OWNERS * (size_t)(128 * nw);
                const dim3 block_b(64 * nw);
#define LNR_LAUNCH_BINNED(F, XP, LIST)                                                                                              \
                do {                                                                                                                \
                    const dim3 grid_b((unsigned)((LIST).n * bpg));                                                                  \
                    sink.xcd_affine = lnr_xcd_affine((LIST).n, cap_points, false) ? 1 : 0;                                                             \
                    const void* fn_ = nullptr;                                                                                      \
                    if (nw == 8) fn_ = dxm_part == ENC_DX_RAYS ? (const void*)encode_backward_binned_kernel<F, ENC_DX_RAYS, XP, 8>       \
                                     : dxm_part == ENC_DX_PLANES ? (const void*)encode_backward_binned_kernel<F, ENC_DX_PLANES, XP, 8>   \
                                                            : (const void*)encode_backward_binned_kernel<F, ENC_DX_NONE, XP, 8>;    \
                    else fn_ = dxm_part == ENC_DX_RAYS ? (const void*)encode_backward_binned_kernel<F, ENC_DX_RAYS, XP, 4>               \
                             : dxm_part == ENC_DX_PLANES ? (const void*)encode_backward_binned_kernel<F, ENC_DX_PLANES, XP, 4>           \
                                                    : (const void*)encode_backward_binned_kernel<F, ENC_DX_NONE, XP, 4>;            \
                    if (hipFuncSetAttribute(fn_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b) != hipSuccess) {           \
                        lnr_set_error("lnr_density_backward: hipFuncSetAttribute(%zu) failed", lds_b); return LNR_ERR_LAUNCH;       \
                    }                                                                                                               \
                    void* args_[] = {(void*)spec, (void*)&table, (void*)src, (void*)&dfeat, (void*)&dx_out, (void*)&m_pad, (void*)&bpg, (void*)&(LIST), (void*)&sink}; \
                    if (hipLaunchKernel(fn_, grid_b, block_b, args_, lds_b, st) != hipSuccess) {                                    \
                        lnr_set_error("lnr_density_backward: launch of the binned partition failed"); return LNR_ERR_LAUNCH;         \
                    }                                                                                                               \
                } while (0)
                if (brec_levels.n > 0) {
                    switch (spec->n_features) {
                        case 2: LNR_LAUNCH_BINNED(2, false, brec_levels); break;
                        case 4: LNR_LAUNCH_BINNED(4, false, brec_levels); break;
                        default: LNR_LAUNCH_BINNED(8, false, brec_levels); break;
                    }
                }
                if (bxp_levels.n > 0) LNR_LAUNCH_BINNED(2, true, bxp_levels);
#undef LNR_LAUNCH_BINNED
            }
#ifdef LNR_PHASE_TIMING
            if (getenv("LNR_PHASE_TIMING")) {
                static const char* names[LNR_N_PHASES] = {"load inputs", "cell/entries/gather/weights", "A rank", "barrier 1", "B scan", "barrier 2",
                                                          "C stage", "barrier 3", "D copy-out", "dx", "-", "loop"};
                unsigned long long h[2 * LNR_N_PHASES];
                if (lnr_phase_fetch(HIP_SYMBOL(lnr_phase_cycles), h, 2 * LNR_N_PHASES, st)) {
                    lnr_phase_print("encode_backward 8-byte", names, h);
                    lnr_phase_print("encode_backward x-pair", names, h + LNR_N_PHASES);
                }
            }
#endif
        }
#undef LNR_LAUNCH_F
#undef LNR_LAUNCH_DXM
    } else if (!hash && do_dx && dxm != ENC_DX_NONE) {
        // frequency encoding: no table, one plane group; the per-ray mode is served through the planes by the caller
        int64_t blocks = (cap_points + ENC_BLOCK - 1) / ENC_BLOCK;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(freq_backward_kernel, dim3((unsigned)blocks), dim3(ENC_BLOCK), 0, st, *spec, *src, dfeat, dxl, m_pad);
    }
    if (!do_dx) return LNR_OK;
    if (dxm == ENC_DX_RAYS) {
        hipLaunchKernelGGL(ray_grad_apply_kernel, dim3((unsigned)((src->n_rays * 6 + ENC_BLOCK - 1) / ENC_BLOCK)), dim3(ENC_BLOCK), 0, st,
                           ray_acc, src->n_rays, src->n_rays_dev, d_rays_acc);
    }
    if (dxm == ENC_DX_PLANES) {
        int64_t blocks = (cap_points + ENC_BLOCK - 1) / ENC_BLOCK;
        if (blocks > 4096) blocks = 4096;
        LnrProfScope prof("sum_dx_planes", st);
        hipLaunchKernelGGL(sum_dx_planes_kernel, dim3((unsigned)blocks), dim3(ENC_BLOCK), 0, st, dxl, n_groups, m_pad, *src, d_pts);
    }
    return LNR_OK;
}
