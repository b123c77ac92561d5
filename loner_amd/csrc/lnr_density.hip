// Density network: host-side dispatch (C ABI) of the level-major pipeline
//   forward : encode (level-major, lnr_encode.hip) -> feature planes -> MLP (fp32 MFMA, lnr_density_impl.h)
//   backward: [encode] -> MLP backward (weight-gradient slabs + d_feature planes) -> encode backward (table-gradient
//             records + d/dx planes) -> table_grad_reduce2 (per-owner LDS reduction) -> slab reduce
// plus the MFMA layout self-test.
#include <stdlib.h>

#include "lnr_density_api.h"
#include <vector>
#include <cstdlib>
#include <cstdio>

#include <mutex>
#include <unordered_map>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LNR_ENC_BWD_MAX_BPG 2048

// ------------------------------------------------------------------------------------------------
// Host-side note per workspace (keyed by its address): which layout last used it, whether the 64-bit overflow accumulators of that layout
// are known to be all-zero, and the call counter ("epoch") the kernels stamp a level's status word with when they add to its accumulators.
// The accumulators are zeroed once per layout change instead of once per backward (59 MB at the default size), and the reduce reads and
// re-zeroes only the levels that were stamped in THIS call (table_grad_reduce2_kernel).  What breaks the note - another network or batch
// size on the same workspace (its planes and regions lie elsewhere), lnr_density_workspace_init, an error between the first and the last
// launch of a backward - marks the accumulators unknown, and the next backward clears them.  The caller's side of the contract
// (include/loner_hip.h): between two density calls the workspace belongs to the library.
// ------------------------------------------------------------------------------------------------
namespace {
struct WsNote { uint64_t layout; uint32_t epoch; bool ovf_zero; };
std::mutex g_ws_mutex;
std::unordered_map<const void*, WsNote> g_ws_notes;

uint64_t layout_signature(const LnrNetSpec* spec, int64_t cap) {
    uint64_t h = 1469598103934665603ull;                        // FNV-1a over the spec and the point capacity
    auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
    mix(spec, sizeof(LnrNetSpec));
    mix(&cap, sizeof(cap));
    return h;
}
// a density call with this layout touches the workspace: a different layout than the note's leaves the accumulators unknown
// (bounded: a caller that never releases its workspaces cannot grow the map without limit - dropping every note is always safe, the
// next backward on each workspace then clears its accumulators once)
void ws_bound_locked() { if (g_ws_notes.size() > 64) g_ws_notes.clear(); }
void ws_touch(const void* ws, uint64_t layout) {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    ws_bound_locked();
    auto it = g_ws_notes.find(ws);
    if (it == g_ws_notes.end()) g_ws_notes[ws] = WsNote{layout, 0u, false};
    else if (it->second.layout != layout) { it->second.layout = layout; it->second.ovf_zero = false; }
}
// start of a backward that writes table gradients: -> {epoch of this call, must the accumulators be cleared first}; they count as unknown
// until ws_backward_done
void ws_backward_begin(const void* ws, uint64_t layout, int* epoch, bool* clear) {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    ws_bound_locked();
    WsNote& n = g_ws_notes[ws];
    if (n.layout != layout) { n.layout = layout; n.ovf_zero = false; }
    *clear = !n.ovf_zero;
    n.ovf_zero = false;
    n.epoch = n.epoch >= 0x7FFFFFF0u ? 1u : n.epoch + 1u;
    if (n.epoch == 1u) *clear = true;                          // (first call, or the counter wrapped: stale stamps must go too)
    *epoch = (int)n.epoch;
}
void ws_backward_done(const void* ws) {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    auto it = g_ws_notes.find(ws);
    if (it != g_ws_notes.end()) it->second.ovf_zero = true;
}
void ws_forget(const void* ws) {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    g_ws_notes.erase(ws);
}
}  // namespace

// grad[i] += sum over workgroup slabs.  A 64 x 16 workgroup: 64 consecutive parameters, the slabs split over 16 thread rows
// (a thousand threads share the reads instead of one per parameter), rows combined through LDS in a fixed order: no
// atomics, bit-reproducible.
#define LNR_SLAB_GROUPS 16
// the fold of 64 consecutive parameters by one 1024-thread workgroup (`block` = which 64): reduce_slabs_kernel, and the extra workgroups
// at the end of table_grad_reduce2_kernel's grid (one launch fewer per backward: ~4.5 us of a one-keyframe rank's 0.34 ms iteration)
__device__ __forceinline__ void slab_fold_block(const float* __restrict__ slabs, int n_slabs, int n_mlp, float* __restrict__ grad, int overwrite, int block) {
    __shared__ float part[LNR_SLAB_GROUPS][64];
    const int px = threadIdx.x & 63, gy = threadIdx.x >> 6;
    const int i = block * 64 + px;
    const int per = (n_slabs + LNR_SLAB_GROUPS - 1) / LNR_SLAB_GROUPS;
    int b = gy * per;
    const int b_end = min(n_slabs, b + per);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (i < n_mlp) {
        for (; b + 3 < b_end; b += 4) {
            s0 += slabs[(size_t)b * n_mlp + i]; s1 += slabs[(size_t)(b + 1) * n_mlp + i];
            s2 += slabs[(size_t)(b + 2) * n_mlp + i]; s3 += slabs[(size_t)(b + 3) * n_mlp + i];
        }
        for (; b < b_end; ++b) s0 += slabs[(size_t)b * n_mlp + i];
    }
    part[gy][px] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (gy == 0 && i < n_mlp) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < LNR_SLAB_GROUPS; ++k) s += part[k][px];
        grad[i] = overwrite ? s : grad[i] + s;
    }
}
__global__ void __launch_bounds__(64 * LNR_SLAB_GROUPS)
reduce_slabs_kernel(const float* __restrict__ slabs, int n_slabs, int n_mlp, float* __restrict__ grad, int overwrite) {
    slab_fold_block(slabs, n_slabs, n_mlp, grad, overwrite, (int)blockIdx.x);
}

// Workgroup `o` owns floats [o << shift, (o+1) << shift) of the table gradient.  The encode-backward workgroups of
// level l (bpg of them) wrote the records addressed to it into regions [l][o - first_owner(l)][chunk];
// it streams them, sums them in LDS in 64-bit fixed point (LDS float atomics run at < 1 lane/clk/CU on CDNA4, integer
// ones ~16x faster; 2^-42 resolution, exact and order-independent) and adds the slice to grad_table with coalesced
// read-modify-writes (it is the only writer of that slice).  PAIR: packed pair records (n_features >= 2) or {idx, v}
// records - see lnr_density_api.h.
//
// The kernel is bound by instruction issue and latency, not by bytes (ablations in docs/HISTORY.md 4.3): a region holds ~30-60 records,
// so every wave instruction serves one short region.  Hence the shape of the loop: a wave takes a contiguous run of its owner's
// regions, lane r holds region r's record count, and everything that depends only on the region - its count (v_readlane), its
// address (scalar arithmetic), the trip count - stays on the scalar unit; the first piece (64 x-pair records or 128 8-byte
// records, one load per lane) of RED_U regions is in flight while the previous RED_U are summed; the rare longer regions are
// finished by a second loop.  64 VGPRs, so that two workgroups (2 x 64 KB of accumulators) share a CU.
template <int PAIR>
__device__ __forceinline__ void reduce_one(long long* acc, uint2 r, uint32_t base) {
    if (PAIR) {
        uint32_t pi; float v0, v1;
        lnr_unpack_pair(r, pi, v0, v1);
        // no test for zero: a record exists because one of its values is non-zero, and a branch per value costs more than adding 0
        unsigned long long* a = reinterpret_cast<unsigned long long*>(acc) + 2 * pi;
        if (__builtin_expect(__builtin_fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1)) < 256.0f, 1)) {
            atomicAdd(a, (unsigned long long)lnr_to_fix_small(v0));
            atomicAdd(a + 1, (unsigned long long)lnr_to_fix_small(v1));
        } else {
            atomicAdd(a, (unsigned long long)lnr_to_fix(v0));
            atomicAdd(a + 1, (unsigned long long)lnr_to_fix(v1));
        }
    } else {
        const float v = __uint_as_float(r.y);
        if (v != 0.0f) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[r.x - base]), (unsigned long long)lnr_to_fix(v));
    }
}
__device__ __forceinline__ void reduce_xpair(long long* acc, const LnrXRec& r) {
    uint32_t pi, t; float a0, a1, fx;
    lnr_unpack_xpair(r, pi, t, a0, a1, fx);
    const uint32_t pj = pi ^ ((2u << t) - 1u);                        // the (x+1) corner's entry: e ^ (2^(t+1) - 1)
    long long q[4];
    lnr_xpair_fix(a0, a1, fx, q);
    unsigned long long* a = reinterpret_cast<unsigned long long*>(acc);
    atomicAdd(a + 2 * pi, (unsigned long long)q[0]);
    atomicAdd(a + 2 * pi + 1, (unsigned long long)q[1]);
    atomicAdd(a + 2 * pj, (unsigned long long)q[2]);
    atomicAdd(a + 2 * pj + 1, (unsigned long long)q[3]);
}

#ifdef LNR_PHASE_TIMING
__device__ unsigned long long lnr_reduce_phase_cycles[LNR_N_PHASES];
#endif
#define RED_SLICE (1 << LNR_SLICE_SHIFT)

// records are read once: streaming (nt) loads
typedef uint32_t red_u32x3 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t red_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ LnrXRec load_xrec_stream(const char* p) {
    const red_u32x3 v = __builtin_nontemporal_load(reinterpret_cast<const red_u32x3*>(p));
    LnrXRec r; r.a = v.x; r.b = v.y; r.c = v.z;
    return r;
}
__device__ __forceinline__ uint4 load_rec2_stream(const uint4* p) {
    const red_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const red_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// One wave sums the records of n_regions consecutive regions (of one level and owner) into the workgroup's LDS accumulators.
template <int PAIR, int RED_U>
__device__ __forceinline__ void reduce_regions(long long* acc, const int* __restrict__ wave_counts, const char* __restrict__ wave_regions, int n_regions,
                                               uint32_t region_bytes, bool xp, uint32_t base, int lane) {
    for (int r0 = 0; r0 < n_regions; r0 += 64) {
        const int nq = min(64, n_regions - r0);                                                      // regions of this round (wave-uniform)
        const int my_n = lane < nq ? wave_counts[r0 + lane] : 0;
        const char* round_regions = wave_regions + (size_t)r0 * region_bytes;
        if (xp) {
            // x-pair levels (lnr_density_api.h): 12-byte records, a piece = 64 records = one 12-byte load per lane; the first two pieces
            // of a region go through the pipelined loop (an encode-backward workgroup leaves ~65 records per region)
            LnrXRec cur[RED_U][2], nxt[RED_U][2];
            auto loadx = [&](int q0, LnrXRec out[RED_U][2]) {
#pragma unroll
                for (int u = 0; u < RED_U; ++u) {
                    const int q = q0 + u < nq ? q0 + u : nq - 1;                                     // clamp: unconditional loads
                    const int n = __builtin_amdgcn_readlane(my_n, q);
                    const char* rg = round_regions + (size_t)q * region_bytes;
                    out[u][0] = load_xrec_stream(rg + (lane < n ? lane : 0) * 12);
                    out[u][1] = load_xrec_stream(rg + (lane + 64 < n ? lane + 64 : 0) * 12);
                }
            };
            loadx(0, nxt);
            for (int q0 = 0; q0 < nq; q0 += RED_U) {
#pragma unroll
                for (int u = 0; u < RED_U; ++u) { cur[u][0] = nxt[u][0]; cur[u][1] = nxt[u][1]; }
                if (q0 + RED_U < nq) loadx(q0 + RED_U, nxt);
#pragma unroll
                for (int u = 0; u < RED_U; ++u) {
                    const int n = q0 + u < nq ? __builtin_amdgcn_readlane(my_n, q0 + u < 64 ? q0 + u : 63) : 0;
                    if (lane < n) reduce_xpair(acc, cur[u][0]);
                    if (lane + 64 < n) reduce_xpair(acc, cur[u][1]);
                }
            }
            unsigned long long more = __ballot(my_n > 128);                                          // regions with further pieces: rare
            while (more) {
                const int q = __builtin_ctzll(more);
                more &= more - 1ull;
                const int n = __builtin_amdgcn_readlane(my_n, q);
                const char* rg = round_regions + (size_t)q * region_bytes;
                for (int off = 128; off < n; off += 64 * 4) {                                        // four pieces in flight
                    LnrXRec r[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int k = off + 64 * u + lane; r[u] = load_xrec_stream(rg + (size_t)(k < n ? k : n - 1) * 12); }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (off + 64 * u + lane < n) reduce_xpair(acc, r[u]);
                }
            }
            continue;
        }
        // 8-byte records: a piece = 128 records = one 16-byte load (two records) per lane
        uint4 cur[RED_U], nxt[RED_U];
        auto load4 = [&](int q0, uint4 out[RED_U]) {
#pragma unroll
            for (int u = 0; u < RED_U; ++u) {
                const int q = q0 + u < nq ? q0 + u : nq - 1;
                const int n = __builtin_amdgcn_readlane(my_n, q);
                out[u] = load_rec2_stream(reinterpret_cast<const uint4*>(round_regions + (size_t)q * region_bytes) + (2 * lane < n ? lane : 0));   // regions are 16-byte aligned
            }
        };
        load4(0, nxt);
        for (int q0 = 0; q0 < nq; q0 += RED_U) {
#pragma unroll
            for (int u = 0; u < RED_U; ++u) cur[u] = nxt[u];
            if (q0 + RED_U < nq) load4(q0 + RED_U, nxt);
#pragma unroll
            for (int u = 0; u < RED_U; ++u) {
                const int n = q0 + u < nq ? __builtin_amdgcn_readlane(my_n, q0 + u < 64 ? q0 + u : 63) : 0;
                if (2 * lane < n) reduce_one<PAIR>(acc, make_uint2(cur[u].x, cur[u].y), base);
                if (2 * lane + 1 < n) reduce_one<PAIR>(acc, make_uint2(cur[u].z, cur[u].w), base);
            }
        }
        unsigned long long more = __ballot(my_n > 128);
        while (more) {
            const int q = __builtin_ctzll(more);
            more &= more - 1ull;
            const int n = __builtin_amdgcn_readlane(my_n, q);
            const uint2* rg = reinterpret_cast<const uint2*>(round_regions + (size_t)q * region_bytes);
            for (int off = 128; off < n; off += 128 * 4) {                                           // four pieces (of 128 records) in flight
                uint4 r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int k = off + 128 * u + 2 * lane; r[u] = load_rec2_stream(reinterpret_cast<const uint4*>(rg) + (k < n ? k : 0) / 2); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = off + 128 * u + 2 * lane;
                    if (k < n) reduce_one<PAIR>(acc, make_uint2(r[u].x, r[u].y), base);
                    if (k + 1 < n) reduce_one<PAIR>(acc, make_uint2(r[u].z, r[u].w), base);
                }
            }
        }
    }
}

// float range of level l, its record-level bookkeeping (shared by the two reduce kernels)
struct RedLevel { uint64_t lo, hi; };
__device__ __forceinline__ RedLevel red_level(const LnrNetSpec& spec, int l) {
    RedLevel r;
    r.lo = (uint64_t)spec.level_offset[l] * spec.n_features;
    r.hi = r.lo + (uint64_t)spec.level_size[l] * spec.n_features;
    return r;
}

// Spatially coherent levels (RegionPlan::split > 1: dense-indexed ones, where an owner slice is a slab of cells and a scan pours most of
// the level's records into two or three of them): `split` workgroups share an owner, each sums a run of the owner's regions in LDS and
// adds its non-zero sums to the level's 64-bit overflow accumulators (integer atomics: exact, order-independent), which the owner's
// workgroup of table_grad_reduce2_kernel folds in afterwards.  Without it those few owners were the whole kernel's critical path
// (0.8 ms for level 1 of the default network against 0.2 ms for everything else).
template <int PAIR, int RED_U, int MINW>
__global__ void __launch_bounds__(1024, MINW)
table_grad_reduce_split_kernel(const LnrNetSpec spec, const void* __restrict__ regions_v, const RegionPlan plan, const int* __restrict__ counts, int bpg,
                               int maxo, long long* __restrict__ ovf) {
    extern __shared__ long long acc[];
    constexpr int slice = RED_SLICE, shift = LNR_SLICE_SHIFT;
    int b = blockIdx.x, l = 0, span = 0;
    int64_t my_ovf = 0;
    RedLevel lv;
    for (;; ++l) {                                                          // the host launched exactly sum(span * split) workgroups
        lv = red_level(spec, l);
        span = (int)(((lv.hi - 1) >> shift) - (lv.lo >> shift)) + 1;
        const int n = plan.split[l] > 1 ? span * plan.split[l] : 0;
        if (b < n) break;
        b -= n;
        my_ovf += (int64_t)(lv.hi - lv.lo);
    }
    const int parts = plan.split[l], local = b / parts, part = b % parts;
    const uint32_t base = ((uint32_t)(lv.lo >> shift) + (uint32_t)local) << shift;
    const uint32_t region_bytes = plan.bytes[l];
    if (region_bytes == 0u) return;
#pragma unroll
    for (int i = threadIdx.x; i < slice; i += 1024) acc[i] = 0ll;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per_part = (bpg + parts - 1) / parts, part_first = part * per_part, part_n = min(per_part, bpg - part_first);
    const int per_wave = (part_n + 15) / 16, first = part_first + wave * per_wave, n_regions = min(per_wave, part_first + part_n - first);
    if (n_regions > 0)
        reduce_regions<PAIR, RED_U>(acc, counts + ((size_t)l * maxo + local) * bpg + first,
                                    reinterpret_cast<const char*>(regions_v) + plan.off[l] + ((size_t)local * bpg + first) * (size_t)region_bytes, n_regions,
                                    region_bytes, PAIR && plan.xp[l] != 0, base, lane);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < slice / 1024; ++k) {
        const int i = threadIdx.x + k * 1024;
        const uint64_t gi = (uint64_t)base + i;
        const long long q = acc[i];
        if (q != 0ll && gi >= lv.lo && gi < lv.hi) atomicAdd(reinterpret_cast<unsigned long long*>(ovf + my_ovf + (int64_t)(gi - lv.lo)), (unsigned long long)q);
    }
}

template <int PAIR, int RED_U, int MINW>
__global__ void __launch_bounds__(1024, MINW)   // HIP: (max threads, min waves per SIMD)
table_grad_reduce2_kernel(const LnrNetSpec spec, const void* __restrict__ regions_v, const RegionPlan plan, const int* __restrict__ counts, int bpg,
                          int maxo, long long* __restrict__ ovf, const int* __restrict__ ovf_flag, int epoch, float* __restrict__ grad_table,
                          int64_t n_table_floats, int overwrite, const float* __restrict__ slabs, int n_slabs, int n_mlp, float* __restrict__ grad_mlp) {
    extern __shared__ long long acc[];
    // slabs != NULL: the last ceil(n_mlp / 64) workgroups of the grid fold the MLP backward's weight-gradient slabs (slab_fold_block)
    const int slab_blocks = slabs != nullptr ? (n_mlp + 63) / 64 : 0;
    const int owner_blocks = (int)gridDim.x - slab_blocks;
    if ((int)blockIdx.x >= owner_blocks) {                                   // (workgroup-uniform)
        slab_fold_block(slabs, n_slabs, n_mlp, grad_mlp, overwrite, (int)blockIdx.x - owner_blocks);
        return;
    }
    // Owners in DESCENDING table order: the hardware starts workgroups in blockIdx order, the chip holds 512 of the ~900 at a time, and
    // the owners of the fine (x-pair) levels - the heaviest, and with the default network exactly 512 of them - sit at the END of the
    // table: started last they were the kernel's tail behind a half-empty chip; started first they fill it, and the lighter owners of
    // the coarse levels follow.
    const int o = owner_blocks - 1 - (int)blockIdx.x;
    constexpr int slice = RED_SLICE, shift = LNR_SLICE_SHIFT;
    const uint32_t base = (uint32_t)o << shift;
    PHASE_INIT();
#pragma unroll
    for (int i = threadIdx.x; i < slice; i += 1024) acc[i] = 0ll;
    __syncthreads();
    PHASE(0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int nwaves = 16;
    int64_t ovf_off = 0;                                                   // running offset of the record levels' overflow accumulators
    for (int l = 0; l < spec.n_levels; ++l) {
        const RedLevel lv = red_level(spec, l);
        const uint64_t lo = lv.lo, hi = lv.hi;
        const int64_t my_ovf = ovf_off;
        ovf_off += (int64_t)(hi - lo);                                      // every level has overflow accumulators
        if (hi <= base || lo >= (uint64_t)base + slice) continue;
        // records that did not fit their region were summed into 64-bit accumulators by the encode kernel (and, on split levels, all
        // records by the split kernel): same fixed point, so region records + overflow add up exactly, whatever the (arrival-order
        // dependent) split between the two was.  [r4] The accumulators are ALL ZERO between calls: a kernel that adds to a level's
        // stores the call's epoch in the level's status word, only then are they read here - and put back to zero by this, their only,
        // reader (no 59 MB memset per call, no 59 MB of reads for levels nothing overflowed on; split levels always carry sums)
        if (plan.split[l] > 1 || ovf_flag[l] == epoch) {               // workgroup-uniform
            long long v[slice / 1024];
#pragma unroll
            for (int k = 0; k < slice / 1024; ++k) {
                const uint64_t gi = (uint64_t)base + threadIdx.x + k * 1024;
                v[k] = (gi >= lo && gi < hi) ? ovf[my_ovf + (int64_t)(gi - lo)] : 0ll;
            }
#pragma unroll
            for (int k = 0; k < slice / 1024; ++k) {
                if (v[k] != 0ll) {
                    acc[threadIdx.x + k * 1024] += v[k];
                    ovf[my_ovf + (int64_t)((uint64_t)base + threadIdx.x + k * 1024 - lo)] = 0ll;
                }
            }
            __syncthreads();
            PHASE(1);
        }
        const int local = o - (int)(lo >> shift);
        if (local < 0 || local >= maxo) continue;
        const uint32_t region_bytes = plan.bytes[l];
        if (region_bytes == 0u || plan.split[l] > 1) continue;              // every record of this level overflowed / went through the split kernel
        const int per_wave = (bpg + nwaves - 1) / nwaves;
        const int first = wave * per_wave;
        const int n_regions = min(per_wave, bpg - first);
        if (n_regions > 0)
            reduce_regions<PAIR, RED_U>(acc, counts + ((size_t)l * maxo + local) * bpg + first,
                                        reinterpret_cast<const char*>(regions_v) + plan.off[l] + ((size_t)local * bpg + first) * (size_t)region_bytes, n_regions,
                                        region_bytes, PAIR && plan.xp[l] != 0, base, lane);
        PHASE(3);
    }
    __syncthreads();
    PHASE(5);
    {
        // LNR_BWD_OVERWRITE_GRAD (workgroup-uniform): the slice IS this call's gradient - every float is stored, none is read
        float g[slice / 1024];
#pragma unroll
        for (int k = 0; k < slice / 1024; ++k) {
            const int64_t gi = (int64_t)base + threadIdx.x + k * 1024;
            g[k] = (!overwrite && gi < n_table_floats) ? grad_table[gi] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < slice / 1024; ++k) {
            const int64_t gi = (int64_t)base + threadIdx.x + k * 1024;
            const long long q = acc[threadIdx.x + k * 1024];
            if (gi < n_table_floats && (overwrite || q != 0ll)) grad_table[gi] = g[k] + (float)((double)q * (1.0 / (double)LNR_FIX_SCALE));
        }
    }
    PHASE(6);
    PHASE_FLUSH(lnr_reduce_phase_cycles, 0);
}

// ------------------------------------------------------------------------------------------------
// workspace layout: [feat enc_dim x m_pad][dfeat enc_dim x m_pad][dxl groups x 3 x m_pad][slabs][counts][regions]
// ------------------------------------------------------------------------------------------------
struct Layout {
    int64_t m_pad;
    int n_groups, bpg, maxo, shift, nown, n_split;
    RegionPlan plan;
    size_t off_status, off_feat, off_wide, off_dfeat, off_dxl, off_dpts, off_rayacc, off_slabs, off_ovf, off_counts, off_regions, total;
};

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Region capacities (RegionPlan, lnr_density_api.h): level l sends n_points*8 corner updates (F/2 pair records each, 1 when
// F == 1; half as many x-pair records) to the `span` owners its table covers, from bpg workgroups.  Each level's regions hold
// that expectation + 25 % + 24 records; anything beyond it (skewed data) goes to the level's 64-bit overflow accumulators, so
// this is a performance knob only.
static Layout make_layout(const LnrNetSpec* spec, int64_t n_points) {
    Layout L;
    L.m_pad = (n_points + 63) / 64 * 64;
    if (L.m_pad < 64) L.m_pad = 64;
    // Plane skew: the MLP kernels read / write all 32 feature (d_feature) planes of the same samples at once, and the bench window's
    // planes are 2^21 samples = 8 MiB apart - a power-of-two stride puts the 32 concurrent streams on the same HBM channel / cache set
    // bits.  A plane stride that is a multiple of 64 KiB gets 17 x 256 bytes on top (LNR_PLANE_SKEW floats; 0 switches it off: A/B).
#ifndef LNR_PLANE_SKEW
#define LNR_PLANE_SKEW 1088
#endif
    if (LNR_PLANE_SKEW > 0 && (L.m_pad * (int64_t)sizeof(float)) % 65536 == 0) L.m_pad += LNR_PLANE_SKEW;
    const bool hash = spec->encoding == LNR_ENC_HASHGRID;
    L.n_groups = hash ? spec->n_levels : 1;
    L.shift = LNR_SLICE_SHIFT;
    const int64_t n_table = spec->n_params - spec->n_mlp_params;
    L.nown = (int)((n_table + (1 << L.shift) - 1) >> L.shift);
    int64_t bpg = (n_points + LNR_ENC_BWD_BLOCK * LNR_BATCHES_PER_WG - 1) / (LNR_ENC_BWD_BLOCK * LNR_BATCHES_PER_WG);   // batches per encode-backward workgroup
    if (bpg < 1) bpg = 1;
    if (bpg > LNR_ENC_BWD_MAX_BPG) bpg = LNR_ENC_BWD_MAX_BPG;
    bpg = (bpg + 3) & ~(int64_t)3;       // the wave-private partition runs 4 waves (= 4 chunks) per workgroup
    L.bpg = (int)bpg;
    L.maxo = 1;
    size_t ovf_total = 0;
    uint64_t region_total = 0;
    for (int l = 0; l < LNR_MAX_LEVELS; ++l) { L.plan.off[l] = 0; L.plan.bytes[l] = 0; L.plan.xp[l] = 0; L.plan.split[l] = 1; L.plan.binned[l] = 0; }
    L.n_split = 0;
    if (hash) {
        const int F = spec->n_features;
        for (int pass = 0; pass < 2; ++pass) {                              // pass 1 only if the plan exceeds the budget: scaled down
            if (pass == 1 && region_total <= LNR_REGION_BUDGET) break;
            const double shrink = pass == 0 ? 1.0 : (double)LNR_REGION_BUDGET / (double)region_total;
            region_total = 0; ovf_total = 0; L.maxo = 1; L.n_split = 0;
            for (int l = 0; l < spec->n_levels; ++l) {
                ovf_total += (size_t)spec->level_size[l] * F;
                const uint64_t lo = (uint64_t)spec->level_offset[l] * F, hi = lo + (uint64_t)spec->level_size[l] * F;
                const int span = (int)(((hi - 1) >> L.shift) - (lo >> L.shift)) + 1;
                if (span > L.maxo) L.maxo = span;
                // expected records per (owner, chunk): 8 corner updates per point (F/2 pair records each), spread over the level's owners
                const bool xp = lnr_level_can_use_xpairs(*spec, l, LNR_XPAIR_SCALE_MIN);
                L.plan.xp[l] = xp ? 1 : 0;
                double r = (double)n_points * 8.0 * (F >= 2 ? F / 2.0 : 1.0) / (double)span / (double)L.bpg;
                if (xp) r *= 0.5;                                           // one record per x-neighbour pair
                else if (spec->level_scale[l] < LNR_COMBINE_SCALE_MAX) r *= LNR_COMBINE_FILL;   // run-length combined along the rays
                // twice the expectation + 64 records, in 256-byte units: a record beyond the capacity costs four (two) 64-bit global atomics,
                // so the capacity is generous (sweep in DESIGN.md 2; the reduce does not care how full a region is)
                const double recs = (r * (xp ? LNR_REGION_HEADROOM_XP : LNR_REGION_HEADROOM) + LNR_REGION_SLACK) * shrink;
                uint64_t bytes = (uint64_t)(recs * (xp ? 12.0 : 8.0));
                // Binned partition (lnr_encode.hip): hashed levels, whose records spread evenly over the owners - a batch (one sample per
                // thread) must be expected to fill at most half a bin, worst case (nothing dead, nothing combined)
                const double bin_fill = (double)LNR_ENC_BWD_BLOCK * 8.0 * (xp ? 0.5 * 12.0 : 8.0) / (double)span;
                const bool binned = F >= 2 && spec->level_hashed[l] != 0 && span <= LNR_BIN_MAX_OWNERS && bin_fill <= 0.5 * LNR_BIN_BYTES + 8.0 &&
                                    LNR_ENC_BWD_BLOCK == 512 && shrink == 1.0;
                L.plan.binned[l] = binned ? 1 : 0;
                if (binned) bytes += LNR_BIN_BYTES + 128;
                bytes = (bytes + 255u) & ~(uint64_t)255u;
                // spatially coherent (dense-indexed) levels: several reduce workgroups per owner (table_grad_reduce_split_kernel)
                int parts = spec->level_hashed[l] == 0 ? LNR_REDUCE_SPLIT : 1;
                while (parts > 1 && L.bpg / parts < 16) parts >>= 1;      // at least one region per wave
                L.plan.split[l] = (uint8_t)(parts < 1 ? 1 : parts);
                if (parts > 1) L.n_split += span * parts;
                L.plan.off[l] = region_total;
                L.plan.bytes[l] = (uint32_t)bytes;
                region_total += (uint64_t)span * (uint64_t)L.bpg * bytes;
            }
        }
    }
    const int64_t blocks = (int64_t)L.n_groups * L.bpg;
    size_t off = 0;
    L.off_status = off; off += LNR_WORKSPACE_STATUS_BYTES;          // status words (include/loner_hip.h), same place for every n_points
    // (fp16 mode + frequency encoding: the MLP kernels evaluate the encoding and its input gradient themselves - no planes at all)
    const size_t plane_set = lnr_f16_fused_freq(spec) ? 0 : (size_t)spec->enc_dim * L.m_pad * sizeof(float);
    L.off_feat = off; off += align256(plane_set);
    L.off_wide = off; off += align256(lnr_wide_workspace(spec, n_points));     // chunk planes of the 256 x 2..3 route (0 bytes for every other network); the forward uses them too
    L.off_dfeat = off; off += align256(plane_set);
    L.off_dxl = off; off += align256(plane_set == 0 ? 0 : (size_t)L.n_groups * 3 * L.m_pad * sizeof(float));
    L.off_dpts = off; off += align256((size_t)3 * L.m_pad * sizeof(float));      // d_pts scratch of the general d_rays route
    L.off_rayacc = off; off += align256(((size_t)(L.m_pad / 64) * 7 + 2) * sizeof(long long));   // per-ray sums of the d_rays route (n_samples >= 64 there) + a non-finite word per ray
    L.off_slabs = off; off += align256((size_t)LNR_BWD_MAX_BLOCKS * spec->n_mlp_params * sizeof(float));
    L.off_ovf = off; off += align256(ovf_total * sizeof(long long));
    L.off_counts = off; off += align256(hash ? (size_t)blocks * L.maxo * sizeof(int) : 0);
    L.off_regions = off; off += (size_t)region_total;
    L.total = off;
    return L;
}

// flag LNR_BWD_REPORT_REGIONS: how full the record regions ran (diagnostic for sizing RegionPlan; synchronises the stream)
static void report_regions(const LnrNetSpec* spec, const Layout& L, const RegionPlan& plan, const int* counts, const float* dfeat, int64_t n_points, hipStream_t st) {
    std::vector<int> h((size_t)spec->n_levels * L.maxo * L.bpg);
    if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(h.data(), counts, h.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return;
    for (int l = 0; l < spec->n_levels; ++l) {
        const int rec = plan.xp[l] ? 12 : 8, cap = (int)(plan.bytes[l] / rec);
        long long sum = 0, over = 0; int mx = 0;
        const uint64_t lo = (uint64_t)spec->level_offset[l] * spec->n_features, hi = lo + (uint64_t)spec->level_size[l] * spec->n_features;
        const int span = (int)(((hi - 1) >> L.shift) - (lo >> L.shift)) + 1;
        for (int o = 0; o < span; ++o)
            for (int c = 0; c < L.bpg; ++c) {
                const int v = h[((size_t)l * L.maxo + o) * L.bpg + c];
                sum += v; if (v > mx) mx = v; if (v > cap) over += v - cap;
            }
        if (dfeat != nullptr && n_points > 0) {                 // how many samples carry a gradient into this level at all
            std::vector<float> plane((size_t)n_points);
            if (hipMemcpy(plane.data(), dfeat + (size_t)l * spec->n_features * L.m_pad, plane.size() * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess) {
                long long nz = 0, runs = 0;
                for (int64_t i = 0; i < n_points; ++i) { nz += plane[i] != 0.0f; runs += (plane[i] != 0.0f) && (i == 0 || plane[i - 1] == 0.0f); }
                fprintf(stderr, "[lnr regions] level %2d: %lld of %lld samples have d_feature != 0, in %lld runs\n", l, nz, (long long)n_points, runs);
            }
        }
        fprintf(stderr, "[lnr regions] level %2d: %d-B records, capacity %d, mean %.1f, max %d, overflowed %lld of %lld\n", l, rec, cap,
                (double)sum / ((double)span * L.bpg), mx, over, sum);
    }
}

static int check_spec(const LnrNetSpec* spec, const char* who) {
    LNR_REQUIRE(spec != nullptr, "%s: null spec", who);
    LNR_REQUIRE(spec->in_dim > 0 && spec->n_mlp_params > 0, "%s: spec not finalized (call lnr_net_spec_finalize)", who);
    const int ht = spec->n_neurons / 16;
    if (!(ht == 1 || ht == 2 || ht == 4 || ht == 8 || ht == 16) || spec->n_neurons % 16) {
        lnr_set_error("%s: n_neurons=%d not supported (need 16, 32, 64, 128 or 256)", who, spec->n_neurons);
        return LNR_ERR_UNSUPPORTED;
    }
    return LNR_OK;
}

// LNR_PREC_F16 is implemented for the reference's sigma network shape class; anything else must say so, not fall back
static int check_f16(const LnrNetSpec* spec, const char* who) {
    if (spec->precision != LNR_PREC_F16 || lnr_f16_supported(spec) || lnr_wide_class(spec)) return LNR_OK;
    lnr_set_error("%s: precision fp16 covers networks with an even number of encoded features per level, 16/32/64/128/256 neurons, "
                  "at most 3 hidden layers and - below 256 neurons - at most 128 (padded) inputs whose weights fit the LDS; use precision "
                  "fp32 for this network", who);
    return LNR_ERR_UNSUPPORTED;
}

static size_t fwd_lds(const LnrNetSpec* s, int w_lds) { return ((w_lds ? (size_t)lnr_w_lds_floats(s->n_neurons, s->in_dim, s->n_hidden, true) : 0) + 4) * sizeof(float); }
static size_t bwd_lds(const LnrNetSpec* s, int w_lds, int waves, int dw64) {
    const size_t H = s->n_neurons;
    const size_t scratch = H * 16 + (s->n_hidden > 1 ? (size_t)(s->n_hidden + 1) * H * 16 : 0);
    return ((w_lds ? 1 : 0) * (size_t)s->n_mlp_params + (dw64 ? 2 : 1) * (size_t)s->n_mlp_params + (size_t)waves * scratch) * sizeof(float);
}

// Launch shape of the MLP kernels.  The reference's default shape class (32 encoded features -> <= 64 ReLU neurons
// -> 1) takes the register-resident kernels; everything else prefers weights in LDS and 4 waves per workgroup and
// falls back to fewer waves, then to weights read from global memory (L2), until the workgroup fits the 160 KB LDS
// of a CDNA4 CU.
static int plan_launch(const LnrNetSpec* spec, int64_t n_points, bool backward, DensityPlan* plan, const char* who) {
    const int64_t tiles = (n_points + 15) / 16;
    if (lnr_wide_class(spec)) {          // 256 x 2..3: the layer-by-layer route plans its own launches (lnr_density_wide.hip)
        plan->fast32 = 0; plan->w_lds = 0; plan->waves = 4; plan->dw64 = 0; plan->regs = 0; plan->lds = 0; plan->grid = 1; plan->n_slabs = 1;
        return LNR_OK;
    }
    // (the register-resident kernels address up to 32 padded planes with 32-bit byte offsets: m_pad < 2^25, see lnr_bf3_class)
    if (spec->activation == LNR_ACT_RELU && spec->n_hidden == 1 && spec->in_dim == 32 && spec->enc_dim == 32 && spec->n_neurons <= 64 &&
        n_points <= (1ll << 25) - 2048) {
        plan->fast32 = 1; plan->w_lds = 1; plan->waves = 4; plan->dw64 = 0; plan->regs = 0;
        plan->lds = backward ? (2 * (size_t)spec->n_mlp_params + 4 * (size_t)spec->n_neurons * 20) * sizeof(float)
                             : (size_t)spec->n_mlp_params * sizeof(float);
        int64_t blocks = (tiles + 3) / 4;
        const int64_t max_blocks = backward ? LNR_BWD_MAX_BLOCKS : LNR_DENSITY_MAX_BLOCKS;
        plan->grid = (int)(blocks > max_blocks ? max_blocks : (blocks < 1 ? 1 : blocks));
        plan->n_slabs = plan->grid;
        return LNR_OK;
    }
    plan->fast32 = 0; plan->dw64 = 0; plan->regs = 0;
    // backward of a general network: the register-accumulating kernel (lnr_density_regs.h) for up to three hidden layers and 128 padded
    // inputs (256 neurons: one hidden layer) - four waves, their dZ / layer-input images in LDS, the weights too when they fit
    // (from 64 neurons up: below that the whole gradient is small enough for the LDS-accumulating kernel to win - 32 Tanh x 2: 1.5 against 2.7 ms)
    if (backward && spec->n_neurons >= 64 && spec->n_hidden <= 3 && spec->in_dim <= 128 && !(spec->n_neurons == 256 && spec->n_hidden > 1)) {
        const size_t scratch = (size_t)4 * (spec->n_hidden > 1 ? 2 : 1) * spec->n_neurons * 20 * sizeof(float);       // LNR_TDZ_STRIDE rows
        const size_t weights = (size_t)lnr_w_lds_floats(spec->n_neurons, spec->in_dim, spec->n_hidden, true) * sizeof(float);       // (padded rows)
        if (scratch <= (size_t)LNR_LDS_LIMIT) {
            plan->regs = 1; plan->waves = 4;
            // weights in LDS: all of them, else the hidden matrices + output rows (read twice per step), else none
            const size_t hidden = (size_t)lnr_w_lds_floats(spec->n_neurons, spec->in_dim, spec->n_hidden, false) * sizeof(float);
            plan->w_lds = scratch + weights <= (size_t)LNR_LDS_LIMIT ? 1 : (spec->n_hidden > 1 && scratch + hidden <= (size_t)LNR_LDS_LIMIT ? 2 : 0);
            plan->lds = scratch + (plan->w_lds == 1 ? weights : (plan->w_lds == 2 ? hidden : 0));
            int64_t blocks = (tiles + 3) / 4;
            if (blocks > 256) blocks = 256;                          // one workgroup per CU (one wave per SIMD)
            plan->grid = (int)(blocks < 1 ? 1 : blocks);
            plan->n_slabs = plan->grid;
            return LNR_OK;
        }
    }
    // {weights in LDS, waves, 64-bit fixed-point weight-gradient accumulators}.  LDS-resident weights matter most (without
    // them every MFMA operand is a global load), then the integer accumulators (LDS float atomics are ~16x slower), then waves.
    static const int opts[12][3] = {{1, 4, 1}, {1, 2, 1}, {1, 1, 1}, {1, 4, 0}, {1, 2, 0}, {1, 1, 0},
                                    {0, 4, 1}, {0, 2, 1}, {0, 1, 1}, {0, 4, 0}, {0, 2, 0}, {0, 1, 0}};
    for (int o = 0; o < 12; ++o) {
        const int w_lds = opts[o][0], waves = opts[o][1], dw64 = opts[o][2];
        if (!backward && (waves != 4 || dw64)) continue;           // forward scratch does not depend on these
        const size_t lds = backward ? bwd_lds(spec, w_lds, waves, dw64) : fwd_lds(spec, w_lds);
        if (lds > (size_t)LNR_LDS_LIMIT) continue;
        plan->w_lds = w_lds; plan->waves = waves; plan->lds = lds; plan->dw64 = backward ? dw64 : 0;
        int64_t blocks = (tiles + waves - 1) / waves;
        const int64_t max_blocks = backward ? LNR_BWD_MAX_BLOCKS : LNR_DENSITY_MAX_BLOCKS;
        if (blocks > max_blocks) blocks = max_blocks;
        if (blocks < 1) blocks = 1;
        plan->grid = (int)blocks;
        plan->n_slabs = plan->grid;
        return LNR_OK;
    }
    lnr_set_error("%s: network (n_neurons=%d, n_hidden_layers=%d, in_dim=%d: %d MLP weights) does not fit the 160 KB LDS of a CU "
                  "even with one wave per workgroup; not supported by the fp32 kernels", who, spec->n_neurons, spec->n_hidden,
                  spec->in_dim, spec->n_mlp_params);
    return LNR_ERR_UNSUPPORTED;
}

// mlp_backward_regs_kernel: one translation unit per (width, depth)
static int mlp_bwd_regs(const LnrNetSpec* spec, const float* params, const float* feat, int64_t m_pad, const MlpPoints* pt, const float* d_sigma,
                        float* dfeat, float* slabs, int want_dfeat, const DensityPlan* plan, hipStream_t st) {
#define LNR_REGS_CASE(HT, NH) case (HT) * 4 + (NH): return lnr_mlp_bwd_regs_ht##HT##_nh##NH(spec, params, feat, m_pad, pt, d_sigma, dfeat, slabs, want_dfeat, plan, st)
    switch ((spec->n_neurons / 16) * 4 + spec->n_hidden) {
        LNR_REGS_CASE(4, 1); LNR_REGS_CASE(4, 2); LNR_REGS_CASE(4, 3);
        LNR_REGS_CASE(8, 1); LNR_REGS_CASE(8, 2); LNR_REGS_CASE(8, 3);
        LNR_REGS_CASE(16, 1);
        default: lnr_set_error("lnr_density_backward: no register-accumulating kernel for %d neurons x %d hidden layers", spec->n_neurons, spec->n_hidden); return LNR_ERR_UNSUPPORTED;
    }
#undef LNR_REGS_CASE
}

namespace {
struct ReduceCtx {
    const LnrNetSpec* spec;
    const void* regions; const int* counts; long long* ovf; const int* ovf_flag; int epoch; float* grad_table;
    RegionPlan plan; int bpg, maxo, shift; int64_t n_table; int n_split; int overwrite;
    const float* slabs; int n_slabs, n_mlp; float* grad_mlp;          // slabs != NULL: the weight-gradient fold rides in the reduce launch
};

template <int PAIR, int U, int W>
static int launch_reduce_variant(const ReduceCtx& c, int n_owners, hipStream_t st) {
    const size_t lds = (size_t)RED_SLICE * sizeof(long long);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(table_grad_reduce2_kernel<PAIR, U, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(table_grad_reduce_split_kernel<PAIR, U, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        lnr_set_error("lnr_density_backward: hipFuncSetAttribute failed"); return LNR_ERR_LAUNCH;
    }
    if (c.n_split > 0) {
        LnrProfScope prof("table_grad_reduce_split", st);
        hipLaunchKernelGGL((table_grad_reduce_split_kernel<PAIR, U, W>), dim3(c.n_split), dim3(1024), lds, st, *c.spec, c.regions, c.plan, c.counts, c.bpg, c.maxo,
                           c.ovf);
    }
    LnrProfScope prof("table_grad_reduce", st);
    const int slab_blocks = c.slabs != nullptr ? (c.n_mlp + 63) / 64 : 0;
    hipLaunchKernelGGL((table_grad_reduce2_kernel<PAIR, U, W>), dim3(n_owners + slab_blocks), dim3(1024), lds, st, *c.spec, c.regions, c.plan, c.counts, c.bpg, c.maxo,
                       c.ovf, c.ovf_flag, c.epoch, c.grad_table, c.n_table, c.overwrite, c.slabs, c.n_slabs, c.n_mlp, c.grad_mlp);
    return LNR_OK;
}
// 2 regions in flight per wave at 8 waves per SIMD (two workgroups per CU) measured best: 0.35 ms against 0.37 (4 in flight, 8 waves),
// 0.40 (4, 4 waves) and 0.43 (8, 4 waves) on the bench workload
#ifndef LNR_RED_U
#define LNR_RED_U 2
#endif
#ifndef LNR_RED_W
#define LNR_RED_W 8
#endif
int launch_table_reduce(const ReduceCtx& c, int n_owners, hipStream_t st) {
    if (c.spec->n_features < 2) return launch_reduce_variant<0, LNR_RED_U, LNR_RED_W>(c, n_owners, st);
    return launch_reduce_variant<1, LNR_RED_U, LNR_RED_W>(c, n_owners, st);
}

}  // namespace

static int make_src(PointSrc* s, MlpPoints* mp, const float* pts, int64_t n_points, const float* rays, const float* z,
                    int32_t n_rays, int32_t n_samples, const int32_t* n_rays_dev, const char* who) {
    if (pts != nullptr) {
        LNR_REQUIRE(n_points >= 0, "%s: negative n_points", who);
        *s = PointSrc{pts, nullptr, nullptr, 1, n_points, 0, nullptr};
        *mp = MlpPoints{n_points, nullptr, 0, 1, nullptr};
    } else {
        LNR_REQUIRE(rays != nullptr && z != nullptr, "%s: need either pts or (rays, z)", who);
        LNR_REQUIRE(n_rays >= 0 && n_samples > 0, "%s: bad n_rays/n_samples", who);
        *s = PointSrc{nullptr, rays, z, n_samples, 0, n_rays, n_rays_dev};
        *mp = MlpPoints{(int64_t)n_rays * n_samples, n_rays_dev, n_rays, n_samples, nullptr};
    }
    return LNR_OK;
}

extern "C" size_t lnr_density_workspace(const LnrNetSpec* spec, int64_t n_points) {
    if (!spec || n_points < 0) return 0;
    return make_layout(spec, n_points).total;
}

extern "C" size_t lnr_density_workspace_forward(const LnrNetSpec* spec, int64_t n_points) {
    if (!spec || n_points < 0) return 0;
    return make_layout(spec, n_points).off_dfeat;          // status words + feature planes: all the forward touches
}

extern "C" int lnr_density_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    LNR_REQUIRE(workspace != nullptr && workspace_bytes >= LNR_WORKSPACE_STATUS_BYTES, "lnr_density_workspace_init: workspace too small");
    ws_forget(workspace);
    if (hipMemsetAsync(workspace, 0, LNR_WORKSPACE_STATUS_BYTES, (hipStream_t)stream) != hipSuccess) {
        lnr_set_error("lnr_density_workspace_init: hipMemsetAsync failed");
        return LNR_ERR_LAUNCH;
    }
    return LNR_OK;
}

extern "C" int lnr_density_workspace_release(void* workspace) {
    LNR_REQUIRE(workspace != nullptr, "lnr_density_workspace_release: null workspace");
    ws_forget(workspace);
    return LNR_OK;
}

extern "C" int lnr_density_forward(const LnrNetSpec* spec, const float* params, const float* pts, int64_t n_points,
                                   const float* rays, const float* z, int32_t n_rays, int32_t n_samples,
                                   const int32_t* n_rays_dev, float* sigma, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_spec(spec, "lnr_density_forward");
    if (rc) return rc;
    if (n_points == 0 && (pts != nullptr || n_rays == 0)) return LNR_OK;          // empty batch: nothing to do, nothing to check
    PointSrc src; MlpPoints mp;
    rc = make_src(&src, &mp, pts, n_points, rays, z, n_rays, n_samples, n_rays_dev, "lnr_density_forward");
    if (rc) return rc;
    const int64_t cap = mp.n_points;
    if (cap == 0) return LNR_OK;
    LNR_REQUIRE(params && sigma && workspace, "lnr_density_forward: null params/sigma/workspace");
    const Layout L = make_layout(spec, cap);
    // (the limit is on the PADDED plane stride - the sample count rounded up to 64, plus the skew: ADVICE r5)
    LNR_REQUIRE(L.m_pad * (spec->n_features > 4 ? spec->n_features : 4) < (1ll << 30),
                "lnr_density_forward: too many points per call for 32-bit plane offsets (padded n_points * max(n_features, 4) must be < 2^30)");
    if (workspace_bytes < L.off_dfeat) {
        lnr_set_error("lnr_density_forward: workspace %zu < %zu (lnr_density_workspace_forward)", workspace_bytes, L.off_dfeat);
        return LNR_ERR_WORKSPACE;
    }
    const bool f16 = spec->precision == LNR_PREC_F16;
    rc = check_f16(spec, "lnr_density_forward");
    if (rc) return rc;
    DensityPlan plan;
    rc = plan_launch(spec, cap, false, &plan, "lnr_density_forward");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ws_touch(workspace, layout_signature(spec, cap));
    float* feat = (float*)((char*)workspace + L.off_feat);
    mp.clip_flag = reinterpret_cast<int32_t*>((char*)workspace + L.off_status) + LNR_STATUS_CLIPPED;
    const bool fused_freq = lnr_f16_fused_freq(spec);          // the MLP kernel evaluates the encoding itself: no encode launch, no planes
    if (!fused_freq) {
        LnrProfScope prof("encode_forward", st);
        rc = lnr_encode_forward(spec, params, &src, cap, feat, L.m_pad, f16, st);
    }
    if (rc) return rc;
    LNR_CHECK_LAUNCH("lnr_density_forward(encode)");
    LnrProfScope prof_mlp("mlp_forward", st);
    if (lnr_wide_class(spec)) {
        rc = lnr_mlp_fwd_wide(spec, params, feat, L.m_pad, &mp, sigma, (char*)workspace + L.off_wide, st);
        if (rc) return rc;
        LNR_CHECK_LAUNCH("lnr_density_forward(mlp 256 x n)");
        return LNR_OK;
    }
    if (f16) {
        rc = lnr_mlp_fwd_f16(spec, params, feat, L.m_pad, &mp, sigma, &src, st);
        if (rc) return rc;
        LNR_CHECK_LAUNCH("lnr_density_forward(mlp f16)");
        return LNR_OK;
    }
    if (lnr_bf3_class(spec, cap)) {
        rc = lnr_mlp_fwd_bf3(spec, params, feat, L.m_pad, &mp, sigma, st);
        if (rc) return rc;
        LNR_CHECK_LAUNCH("lnr_density_forward(mlp bf16x3)");
        return LNR_OK;
    }
    switch (spec->n_neurons / 16) {
        case 1: rc = lnr_mlp_fwd_ht1(spec, params, feat, L.m_pad, &mp, sigma, &plan, st); break;
        case 2: rc = lnr_mlp_fwd_ht2(spec, params, feat, L.m_pad, &mp, sigma, &plan, st); break;
        case 4: rc = lnr_mlp_fwd_ht4(spec, params, feat, L.m_pad, &mp, sigma, &plan, st); break;
        case 8: rc = lnr_mlp_fwd_ht8(spec, params, feat, L.m_pad, &mp, sigma, &plan, st); break;
        default: rc = lnr_mlp_fwd_ht16(spec, params, feat, L.m_pad, &mp, sigma, &plan, st); break;
    }
    if (rc) return rc;
    LNR_CHECK_LAUNCH("lnr_density_forward(mlp)");
    return LNR_OK;
}

extern "C" int lnr_density_backward(const LnrNetSpec* spec, const float* params, const float* pts, int64_t n_points,
                                    const float* rays, const float* z, int32_t n_rays, int32_t n_samples,
                                    const int32_t* n_rays_dev, const float* d_sigma, float* grad_params, float* d_pts, float* d_rays,
                                    int32_t reuse_features, int32_t flags, void* workspace, size_t workspace_bytes, void* input_grad_event,
                                    void* stream) {
    int rc = check_spec(spec, "lnr_density_backward");
    if (rc) return rc;
    // an empty batch has nothing to compute, but the input-gradient event is part of the contract of every successful call: a caller's
    // second stream waits for it, and an unrecorded event would let that stream run against the PREVIOUS call's record
    auto empty_batch = [&]() -> int {
        if (input_grad_event != nullptr && hipEventRecord((hipEvent_t)input_grad_event, (hipStream_t)stream) != hipSuccess) {
            lnr_set_error("lnr_density_backward: hipEventRecord failed");
            return LNR_ERR_LAUNCH;
        }
        return LNR_OK;
    };
    if (n_points == 0 && (pts != nullptr || n_rays == 0)) return empty_batch();
    PointSrc src; MlpPoints mp;
    rc = make_src(&src, &mp, pts, n_points, rays, z, n_rays, n_samples, n_rays_dev, "lnr_density_backward");
    if (rc) return rc;
    const int64_t cap = mp.n_points;
    if (cap == 0) return empty_batch();
    LNR_REQUIRE(params && d_sigma && workspace, "lnr_density_backward: null argument");
    LNR_REQUIRE(grad_params || d_pts || d_rays, "lnr_density_backward: nothing to compute (no grad_params, d_pts or d_rays)");
    const Layout L = make_layout(spec, cap);
    LNR_REQUIRE(L.m_pad * (spec->n_features > 4 ? spec->n_features : 4) < (1ll << 30),
                "lnr_density_backward: too many points per call for 32-bit plane offsets (padded n_points * max(n_features, 4) must be < 2^30)");
    if (workspace_bytes < L.total) {
        lnr_set_error("lnr_density_backward: workspace %zu < %zu (lnr_density_workspace)", workspace_bytes, L.total);
        return LNR_ERR_WORKSPACE;
    }
    DensityPlan plan;
    rc = plan_launch(spec, cap, true, &plan, "lnr_density_backward");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* feat = (float*)(ws + L.off_feat);
    float* dfeat = (float*)(ws + L.off_dfeat);
    float* dxl = (float*)(ws + L.off_dxl);
    float* slabs = (float*)(ws + L.off_slabs);
    long long* ovf = (long long*)(ws + L.off_ovf);
    int* counts = (int*)(ws + L.off_counts);
    void* regions = (void*)(ws + L.off_regions);
    const bool hash = spec->encoding == LNR_ENC_HASHGRID;
    // test hook: LNR_BWD_TABLE_ATOMICS sends every record down the global-atomic fallback path (same result, ~20x slower);
    // the tests use it as an independent implementation of the record partition
    RegionPlan rplan = L.plan;
    if (flags & LNR_BWD_TABLE_ATOMICS)                       // no room anywhere: every record takes the overflow path
        for (int l = 0; l < LNR_MAX_LEVELS; ++l) rplan.bytes[l] = 0;
    if (!(flags & LNR_BWD_BINS))
        for (int l = 0; l < LNR_MAX_LEVELS; ++l) rplan.binned[l] = 0;
    const bool f16 = spec->precision == LNR_PREC_F16;
    rc = check_f16(spec, "lnr_density_backward");
    if (rc) return rc;
    const bool want_grad = grad_params != nullptr;

    const bool fused_freq = lnr_f16_fused_freq(spec);          // encoding and its input gradient inside the MLP backward kernel
    if (!reuse_features && !fused_freq) {
        LnrProfScope prof("encode_forward", st);
        rc = lnr_encode_forward(spec, params, &src, cap, feat, L.m_pad, f16, st);
        if (rc) return rc;
        LNR_CHECK_LAUNCH("lnr_density_backward(encode)");
    }
    // d_rays (rays form only): the input gradient goes straight into the ray records.  When a wave's 64 samples lie on one
    // ray (n_samples % 64 == 0) and there is a table to walk, the encode kernels accumulate it themselves; otherwise it
    // takes the general route (d/dx planes -> d_pts scratch in the workspace -> lnr_points_grad_to_rays).
    LNR_REQUIRE(!(d_rays && pts), "lnr_density_backward: d_rays needs the rays form");
    LNR_REQUIRE(!(d_rays && d_pts), "lnr_density_backward: pass d_pts or d_rays, not both");
    const bool ray_accum = d_rays != nullptr && hash && (n_samples % 64 == 0);
    float* d_pts_eff = d_pts;
    if (d_rays && !ray_accum) d_pts_eff = (float*)(ws + L.off_dpts);
    const int want_dfeat = ((hash && want_grad) || d_pts_eff != nullptr || ray_accum) ? 1 : 0;
    int n_slabs = plan.n_slabs;
    {
    LnrProfScope prof("mlp_backward", st);
    if (lnr_wide_class(spec)) rc = lnr_mlp_bwd_wide(spec, params, feat, L.m_pad, &mp, d_sigma, dfeat, slabs, want_dfeat, want_grad ? 1 : 0, &n_slabs, ws + L.off_wide, st);
    else if (f16) rc = lnr_mlp_bwd_f16(spec, params, feat, L.m_pad, &mp, d_sigma, dfeat, slabs, want_dfeat, &n_slabs, &src, d_pts_eff, st);
    else if (lnr_bf3_class(spec, cap)) rc = lnr_mlp_bwd_bf3(spec, params, feat, L.m_pad, &mp, d_sigma, dfeat, slabs, want_dfeat, &n_slabs, st);
    else if (plan.regs) rc = mlp_bwd_regs(spec, params, feat, L.m_pad, &mp, d_sigma, dfeat, slabs, want_dfeat, &plan, st);
    else switch (spec->n_neurons / 16) {
        case 1: rc = lnr_mlp_bwd_ht1(spec, params, feat, L.m_pad, &mp, d_sigma, dfeat, slabs, want_dfeat, &plan, st); break;
        case 2: rc = lnr_mlp_bwd_ht2(spec, params, feat, L.m_pad, &mp, d_sigma, dfeat, slabs, want_dfeat, &plan, st); break;
        case 4: rc = lnr_mlp_bwd_ht4(spec, params, feat, L.m_pad, &mp, d_sigma, dfeat, slabs, want_dfeat, &plan, st); break;
        case 8: rc = lnr_mlp_bwd_ht8(spec, params, feat, L.m_pad, &mp, d_sigma, dfeat, slabs, want_dfeat, &plan, st); break;
        default: rc = lnr_mlp_bwd_ht16(spec, params, feat, L.m_pad, &mp, d_sigma, dfeat, slabs, want_dfeat, &plan, st); break;
    }
    }
    if (rc) return rc;
    LNR_CHECK_LAUNCH("lnr_density_backward(mlp)");
    float* grad_table = want_grad ? grad_params + spec->n_mlp_params : nullptr;
    // the overflow accumulators: all-zero between calls (WsNote above); cleared here only when that is not known
    int* ovf_flag = reinterpret_cast<int32_t*>(ws + L.off_status) + LNR_STATUS_OVF_LEVEL0;
    int epoch = 0;
    const bool uses_ovf = hash && want_grad;
    if (uses_ovf) {
        bool clear = false;
        ws_backward_begin(workspace, layout_signature(spec, cap), &epoch, &clear);
        if (clear && (hipMemsetAsync(ovf, 0, L.off_counts - L.off_ovf, st) != hipSuccess ||
                      hipMemsetAsync(ovf_flag, 0, LNR_MAX_LEVELS * sizeof(int32_t), st) != hipSuccess)) {
            lnr_set_error("lnr_density_backward: hipMemsetAsync failed");
            return LNR_ERR_LAUNCH;
        }
    } else {
        ws_touch(workspace, layout_signature(spec, cap));
    }
    // the weight-gradient fold rides in the table reduce's launch whenever both run in this call
    const bool fold_in_reduce = want_grad && hash && L.nown > 0 && !(flags & LNR_BWD_DEFER_WEIGHT_FOLD);
    ReduceCtx rctx{spec, regions, counts, ovf, ovf_flag, epoch, grad_table, rplan, L.bpg, L.maxo, L.shift, spec->n_params - spec->n_mlp_params,
                   (flags & LNR_BWD_TABLE_ATOMICS) ? 0 : L.n_split, (flags & LNR_BWD_OVERWRITE_GRAD) ? 1 : 0,
                   fold_in_reduce ? slabs : nullptr, n_slabs, spec->n_mlp_params, grad_params};
    // hash grids (LNR_SPLIT_DX): the input gradient first, as launches of its own; the table-gradient partition follows behind the event
    const bool split = LNR_SPLIT_DX && hash;
    if (want_dfeat && !fused_freq) {
        rc = lnr_encode_backward(spec, params, &src, cap, dfeat, dxl, L.m_pad, grad_table, want_grad ? regions : nullptr, &rplan, counts,
                                 L.bpg, L.maxo, L.shift,
                                 ovf, ovf_flag, epoch, d_pts_eff, ray_accum ? d_rays : nullptr, (long long*)(ws + L.off_rayacc), (flags & LNR_BWD_BINS_W8) != 0,
                                 split ? LNR_ENC_PART_DX : (LNR_ENC_PART_DX | LNR_ENC_PART_RECORDS), st);
        if (rc) return rc;
        LNR_CHECK_LAUNCH("lnr_density_backward(encode backward)");
    }
    if (d_rays && !ray_accum) {
        rc = lnr_points_grad_to_rays(d_pts_eff, z, n_rays, n_rays_dev, n_samples, d_rays, stream);
        if (rc) return rc;
    }
    // the input gradient (d_pts / d_rays) is complete here; what follows only finishes grad_params.  A caller that continues with
    // the input gradient on another stream (pose gradient, pose step, the next batch's rays and samples) waits for this event
    // instead of for the whole call
    if (input_grad_event != nullptr && hipEventRecord((hipEvent_t)input_grad_event, st) != hipSuccess) {
        lnr_set_error("lnr_density_backward: hipEventRecord failed");
        return LNR_ERR_LAUNCH;
    }
    if (!want_grad) return LNR_OK;       // frozen parameters: no table reduce, no weight-gradient fold
    if (want_dfeat) {
        if (split) {
            rc = lnr_encode_backward(spec, params, &src, cap, dfeat, dxl, L.m_pad, grad_table, regions, &rplan, counts, L.bpg, L.maxo, L.shift,
                                     ovf, ovf_flag, epoch, nullptr, nullptr, nullptr, (flags & LNR_BWD_BINS_W8) != 0, LNR_ENC_PART_RECORDS, st);
            if (rc) return rc;
            LNR_CHECK_LAUNCH("lnr_density_backward(table-gradient partition)");
        }
        if (hash && (flags & LNR_BWD_REPORT_REGIONS)) {           // diagnostic, call-time flag: synchronises the stream
            int64_t live = cap;
            int32_t nr = 0;
            if (n_rays_dev && hipStreamSynchronize(st) == hipSuccess && hipMemcpy(&nr, n_rays_dev, sizeof(nr), hipMemcpyDeviceToHost) == hipSuccess) live = (int64_t)nr * n_samples;
            report_regions(spec, L, rplan, counts, dfeat, live, st);
        }
    }
    if (hash && L.nown > 0) {          // also with cap_rec == 0 (all-atomic test path): it folds in the overflow accumulators
        rc = launch_table_reduce(rctx, L.nown, st);
        if (rc) return rc;
        LNR_CHECK_LAUNCH("lnr_density_backward(table reduce)");
        ws_backward_done(workspace);                  // the reduce has put every accumulator it read back to zero
#ifdef LNR_PHASE_TIMING
        if (getenv("LNR_PHASE_TIMING")) {
            static const char* names[LNR_N_PHASES] = {"zero LDS", "overflow fold", "-", "records", "-", "final barrier", "write-out"};
            unsigned long long h[LNR_N_PHASES];
            if (lnr_phase_fetch(HIP_SYMBOL(lnr_reduce_phase_cycles), h, LNR_N_PHASES, st)) lnr_phase_print("table_grad_reduce", names, h);
        }
#endif
    }
    // (folding the slabs right after the MLP backward instead was measured: +15 us in front of the encode backward, nothing gained)
    if ((flags & LNR_BWD_DEFER_WEIGHT_FOLD) || fold_in_reduce) return LNR_OK;      // the caller folds them (lnr_density_fold_weight_grads) / done above
    const int n_mlp = spec->n_mlp_params;
    LnrProfScope prof_slabs("reduce_slabs", st);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(lnr_div_up(n_mlp, 64)), dim3(64 * LNR_SLAB_GROUPS), 0, st, slabs, n_slabs, n_mlp, grad_params,
                       (flags & LNR_BWD_OVERWRITE_GRAD) ? 1 : 0);
    LNR_CHECK_LAUNCH("lnr_density_backward(reduce)");
    return LNR_OK;
}

extern "C" int lnr_density_fold_weight_grads(const LnrNetSpec* spec, int64_t n_points, float* grad_params, void* workspace,
                                             size_t workspace_bytes, int32_t flags, void* stream) {
    int rc = check_spec(spec, "lnr_density_fold_weight_grads");
    if (rc) return rc;
    LNR_REQUIRE(grad_params && workspace && n_points > 0, "lnr_density_fold_weight_grads: bad argument");
    const Layout L = make_layout(spec, n_points);
    LNR_REQUIRE(workspace_bytes >= L.total, "lnr_density_fold_weight_grads: workspace too small");
    int n_slabs;
    if (lnr_wide_class(spec)) n_slabs = 1;
    else if (spec->precision == LNR_PREC_F16) n_slabs = lnr_f16_bwd_slabs(spec, n_points);
    else if (lnr_bf3_class(spec, n_points)) n_slabs = lnr_bf3_bwd_slabs(spec, n_points);
    else {
        DensityPlan plan;
        rc = plan_launch(spec, n_points, true, &plan, "lnr_density_fold_weight_grads");
        if (rc) return rc;
        n_slabs = plan.n_slabs;
    }
    const int n_mlp = spec->n_mlp_params;
    hipStream_t st = (hipStream_t)stream;
    LnrProfScope prof_slabs("reduce_slabs", st);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(lnr_div_up(n_mlp, 64)), dim3(64 * LNR_SLAB_GROUPS), 0, st,
                       (const float*)((const char*)workspace + L.off_slabs), n_slabs, n_mlp, grad_params, (flags & LNR_BWD_OVERWRITE_GRAD) ? 1 : 0);
    LNR_CHECK_LAUNCH("lnr_density_fold_weight_grads");
    return LNR_OK;
}

// ------------------------------------------------------------------------------------------------
// MFMA layout self-test: D = A(16x4) * B(4x16) with asymmetric operands, checked against the
// layout the kernels assume (A: lane->A[l&15][l>>4]; B: lane->B[l>>4][l&15]; D: reg r ->
// D[4*(l>>4)+r][l&15]).
// ------------------------------------------------------------------------------------------------
__global__ void selftest_mfma_kernel(float* out) {
    const int lane = threadIdx.x;
    const int i = lane & 15, k = lane >> 4;
    const float a = (float)(i * 7 + k * 3 + 1) * 0.25f;          // A[i][k]
    const float b = (float)((lane & 15) * 5 - k * 11 + 2) * 0.5f; // B[k][j], j = lane&15
    f32x4 d = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
    float err = 0.0f;
    const int j = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float ref = 0.0f;
        for (int kk = 0; kk < 4; ++kk) ref += ((float)(row * 7 + kk * 3 + 1) * 0.25f) * ((float)(j * 5 - kk * 11 + 2) * 0.5f);
        err = fmaxf(err, fabsf(ref - d[r]));
    }
    err = wave_max(err);
    if (lane == 0) out[0] = err;
}

extern "C" int lnr_selftest_mfma(float* out, void* stream) {
    LNR_REQUIRE(out != nullptr, "lnr_selftest_mfma: null out");
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    lnr_selftest_mfma_f16(out, (hipStream_t)stream);
    lnr_selftest_mfma_bf3(out, (hipStream_t)stream);
    LNR_CHECK_LAUNCH("lnr_selftest_mfma");
    return LNR_OK;
}
