// Density network: host-side dispatch (C ABI) of the level-major pipeline
//   forward : encode (level-major, lnr_encode.hip) -> feature planes -> MLP (fp32 MFMA, lnr_density_impl.h)
//   backward: [encode] -> MLP backward (weight-gradient slabs + d_feature planes) -> encode backward (table-gradient
//             records + d/dx planes) -> table_grad_reduce2 (per-owner LDS reduction) -> slab reduce
// plus the MFMA layout self-test.
#include <stdlib.h>

#include "lnr_density_api.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LNR_ENC_BWD_MAX_BPG 2048

// grad[i] += sum over workgroup slabs.  A 64 x 16 workgroup: 64 consecutive parameters, the slabs split over 16 thread rows
// (a thousand threads share the reads instead of one per parameter), rows combined through LDS in a fixed order: no
// atomics, bit-reproducible.
#define LNR_SLAB_GROUPS 16
__global__ void __launch_bounds__(64 * LNR_SLAB_GROUPS)
reduce_slabs_kernel(const float* __restrict__ slabs, int n_slabs, int n_mlp, float* __restrict__ grad) {
    __shared__ float part[LNR_SLAB_GROUPS][64];
    const int px = threadIdx.x & 63, gy = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + px;
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
        grad[i] += s;
    }
}

// Workgroup `o` owns floats [o << shift, (o+1) << shift) of the table gradient.  The encode-backward workgroups of
// level l (bpg of them) wrote the records addressed to it into regions [l][o - first_owner(l)][chunk];
// it streams them (4 x 16-byte loads = 8 records in flight per lane), sums them in LDS in 64-bit fixed point (LDS
// float atomics run at < 1 lane/clk/CU on CDNA4, integer ones ~16x faster; 2^-42 resolution, exact and
// order-independent) and adds the slice to grad_table with coalesced read-modify-writes (it is the only writer of
// that slice).  PAIR: packed pair records (n_features >= 2) or {idx, v} records - see lnr_density_api.h.
template <int PAIR>
__device__ __forceinline__ void reduce_one(long long* acc, uint2 r, uint32_t base) {
    if (PAIR) {
        uint32_t pi; float v0, v1;
        lnr_unpack_pair(r, pi, v0, v1);
        // no test for zero: a record exists because one of its values is non-zero, and a branch per value costs more than adding 0
        atomicAdd(reinterpret_cast<unsigned long long*>(&acc[2 * pi]), (unsigned long long)__float2ll_rn(v0 * LNR_FIX_SCALE));
        atomicAdd(reinterpret_cast<unsigned long long*>(&acc[2 * pi + 1]), (unsigned long long)__float2ll_rn(v1 * LNR_FIX_SCALE));
    } else {
        const float v = __uint_as_float(r.y);
        if (v != 0.0f) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[r.x - base]), (unsigned long long)__float2ll_rn(v * LNR_FIX_SCALE));
    }
}

#define RED_U 8      // 16-byte loads (= pieces of 128 records) in flight per lane
template <int PAIR>
__global__ void __launch_bounds__(1024)
table_grad_reduce2_kernel(const LnrNetSpec spec, const void* __restrict__ regions_v, const int* __restrict__ counts, int bpg, int maxo,
                          int cap, int shift, const long long* __restrict__ ovf, float* __restrict__ grad_table, int64_t n_table_floats) {
    extern __shared__ long long acc[];
    const int o = blockIdx.x;
    const int slice = 1 << shift;
    const uint32_t base = (uint32_t)o << shift;
    for (int i = threadIdx.x; i < slice; i += blockDim.x) acc[i] = 0ll;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int F = spec.n_features;
    int64_t ovf_off = 0;                                                   // running offset of the coherent levels' overflow accumulators
    for (int l = 0; l < spec.n_levels; ++l) {
        const uint64_t lo = (uint64_t)spec.level_offset[l] * F, hi = lo + (uint64_t)spec.level_size[l] * F;   // float range of the level
        const bool dense = hi - lo <= (uint64_t)LNR_DENSE_LEVEL_FLOATS;
        const bool coherent = !dense;                                       // every record level has overflow accumulators
        const int64_t my_ovf = ovf_off;
        if (coherent) ovf_off += (int64_t)(hi - lo);
        if (hi <= base || lo >= (uint64_t)base + slice) continue;
        if (dense) continue;                                                // dense level: arrives through the slabs
        if (coherent) {
            // records that did not fit their region were summed into 64-bit accumulators by the encode kernel: same fixed point,
            // so region records + overflow add up exactly, whatever the (arrival-order dependent) split between the two was
            for (int i = threadIdx.x; i < slice; i += blockDim.x) {
                const uint64_t gi = (uint64_t)base + i;
                if (gi >= lo && gi < hi) acc[i] += ovf[my_ovf + (int64_t)(gi - lo)];
            }
            __syncthreads();
        }
        const int local = o - (int)(lo >> shift);
        if (local < 0 || local >= maxo) continue;
        // The regions of (level, owner) lie back to back, one per encode-backward workgroup (chunk); a wave takes a
        // contiguous run of them.  Their record counts are fetched with one load (lane r holds region r's count) and cut
        // into pieces of 128 records (= one 16-byte load per lane); the wave walks the flattened piece list four pieces per
        // step and loads the next four before it sums the current ones, so the many short regions (one encode-backward
        // batch appends ~30 records per owner) never serialise on HBM latency.
        const int per_wave = (bpg + nwaves - 1) / nwaves;
        const int first = wave * per_wave;
        const int n_regions = min(per_wave, bpg - first);
        const size_t owner_regions = ((size_t)l * maxo + local) * bpg + first;
        const uint2* level_regions = reinterpret_cast<const uint2*>(regions_v) + owner_regions * cap;
        const size_t region_stride = (size_t)cap;                          // 8-byte units between consecutive regions of this wave
        // x-pair levels (lnr_density_api.h): 12-byte records, pieces of 64 records = one 12-byte load per lane
        const bool xp = PAIR && lnr_level_uses_xpairs(spec, l);
        const int psh = xp ? 6 : 7;                                         // log2 records per piece
        for (int r0 = 0; r0 < n_regions; r0 += 64) {
            const int my_r = r0 + lane;
            const int my_n = my_r < n_regions ? counts[owner_regions + my_r] : 0;
            const int my_chunks = (my_n + (1 << psh) - 1) >> psh;
            int incl = my_chunks;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
            const int excl = incl - my_chunks;
            const int total = __shfl(incl, 63, 64);
            if (total == 0) continue;
            // the region holding piece c = the last lane with pieces whose first piece is at or before c; -> (region, record offset, records left)
            auto locate = [&](int c, int& qq, int& off, int& n) {
                const unsigned long long starts = __ballot(excl <= c && my_chunks > 0);
                qq = __builtin_amdgcn_readfirstlane(starts ? 63 - __clzll((long long)starts) : 0);
                off = (c - __shfl(excl, qq, 64)) << psh;
                n = __shfl(my_n, qq, 64) - off;
            };
            if (xp) {
                LnrXRec rec[RED_U], nxt[RED_U];
                int rem[RED_U], nrem[RED_U];
                auto loadx = [&](int c0, LnrXRec out[RED_U], int left[RED_U]) {
#pragma unroll
                    for (int u = 0; u < RED_U; ++u) {
                        const int c = c0 + u < total ? c0 + u : total - 1;                   // clamp: unconditional loads
                        int qq, off, n;
                        locate(c, qq, off, n);
                        left[u] = c0 + u < total ? (n < 64 ? n : 64) : 0;
                        const char* rg = reinterpret_cast<const char*>(level_regions + (size_t)(r0 + qq) * region_stride) + (size_t)off * 12u;
                        out[u] = *reinterpret_cast<const LnrXRec*>(rg + (lane < left[u] ? lane : 0) * 12);
                    }
                };
                loadx(0, nxt, nrem);
                for (int c0 = 0; c0 < total; c0 += RED_U) {
#pragma unroll
                    for (int u = 0; u < RED_U; ++u) { rec[u] = nxt[u]; rem[u] = nrem[u]; }
                    if (c0 + RED_U < total) loadx(c0 + RED_U, nxt, nrem);
#pragma unroll
                    for (int u = 0; u < RED_U; ++u) {
                        if (lane < rem[u]) {
                            uint32_t pi, t; float a0, a1, fx;
                            lnr_unpack_xpair(rec[u], pi, t, a0, a1, fx);
                            const float gx = 1.0f - fx;
                            const uint32_t pj = pi ^ ((2u << t) - 1u);                        // the (x+1) corner's entry: e ^ (2^(t+1) - 1)
                            atomicAdd(reinterpret_cast<unsigned long long*>(&acc[2 * pi]), (unsigned long long)__float2ll_rn(gx * a0 * LNR_FIX_SCALE));
                            atomicAdd(reinterpret_cast<unsigned long long*>(&acc[2 * pi + 1]), (unsigned long long)__float2ll_rn(gx * a1 * LNR_FIX_SCALE));
                            atomicAdd(reinterpret_cast<unsigned long long*>(&acc[2 * pj]), (unsigned long long)__float2ll_rn(fx * a0 * LNR_FIX_SCALE));
                            atomicAdd(reinterpret_cast<unsigned long long*>(&acc[2 * pj + 1]), (unsigned long long)__float2ll_rn(fx * a1 * LNR_FIX_SCALE));
                        }
                    }
                }
                continue;
            }
            uint4 rec[RED_U], nxt[RED_U];
            int rem[RED_U], nrem[RED_U];              // records of the chunk (<= 128), wave-uniform
            auto load4 = [&](int c0, uint4 out[RED_U], int left[RED_U]) {
#pragma unroll
                for (int u = 0; u < RED_U; ++u) {
                    const int c = c0 + u < total ? c0 + u : total - 1;                       // clamp: unconditional loads
                    int qq, off, n;
                    locate(c, qq, off, n);
                    left[u] = c0 + u < total ? (n < 128 ? n : 128) : 0;
                    const uint4* rg = reinterpret_cast<const uint4*>(level_regions + (size_t)(r0 + qq) * region_stride + off);   // cap is even
                    out[u] = rg[2 * lane < left[u] ? lane : 0];
                }
            };
            load4(0, nxt, nrem);
            for (int c0 = 0; c0 < total; c0 += RED_U) {
#pragma unroll
                for (int u = 0; u < RED_U; ++u) { rec[u] = nxt[u]; rem[u] = nrem[u]; }
                if (c0 + RED_U < total) load4(c0 + RED_U, nxt, nrem);
#pragma unroll
                for (int u = 0; u < RED_U; ++u) {
                    if (2 * lane < rem[u]) reduce_one<PAIR>(acc, make_uint2(rec[u].x, rec[u].y), base);
                    if (2 * lane + 1 < rem[u]) reduce_one<PAIR>(acc, make_uint2(rec[u].z, rec[u].w), base);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < slice; i += blockDim.x) {
        const int64_t gi = (int64_t)base + i;
        const long long q = acc[i];
        if (gi < n_table_floats && q != 0ll) grad_table[gi] += (float)((double)q * (1.0 / (double)LNR_FIX_SCALE));
    }
}

// ------------------------------------------------------------------------------------------------
// workspace layout: [feat enc_dim x m_pad][dfeat enc_dim x m_pad][dxl groups x 3 x m_pad][slabs][counts][regions]
// ------------------------------------------------------------------------------------------------
struct Layout {
    int64_t m_pad;
    int n_groups, bpg, maxo, cap, shift, nown, rec_bytes;
    size_t off_feat, off_dfeat, off_dxl, off_dpts, off_rayacc, off_slabs, off_dense, off_ovf, off_counts, off_regions, total;
};

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Region capacity: level l sends n_points*8 corner updates (F/2 pair records each, 1 when F == 1) to the `span`
// owners its table covers, from bpg workgroups; run-length combined (coarse) levels emit far fewer.  The busiest
// level sets the capacity (+25 % + 128 slack); anything beyond it (skewed data) falls back to global atomics, so
// this is a performance knob only.
static Layout make_layout(const LnrNetSpec* spec, int64_t n_points) {
    Layout L;
    L.m_pad = (n_points + 63) / 64 * 64;
    if (L.m_pad < 64) L.m_pad = 64;
    const bool hash = spec->encoding == LNR_ENC_HASHGRID;
    L.n_groups = hash ? spec->n_levels : 1;
    L.shift = LNR_SLICE_SHIFT;
    const int64_t n_table = spec->n_params - spec->n_mlp_params;
    L.nown = (int)((n_table + (1 << L.shift) - 1) >> L.shift);
    L.rec_bytes = 8;
    int64_t bpg = (n_points + 256 * 4 - 1) / (256 * 4);        // ~4 batches of 256 samples per encode-backward workgroup
    if (bpg < 1) bpg = 1;
    if (bpg > LNR_ENC_BWD_MAX_BPG) bpg = LNR_ENC_BWD_MAX_BPG;
    bpg = (bpg + 3) & ~(int64_t)3;       // the wave-private partition runs 4 waves (= 4 chunks) per workgroup
    L.bpg = (int)bpg;
    L.maxo = 1;
    double per_region = 0.0;
    size_t dense_total = 0, ovf_total = 0;
    if (hash) {
        const int F = spec->n_features;
        for (int l = 0; l < spec->n_levels; ++l) {
            if (lnr_level_is_dense(spec, l)) { dense_total += (size_t)spec->level_size[l] * F; continue; }
            if (lnr_level_has_overflow_acc(spec, l)) ovf_total += (size_t)spec->level_size[l] * F;
            const uint64_t lo = (uint64_t)spec->level_offset[l] * F, hi = lo + (uint64_t)spec->level_size[l] * F;
            const int span = (int)(((hi - 1) >> L.shift) - (lo >> L.shift)) + 1;
            if (span > L.maxo) L.maxo = span;
            double r = (double)n_points * 8.0 * (F >= 2 ? F / 2.0 : 1.0) / (double)span / (double)L.bpg;
            if (spec->level_scale[l] < LNR_COMBINE_SCALE_MAX) r *= 0.25;
            if (r > per_region) per_region = r;
        }
    }
    const int64_t blocks = (int64_t)L.n_groups * L.bpg;
    int64_t cap = (int64_t)(per_region * 1.25) + 128;
    const int64_t budget_cap = (int64_t)(LNR_REGION_BUDGET / ((uint64_t)L.rec_bytes * (uint64_t)blocks * (uint64_t)L.maxo));
    if (cap > budget_cap) cap = budget_cap;
    if (cap < 64) cap = 64;
    cap = (cap + 1) & ~(int64_t)1;       // even: regions stay 16-byte aligned
    L.cap = hash ? (int)cap : 0;
    size_t off = 0;
    L.off_feat = off; off += align256((size_t)spec->enc_dim * L.m_pad * sizeof(float));
    L.off_dfeat = off; off += align256((size_t)spec->enc_dim * L.m_pad * sizeof(float));
    L.off_dxl = off; off += align256((size_t)L.n_groups * 3 * L.m_pad * sizeof(float));
    L.off_dpts = off; off += align256((size_t)3 * L.m_pad * sizeof(float));      // d_pts scratch of the general d_rays route
    L.off_rayacc = off; off += align256((size_t)(L.m_pad / 64) * 6 * sizeof(long long));   // per-ray sums of the d_rays route (n_samples >= 64 there)
    L.off_slabs = off; off += align256((size_t)LNR_BWD_MAX_BLOCKS * spec->n_mlp_params * sizeof(float));
    L.off_dense = off; off += align256(dense_total * (size_t)lnr_dense_bpg(L.bpg) * sizeof(float));
    L.off_ovf = off; off += align256(ovf_total * sizeof(long long));
    L.off_counts = off; off += align256(hash ? (size_t)blocks * L.maxo * sizeof(int) : 0);
    L.off_regions = off; off += hash ? (size_t)blocks * L.maxo * (size_t)L.cap * L.rec_bytes : 0;
    L.total = off;
    return L;
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
    if (spec->precision != LNR_PREC_F16 || lnr_f16_supported(spec)) return LNR_OK;
    lnr_set_error("%s: precision fp16 covers networks with an even number of encoded features per level, 16/32/64/128 neurons, at most 3 "
                  "hidden layers and at most 128 (padded) inputs whose weights fit the LDS; use precision fp32 for this network", who);
    return LNR_ERR_UNSUPPORTED;
}

static size_t fwd_lds(const LnrNetSpec* s, int w_lds) { return ((w_lds ? (size_t)s->n_mlp_params : 0) + 4) * sizeof(float); }
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
    // (the register-resident kernels address the planes with 32-bit byte offsets up to 17 planes: n_points <= 2^25)
    if (spec->activation == LNR_ACT_RELU && spec->n_hidden == 1 && spec->in_dim == 32 && spec->enc_dim == 32 && spec->n_neurons <= 64 &&
        n_points <= (1ll << 25)) {
        plan->fast32 = 1; plan->w_lds = 1; plan->waves = 4; plan->dw64 = 0;
        plan->lds = backward ? (2 * (size_t)spec->n_mlp_params + 4 * (size_t)spec->n_neurons * 20) * sizeof(float)
                             : (size_t)spec->n_mlp_params * sizeof(float);
        int64_t blocks = (tiles + 3) / 4;
        const int64_t max_blocks = backward ? LNR_BWD_MAX_BLOCKS : LNR_DENSITY_MAX_BLOCKS;
        plan->grid = (int)(blocks > max_blocks ? max_blocks : (blocks < 1 ? 1 : blocks));
        return LNR_OK;
    }
    plan->fast32 = 0; plan->dw64 = 0;
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
        return LNR_OK;
    }
    lnr_set_error("%s: network (n_neurons=%d, n_hidden_layers=%d, in_dim=%d: %d MLP weights) does not fit the 160 KB LDS of a CU "
                  "even with one wave per workgroup; not supported by the fp32 kernels", who, spec->n_neurons, spec->n_hidden,
                  spec->in_dim, spec->n_mlp_params);
    return LNR_ERR_UNSUPPORTED;
}

namespace {
struct ReduceCtx {
    const LnrNetSpec* spec;
    const void* regions; const int* counts; const long long* ovf; float* grad_table;
    int bpg, maxo, cap, shift; int64_t n_table;
};

int launch_table_reduce(const ReduceCtx& c, int n_owners, hipStream_t st) {
    const size_t lds = ((size_t)1 << c.shift) * sizeof(long long);
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(table_grad_reduce2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(table_grad_reduce2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e0 != hipSuccess || e1 != hipSuccess) { lnr_set_error("lnr_density_backward: hipFuncSetAttribute failed"); return LNR_ERR_LAUNCH; }
    LnrProfScope prof("table_grad_reduce", st);
    if (c.spec->n_features >= 2)
        hipLaunchKernelGGL(table_grad_reduce2_kernel<1>, dim3(n_owners), dim3(1024), lds, st, *c.spec, c.regions, c.counts, c.bpg, c.maxo, c.cap,
                           c.shift, c.ovf, c.grad_table, c.n_table);
    else
        hipLaunchKernelGGL(table_grad_reduce2_kernel<0>, dim3(n_owners), dim3(1024), lds, st, *c.spec, c.regions, c.counts, c.bpg, c.maxo, c.cap,
                           c.shift, c.ovf, c.grad_table, c.n_table);
    return LNR_OK;
}

}  // namespace

static int make_src(PointSrc* s, MlpPoints* mp, const float* pts, int64_t n_points, const float* rays, const float* z,
                    int32_t n_rays, int32_t n_samples, const int32_t* n_rays_dev, const char* who) {
    if (pts != nullptr) {
        LNR_REQUIRE(n_points >= 0, "%s: negative n_points", who);
        *s = PointSrc{pts, nullptr, nullptr, 1, n_points, 0, nullptr};
        *mp = MlpPoints{n_points, nullptr, 0, 1};
    } else {
        LNR_REQUIRE(rays != nullptr && z != nullptr, "%s: need either pts or (rays, z)", who);
        LNR_REQUIRE(n_rays >= 0 && n_samples > 0, "%s: bad n_rays/n_samples", who);
        *s = PointSrc{nullptr, rays, z, n_samples, 0, n_rays, n_rays_dev};
        *mp = MlpPoints{(int64_t)n_rays * n_samples, n_rays_dev, n_rays, n_samples};
    }
    return LNR_OK;
}

extern "C" size_t lnr_density_workspace(const LnrNetSpec* spec, int64_t n_points) {
    if (!spec || n_points < 0) return 0;
    return make_layout(spec, n_points).total;
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
    LNR_REQUIRE(cap * (spec->n_features > 4 ? spec->n_features : 4) < (1ll << 30),
                "lnr_density_forward: too many points per call for 32-bit plane offsets (n_points * max(n_features, 4) must be < 2^30)");
    const Layout L = make_layout(spec, cap);
    if (workspace_bytes < L.total) {
        lnr_set_error("lnr_density_forward: workspace %zu < %zu (lnr_density_workspace)", workspace_bytes, L.total);
        return LNR_ERR_WORKSPACE;
    }
    const bool f16 = spec->precision == LNR_PREC_F16;
    rc = check_f16(spec, "lnr_density_forward");
    if (rc) return rc;
    DensityPlan plan;
    rc = plan_launch(spec, cap, false, &plan, "lnr_density_forward");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* feat = (float*)((char*)workspace + L.off_feat);
    {
        LnrProfScope prof("encode_forward", st);
        rc = lnr_encode_forward(spec, params, &src, cap, feat, L.m_pad, f16, st);
    }
    if (rc) return rc;
    LNR_CHECK_LAUNCH("lnr_density_forward(encode)");
    LnrProfScope prof_mlp("mlp_forward", st);
    if (f16) {
        rc = lnr_mlp_fwd_f16(spec, params, feat, L.m_pad, &mp, sigma, st);
        if (rc) return rc;
        LNR_CHECK_LAUNCH("lnr_density_forward(mlp f16)");
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
                                    int32_t reuse_features, int32_t flags, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_spec(spec, "lnr_density_backward");
    if (rc) return rc;
    if (n_points == 0 && (pts != nullptr || n_rays == 0)) return LNR_OK;          // empty batch: nothing to do, nothing to check
    PointSrc src; MlpPoints mp;
    rc = make_src(&src, &mp, pts, n_points, rays, z, n_rays, n_samples, n_rays_dev, "lnr_density_backward");
    if (rc) return rc;
    const int64_t cap = mp.n_points;
    if (cap == 0) return LNR_OK;
    LNR_REQUIRE(params && d_sigma && workspace, "lnr_density_backward: null argument");
    LNR_REQUIRE(grad_params || d_pts || d_rays, "lnr_density_backward: nothing to compute (no grad_params, d_pts or d_rays)");
    LNR_REQUIRE(cap * (spec->n_features > 4 ? spec->n_features : 4) < (1ll << 30),
                "lnr_density_backward: too many points per call for 32-bit plane offsets (n_points * max(n_features, 4) must be < 2^30)");
    const Layout L = make_layout(spec, cap);
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
    float* dense_slabs = (float*)(ws + L.off_dense);
    long long* ovf = (long long*)(ws + L.off_ovf);
    int* counts = (int*)(ws + L.off_counts);
    void* regions = (void*)(ws + L.off_regions);
    const bool hash = spec->encoding == LNR_ENC_HASHGRID;
    // test hook: LNR_BWD_TABLE_ATOMICS sends every record down the global-atomic fallback path (same result, ~20x slower);
    // the tests use it as an independent implementation of the record partition
    const int cap_rec = (flags & LNR_BWD_TABLE_ATOMICS) ? 0 : L.cap;
    const bool f16 = spec->precision == LNR_PREC_F16;
    rc = check_f16(spec, "lnr_density_backward");
    if (rc) return rc;
    const bool want_grad = grad_params != nullptr;

    if (!reuse_features) {
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
    int n_slabs = plan.grid;
    {
    LnrProfScope prof("mlp_backward", st);
    if (f16) rc = lnr_mlp_bwd_f16(spec, params, feat, L.m_pad, &mp, d_sigma, dfeat, slabs, want_dfeat, &n_slabs, st);
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
    ReduceCtx rctx{spec, regions, counts, ovf, grad_table, L.bpg, L.maxo, cap_rec, L.shift, spec->n_params - spec->n_mlp_params};
    if (want_dfeat) {
        rc = lnr_encode_backward(spec, params, &src, cap, dfeat, dxl, L.m_pad, grad_table, want_grad ? regions : nullptr, counts,
                                 want_grad ? dense_slabs : nullptr, L.bpg, L.maxo, cap_rec, L.shift,
                                 ovf, d_pts_eff, ray_accum ? d_rays : nullptr, (long long*)(ws + L.off_rayacc), st);
        if (rc) return rc;
        LNR_CHECK_LAUNCH("lnr_density_backward(encode backward)");
    }
    if (d_rays && !ray_accum) {
        rc = lnr_points_grad_to_rays(d_pts_eff, z, n_rays, n_rays_dev, n_samples, d_rays, stream);
        if (rc) return rc;
    }
    if (!want_grad) return LNR_OK;       // frozen parameters: no table reduce, no weight-gradient fold
    if (hash && L.nown > 0) {          // also with cap_rec == 0 (all-atomic test path): it folds in the overflow accumulators
        rc = launch_table_reduce(rctx, L.nown, st);
        if (rc) return rc;
        LNR_CHECK_LAUNCH("lnr_density_backward(table reduce)");
    }
    const int n_mlp = spec->n_mlp_params;
    LnrProfScope prof_slabs("reduce_slabs", st);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(lnr_div_up(n_mlp, 64)), dim3(64 * LNR_SLAB_GROUPS), 0, st, slabs, n_slabs, n_mlp, grad_params);
    LNR_CHECK_LAUNCH("lnr_density_backward(reduce)");
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
    LNR_CHECK_LAUNCH("lnr_selftest_mfma");
    return LNR_OK;
}
