// LiDAR ray construction from the keyframe point buffer, compaction, and the backward that turns
// dL/drays into dL/d[R|t] per keyframe (gfx950).  Compile with -ffp-contract=off.
//
// Replaces  LidarRayDirections.build_lidar_rays  src/common/ray_utils.py:269-322
//           get_far_val                          src/common/ray_utils.py:31-60
//           KeyFrame.build_lidar_rays            src/mapping/keyframe.py:71-101
//           the vstack/cat over the window       src/mapping/optimizer.py:333-338
//           and their autograd backward (the pose-Jacobian tail of loss.backward()).
//
// The keyframe point buffer is SoA (directions [3,n], distances [n]); a gather of 512 random
// columns touches 4 x 512 separate 64-B lines whatever the layout, so one thread per candidate
// ray with four independent loads in flight is already the memory-optimal form.
#include "lnr_common.h"

#define LNR_MAX_SEG 64

struct SegTable {
    int32_t n;
    int32_t start[LNR_MAX_SEG + 1];
};
struct DirTable {
    const float* dirs[LNR_MAX_SEG];
    int64_t n_points[LNR_MAX_SEG];
};

// exit distance of the ray from the cube [-1,1]^3 (ray_utils.py:55-58, no_nan=True); also reports
// which axis/plane was selected so the backward can route the gradient.
__device__ __forceinline__ float cube_exit(const float o[3], const float d[3], int* axis_out, float* t_out, float* dd_out) {
    float best = 0.0f;
    int best_axis = 0;
    float best_t = 0.0f, best_dd = 1.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float dd = d[a] + 1e-15f;
        const float t_lo = (-1.0f - o[a]) / dd;
        const float t_hi = (1.0f - o[a]) / dd;
        const float c_lo = fmaxf(t_lo, 0.0f), c_hi = fmaxf(t_hi, 0.0f);
        const bool hi_sel = c_hi > c_lo;                 // torch.max(dim=0) keeps the first on ties
        const float c = hi_sel ? c_hi : c_lo;
        const float t_raw = hi_sel ? t_hi : t_lo;
        if (a == 0 || c < best) { best = c; best_axis = a; best_t = t_raw; best_dd = dd; }
    }
    if (axis_out) { *axis_out = best_axis; *t_out = best_t; *dd_out = best_dd; }
    return best;
}

__global__ void build_lidar_rays_kernel(const float* __restrict__ directions, const float* __restrict__ distances, int64_t n_points,
                                        const int64_t* __restrict__ index, int n_index, const float* __restrict__ T,
                                        float range_min, float range_max, float scale, float sx, float sy, float sz,
                                        float* __restrict__ rays, float* __restrict__ depths, uint8_t* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_index) return;
    const int64_t src = index[i];
    const float l0 = directions[src], l1 = directions[n_points + src], l2 = directions[2 * n_points + src];
    const float dist = distances[src];
    float o[3], d[3];
    o[0] = (T[3] + sx) / scale;
    o[1] = (T[7] + sy) / scale;
    o[2] = (T[11] + sz) / scale;
    float v[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) v[r] = T[4 * r] * l0 + T[4 * r + 1] * l1 + T[4 * r + 2] * l2;
    const float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = v[r] / nrm;
    const float near = range_min / scale;
    const float far_range = range_max / scale;
    const float far = fminf(far_range, cube_exit(o, d, nullptr, nullptr, nullptr));
    float* rec = rays + (size_t)i * LNR_RAY_STRIDE;
    rec[0] = o[0]; rec[1] = o[1]; rec[2] = o[2];
    rec[3] = d[0]; rec[4] = d[1]; rec[5] = d[2];
    rec[6] = -d[0]; rec[7] = -d[1]; rec[8] = -d[2];
    rec[9] = 0.0f; rec[10] = 0.0f;
    rec[11] = near; rec[12] = far;
    depths[i] = dist / scale;
    keep[i] = (far > near + 1.0f / scale) ? 1 : 0;     // only rays with more than 1 m inside the world cube
}

extern "C" int lnr_build_lidar_rays(const float* directions, const float* distances, int64_t n_points, const int64_t* index,
                                    int32_t n_index, const float* transform, float range_min, float range_max, float scale,
                                    const float* shift, float* rays, float* depths, uint8_t* keep, void* stream) {
    LNR_REQUIRE(directions && distances && index && transform && shift && rays && depths && keep, "lnr_build_lidar_rays: null argument");
    LNR_REQUIRE(n_points > 0 && n_index >= 0 && scale > 0.0f, "lnr_build_lidar_rays: bad sizes");
    if (n_index == 0) return LNR_OK;
    hipLaunchKernelGGL(build_lidar_rays_kernel, dim3(lnr_div_up(n_index, 256)), dim3(256), 0, (hipStream_t)stream, directions, distances,
                       n_points, index, n_index, transform, range_min, range_max, scale, shift[0], shift[1], shift[2], rays, depths, keep);
    LNR_CHECK_LAUNCH("lnr_build_lidar_rays");
    return LNR_OK;
}

// Whole-window variant: every candidate ray of every keyframe (lidar and sky segments) in ONE launch.
// index == NULL: the source index of each candidate is drawn here (counter-based generator keyed by
// (seed, segment, i)) - the torch.randint of optimizer.py:288/301 - and written to index_out.
struct WindowTable {
    int32_t n_seg;
    int32_t start[LNR_MAX_SEG + 1];     // candidate range of each segment
    int32_t pose[LNR_MAX_SEG];          // row of `transforms` used by the segment
    const float* dirs[LNR_MAX_SEG];
    const float* dist[LNR_MAX_SEG];     // NULL -> const_dist (sky rays: ray_range[1] + 1)
    float const_dist[LNR_MAX_SEG];
    int64_t n_points[LNR_MAX_SEG];
};

__global__ void build_window_rays_kernel(const WindowTable tab, const int64_t* __restrict__ index, int64_t* __restrict__ index_out,
                                         uint64_t seed, const float* __restrict__ transforms, float range_min, float range_max,
                                         float scale, float sx, float sy, float sz, float* __restrict__ rays,
                                         float* __restrict__ depths, uint8_t* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = tab.start[tab.n_seg];
    if (i >= total) return;
    int seg = 0;
    while (seg + 1 < tab.n_seg && i >= tab.start[seg + 1]) ++seg;
    const int64_t n_points = tab.n_points[seg];
    int64_t src;
    if (index) src = index[i];
    else {
        const float u = lnr_rand_uniform(seed, 0x44ull + (uint64_t)seg, (uint64_t)(i - tab.start[seg]) >> 2, (uint32_t)(i - tab.start[seg]) & 3u);
        src = (int64_t)(u * (float)n_points);
        if (src >= n_points) src = n_points - 1;
    }
    if (index_out) index_out[i] = src;
    const float* D = tab.dirs[seg];
    const float l0 = D[src], l1 = D[n_points + src], l2 = D[2 * n_points + src];
    const float dist = tab.dist[seg] ? tab.dist[seg][src] : tab.const_dist[seg];
    const float* T = transforms + 12 * tab.pose[seg];
    float o[3], d[3], v[3];
    o[0] = (T[3] + sx) / scale; o[1] = (T[7] + sy) / scale; o[2] = (T[11] + sz) / scale;
#pragma unroll
    for (int r = 0; r < 3; ++r) v[r] = T[4 * r] * l0 + T[4 * r + 1] * l1 + T[4 * r + 2] * l2;
    const float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = v[r] / nrm;
    const float near = range_min / scale;
    const float far = fminf(range_max / scale, cube_exit(o, d, nullptr, nullptr, nullptr));
    float* rec = rays + (size_t)i * LNR_RAY_STRIDE;
    rec[0] = o[0]; rec[1] = o[1]; rec[2] = o[2];
    rec[3] = d[0]; rec[4] = d[1]; rec[5] = d[2];
    rec[6] = -d[0]; rec[7] = -d[1]; rec[8] = -d[2];
    rec[9] = 0.0f; rec[10] = 0.0f; rec[11] = near; rec[12] = far;
    depths[i] = dist / scale;
    keep[i] = (far > near + 1.0f / scale) ? 1 : 0;
}

extern "C" int lnr_build_window_rays(const float* const* directions, const float* const* distances, const float* const_distance,
                                     const int64_t* n_points, const int32_t* seg_start, const int32_t* seg_pose, int32_t n_seg,
                                     const int64_t* index, int64_t* index_out, uint64_t seed, const float* transforms,
                                     float range_min, float range_max, float scale, const float* shift, float* rays, float* depths,
                                     uint8_t* keep, void* stream) {
    LNR_REQUIRE(directions && distances && const_distance && n_points && seg_start && seg_pose && transforms && shift && rays && depths && keep,
                "lnr_build_window_rays: null argument");
    LNR_REQUIRE(n_seg >= 1 && n_seg <= LNR_MAX_SEG, "lnr_build_window_rays: n_seg must be in [1,%d]", LNR_MAX_SEG);
    LNR_REQUIRE(index != nullptr || index_out != nullptr, "lnr_build_window_rays: need index or index_out");
    WindowTable tab;
    tab.n_seg = n_seg;
    for (int s = 0; s < n_seg; ++s) {
        tab.start[s] = seg_start[s]; tab.pose[s] = seg_pose[s]; tab.dirs[s] = directions[s]; tab.dist[s] = distances[s];
        tab.const_dist[s] = const_distance[s]; tab.n_points[s] = n_points[s];
        LNR_REQUIRE(directions[s] != nullptr && n_points[s] > 0, "lnr_build_window_rays: segment %d has no points", s);
    }
    tab.start[n_seg] = seg_start[n_seg];
    const int total = seg_start[n_seg];
    if (total <= 0) return LNR_OK;
    hipLaunchKernelGGL(build_window_rays_kernel, dim3(lnr_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, tab, index, index_out, seed,
                       transforms, range_min, range_max, scale, shift[0], shift[1], shift[2], rays, depths, keep);
    LNR_CHECK_LAUNCH("lnr_build_window_rays");
    return LNR_OK;
}

struct SegOrder { int n; int order[LNR_MAX_SEG]; };
// what lnr_compact_rays_front appends to the compaction (one launch instead of two: a one-keyframe rank's iteration is a chain of
// ~4.7 us launches around 0.3 ms of real kernels): the loss normalisers of lnr_count_opaque (counts != NULL) or the rank's front record
// of lnr_shard_front_pack (record != NULL)
struct CompactTail { int32_t* counts; float* record; int cap; SegOrder ord; };

// single-workgroup, order-preserving stream compaction (a window is at most a few thousand rays)
__global__ void __launch_bounds__(1024)
compact_rays_kernel(const float* __restrict__ rays_in, const float* __restrict__ depths_in, const uint8_t* __restrict__ keep,
                    const int64_t* __restrict__ src_in, int n_in, const SegTable seg, float* __restrict__ rays_out,
                    float* __restrict__ depths_out, int64_t* __restrict__ src_out, int32_t* __restrict__ out_seg_start,
                    int32_t* __restrict__ n_out, const CompactTail tail) {
    __shared__ int wave_tot[16];
    __shared__ int running;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) running = 0;
    __syncthreads();
    for (int base = 0; base < n_in; base += 1024) {
        const int i = base + tid;
        const bool k = (i < n_in) && keep[i];
        const unsigned long long b = __ballot(k);
        const int within = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(b);
        __syncthreads();
        int before = running;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        const int pos = before + within;
        if (i < n_in) {
            for (int s = 0; s <= seg.n; ++s) if (seg.start[s] == i) out_seg_start[s] = pos;
        }
        if (k) {
            const float* a = rays_in + (size_t)i * LNR_RAY_STRIDE;
            float* o = rays_out + (size_t)pos * LNR_RAY_STRIDE;
#pragma unroll
            for (int c = 0; c < LNR_RAY_STRIDE; ++c) o[c] = a[c];
            depths_out[pos] = depths_in[i];
            if (src_out) src_out[pos] = src_in[i];
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_tot[w]; running += t; }
        __syncthreads();
    }
    if (tid <= seg.n && seg.start[tid] >= n_in) out_seg_start[tid] = running;
    if (tid == 0) *n_out = running;
    if (tail.counts == nullptr && tail.record == nullptr) return;
    // ---- the tail reads what this workgroup has just written (compacted rays / depths / segment starts): visible after the barrier
    __syncthreads();
    const int n = running;
    if (tail.counts != nullptr) {                       // = count_opaque_kernel (lnr_render.hip) with far[0] = this batch's own first ray
        __shared__ int partial[16];
        const float far0 = n > 0 ? rays_out[12] : 0.0f;
        int c = 0;
        for (int i = tid; i < n; i += 1024) {
            const float d = depths_out[i];
            c += ((d > 0.0f) && !(d > far0)) ? 1 : 0;
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0) partial[wave] = c;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += partial[w];
            tail.counts[0] = n;
            tail.counts[1] = t;
        }
    }
    if (tail.record != nullptr) {                       // = shard_front_pack_kernel below
        const int m = min(n, tail.cap);
        if (tid == 0) {
            long long k = 0x7FFFFFFFFFFFFFFFll;                      // no live ray on this rank
            for (int sgi = 0; sgi < tail.ord.n; ++sgi) {
                const int lo = out_seg_start[sgi], hi = out_seg_start[sgi + 1];
                if (hi > lo) {
                    k = ((long long)tail.ord.order[sgi] << 32) | (long long)__float_as_uint(rays_out[(size_t)lo * LNR_RAY_STRIDE + 12]);
                    break;
                }
            }
            tail.record[0] = __uint_as_float((uint32_t)((unsigned long long)k & 0xFFFFFFFFull));
            tail.record[1] = __uint_as_float((uint32_t)((unsigned long long)k >> 32));
            tail.record[2] = __int_as_float(m);
            tail.record[3] = 0.0f;
        }
        for (int i = tid; i < tail.cap; i += 1024) tail.record[LNR_FRONT_HEADER + i] = i < m ? depths_out[i] : 0.0f;
    }
}

extern "C" int lnr_compact_rays(const float* rays_in, const float* depths_in, const uint8_t* keep, const int64_t* src_index, int32_t n_in,
                                const int32_t* seg_start, int32_t n_seg, float* rays_out, float* depths_out, int64_t* src_index_out,
                                int32_t* out_seg_start, int32_t* n_out_dev, void* stream) {
    LNR_REQUIRE(rays_in && depths_in && keep && seg_start && rays_out && depths_out && out_seg_start && n_out_dev, "lnr_compact_rays: null argument");
    LNR_REQUIRE(n_seg >= 1 && n_seg <= LNR_MAX_SEG, "lnr_compact_rays: n_seg must be in [1,%d]", LNR_MAX_SEG);
    LNR_REQUIRE((src_index == nullptr) == (src_index_out == nullptr), "lnr_compact_rays: src_index in/out must both be given or both be null");
    SegTable seg;
    seg.n = n_seg;
    for (int s = 0; s <= n_seg; ++s) seg.start[s] = seg_start[s];
    LNR_REQUIRE(seg.start[0] == 0 && seg.start[n_seg] == n_in, "lnr_compact_rays: seg_start must run from 0 to n_in");
    CompactTail tail;
    tail.counts = nullptr; tail.record = nullptr; tail.cap = 0; tail.ord.n = 0;
    hipLaunchKernelGGL(compact_rays_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rays_in, depths_in, keep, src_index, n_in, seg,
                       rays_out, depths_out, src_index_out, out_seg_start, n_out_dev, tail);
    LNR_CHECK_LAUNCH("lnr_compact_rays");
    return LNR_OK;
}

extern "C" int lnr_compact_rays_front(const float* rays_in, const float* depths_in, const uint8_t* keep, const int64_t* src_index, int32_t n_in,
                                      const int32_t* seg_start, int32_t n_seg, float* rays_out, float* depths_out, int64_t* src_index_out,
                                      int32_t* out_seg_start, int32_t* n_out_dev, int32_t* counts_dev, const int32_t* seg_order, int32_t cap,
                                      float* record, void* stream) {
    LNR_REQUIRE(rays_in && depths_in && keep && seg_start && rays_out && depths_out && out_seg_start && n_out_dev, "lnr_compact_rays_front: null argument");
    LNR_REQUIRE(n_seg >= 1 && n_seg <= LNR_MAX_SEG, "lnr_compact_rays_front: n_seg must be in [1,%d]", LNR_MAX_SEG);
    LNR_REQUIRE((src_index == nullptr) == (src_index_out == nullptr), "lnr_compact_rays_front: src_index in/out must both be given or both be null");
    LNR_REQUIRE((counts_dev != nullptr) != (record != nullptr), "lnr_compact_rays_front: pass counts_dev (unsharded batch) or record (sharded), one of them");
    SegTable seg;
    seg.n = n_seg;
    for (int s = 0; s <= n_seg; ++s) seg.start[s] = seg_start[s];
    LNR_REQUIRE(seg.start[0] == 0 && seg.start[n_seg] == n_in, "lnr_compact_rays_front: seg_start must run from 0 to n_in");
    CompactTail tail;
    tail.counts = counts_dev; tail.record = record; tail.cap = cap; tail.ord.n = 0;
    if (record != nullptr) {
        LNR_REQUIRE(seg_order != nullptr && cap >= n_in, "lnr_compact_rays_front: a front record of %d depth slots cannot hold the %d candidate rays of this rank", cap, n_in);
        tail.ord.n = n_seg;
        for (int s = 0; s < n_seg; ++s) {
            LNR_REQUIRE(seg_order[s] >= 0 && (s == 0 || seg_order[s] > seg_order[s - 1]), "lnr_compact_rays_front: seg_order must be non-negative and ascending");
            tail.ord.order[s] = seg_order[s];
        }
    }
    hipLaunchKernelGGL(compact_rays_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rays_in, depths_in, keep, src_index, n_in, seg,
                       rays_out, depths_out, src_index_out, out_seg_start, n_out_dev, tail);
    LNR_CHECK_LAUNCH("lnr_compact_rays_front");
    return LNR_OK;
}

// Sharded windows: which ray is "the first ray of the whole batch" (the reference compares every depth with ITS far, optimizer.py:460-461).
// Every rank reports {window order of its first segment that kept a ray, the far of that ray} as one 64-bit key - order in the high
// word, the float's bits in the low word - and a MIN all-reduce over the ranks leaves the key of the batch's first ray everywhere.
__global__ void first_ray_key_kernel(const float* __restrict__ rays, const int32_t* __restrict__ out_seg_start, const SegOrder ord, long long* __restrict__ key) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long k = 0x7FFFFFFFFFFFFFFFll;                      // no live ray on this rank
    for (int s = 0; s < ord.n; ++s) {
        const int lo = out_seg_start[s], hi = out_seg_start[s + 1];
        if (hi > lo) {
            k = ((long long)ord.order[s] << 32) | (long long)__float_as_uint(rays[(size_t)lo * LNR_RAY_STRIDE + 12]);
            break;
        }
    }
    *key = k;
}

extern "C" int lnr_first_ray_key(const float* rays, const int32_t* out_seg_start, const int32_t* seg_order, int32_t n_seg, int64_t* key_out, void* stream) {
    LNR_REQUIRE(rays && out_seg_start && seg_order && key_out, "lnr_first_ray_key: null argument");
    LNR_REQUIRE(n_seg >= 1 && n_seg <= LNR_MAX_SEG, "lnr_first_ray_key: n_seg must be in [1,%d]", LNR_MAX_SEG);
    SegOrder ord;
    ord.n = n_seg;
    for (int s = 0; s < n_seg; ++s) {
        LNR_REQUIRE(seg_order[s] >= 0 && (s == 0 || seg_order[s] > seg_order[s - 1]), "lnr_first_ray_key: seg_order must be non-negative and ascending");
        ord.order[s] = seg_order[s];
    }
    hipLaunchKernelGGL(first_ray_key_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rays, out_seg_start, ord, reinterpret_cast<long long*>(key_out));
    LNR_CHECK_LAUNCH("lnr_first_ray_key");
    return LNR_OK;
}

// ------------------------------------------------------------------------------------------------
// The sharded loop's ONE small collective per iteration (mapping/sharding.py).  What the loss needs from the other ranks - far[0] of
// the whole batch and the two global normalisers #rays / #opaque rays (optimizer.py:460-463,488-489,569-578) - depends on far[0] in
// its second half (opaque = depth > 0 and not depth > far[0]), so a key exchange followed by a count exchange would be two dependent
// collectives.  Instead every rank contributes one "front record": its first-ray key, its live-ray count and the ground-truth depths of
// its kept rays (2 KB for 512 rays), the records are all-gathered, and every rank derives far[0] and both counts from ALL depths
// locally - identical integer results everywhere, one latency-bound collective that hides behind the sampler and the density forward.
//   record = float32 words [LNR_FRONT_HEADER + cap]:  [0..1] key (int64 bits)   [2] live rays (int32 bits)   [3] 0   [4..] depths
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
shard_front_pack_kernel(const float* __restrict__ rays, const int32_t* __restrict__ out_seg_start, const SegOrder ord,
                        const float* __restrict__ depths, int n_rays, const int32_t* __restrict__ n_rays_dev, int cap, float* __restrict__ rec) {
    const int n = min(lnr_live_rays(n_rays, n_rays_dev), cap);
    if (threadIdx.x == 0) {
        long long k = 0x7FFFFFFFFFFFFFFFll;                      // no live ray on this rank
        for (int s = 0; s < ord.n; ++s) {
            const int lo = out_seg_start[s], hi = out_seg_start[s + 1];
            if (hi > lo) {
                k = ((long long)ord.order[s] << 32) | (long long)__float_as_uint(rays[(size_t)lo * LNR_RAY_STRIDE + 12]);
                break;
            }
        }
        rec[0] = __uint_as_float((uint32_t)((unsigned long long)k & 0xFFFFFFFFull));
        rec[1] = __uint_as_float((uint32_t)((unsigned long long)k >> 32));
        rec[2] = __int_as_float(n);
        rec[3] = 0.0f;
    }
    for (int i = threadIdx.x; i < cap; i += blockDim.x) rec[LNR_FRONT_HEADER + i] = i < n ? depths[i] : 0.0f;
}

__global__ void __launch_bounds__(1024)
shard_front_reduce_kernel(const float* __restrict__ recs, int world, int stride, int32_t* __restrict__ counts, float* __restrict__ far0_out) {
    __shared__ int partial[16];
    unsigned long long kmin = 0x7FFFFFFFFFFFFFFFull;
    int n_all = 0;
    for (int r = 0; r < world; ++r) {
        const float* rec = recs + (size_t)r * stride;
        const unsigned long long k = (unsigned long long)__float_as_uint(rec[0]) | ((unsigned long long)__float_as_uint(rec[1]) << 32);
        kmin = k < kmin ? k : kmin;                              // (keys are non-negative as signed values: the unsigned order is the same)
        n_all += max(0, min(__float_as_int(rec[2]), stride - LNR_FRONT_HEADER));
    }
    const float far0 = __uint_as_float((uint32_t)(kmin & 0xFFFFFFFFull));      // nobody has a ray: NaN bits, and no depth is compared
    int c = 0;
    for (int r = 0; r < world; ++r) {
        const float* rec = recs + (size_t)r * stride;
        const int n = max(0, min(__float_as_int(rec[2]), stride - LNR_FRONT_HEADER));
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float d = rec[LNR_FRONT_HEADER + i];
            c += ((d > 0.0f) && !(d > far0)) ? 1 : 0;            // lnr_count_opaque's test
        }
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += partial[w];
        counts[0] = n_all;
        counts[1] = t;
        far0_out[0] = far0;
    }
}

extern "C" int lnr_shard_front_pack(const float* rays, const int32_t* out_seg_start, const int32_t* seg_order, int32_t n_seg,
                                    const float* depths, int32_t n_rays, const int32_t* n_rays_dev, int32_t cap, float* record, void* stream) {
    LNR_REQUIRE(record && cap >= 0 && n_rays >= 0 && n_seg >= 0 && n_seg <= LNR_MAX_SEG, "lnr_shard_front_pack: bad argument");
    LNR_REQUIRE(n_seg == 0 || (rays && out_seg_start && seg_order && depths), "lnr_shard_front_pack: null argument");
    // (n_rays is the capacity of the compacted ray buffer: the live count on the device cannot exceed it.  A record with room for fewer
    // depths would truncate the global #rays / #opaque normalisers silently and scale the loss wrongly - ADVICE r5)
    LNR_REQUIRE(n_seg == 0 || n_rays <= cap, "lnr_shard_front_pack: a front record of %d depth slots cannot hold the %d rays of this rank "
                "(every rank must size its record for the fullest rank: DistContext.front_capacity)", cap, n_rays);
    SegOrder ord;
    ord.n = n_seg;
    for (int s = 0; s < n_seg; ++s) {
        LNR_REQUIRE(seg_order[s] >= 0 && (s == 0 || seg_order[s] > seg_order[s - 1]), "lnr_shard_front_pack: seg_order must be non-negative and ascending");
        ord.order[s] = seg_order[s];
    }
    hipLaunchKernelGGL(shard_front_pack_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rays, out_seg_start, ord, depths,
                       n_seg == 0 ? 0 : n_rays, n_seg == 0 ? nullptr : n_rays_dev, cap, record);
    LNR_CHECK_LAUNCH("lnr_shard_front_pack");
    return LNR_OK;
}

extern "C" int lnr_shard_front_reduce(const float* records, int32_t world, int32_t stride, int32_t* counts_dev, float* far0_dev, void* stream) {
    LNR_REQUIRE(records && counts_dev && far0_dev && world >= 1 && stride >= LNR_FRONT_HEADER, "lnr_shard_front_reduce: bad argument");
    hipLaunchKernelGGL(shard_front_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, records, world, stride, counts_dev, far0_dev);
    LNR_CHECK_LAUNCH("lnr_shard_front_reduce");
    return LNR_OK;
}

// one workgroup per keyframe: reduce the 12 entries of dL/d[R|t]
__global__ void __launch_bounds__(256)
lidar_rays_backward_kernel(const float* __restrict__ d_rays, const float* __restrict__ rays, const int64_t* __restrict__ src_index,
                           const int32_t* __restrict__ seg_start, const DirTable dirs, const float* __restrict__ transforms,
                           float scale, float far_range_unused, float* __restrict__ d_transform) {
    __shared__ float red[4][12];
    const int seg = blockIdx.x;
    const int lo = seg_start[seg], hi = seg_start[seg + 1];
    const float* T = transforms + seg * 12;
    const float* D = dirs.dirs[seg];
    const int64_t n = dirs.n_points[seg];
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0f;
    for (int ray = lo + threadIdx.x; ray < hi; ray += blockDim.x) {
        const float* g = d_rays + (size_t)ray * LNR_RAY_STRIDE;
        const float* rec = rays + (size_t)ray * LNR_RAY_STRIDE;
        const int64_t src = src_index[ray];
        const float l[3] = {D[src], D[n + src], D[2 * n + src]};
        float go[3] = {g[0], g[1], g[2]};
        float gd[3] = {g[3] - g[6], g[4] - g[7], g[5] - g[8]};        // viewdir = -dir
        const float gfar = g[12];
        const float o[3] = {rec[0], rec[1], rec[2]};
        const float d[3] = {rec[3], rec[4], rec[5]};
        if (gfar != 0.0f) {
            int axis; float t_raw, dd;
            const float clip = cube_exit(o, d, &axis, &t_raw, &dd);
            // far = min(range_max/scale, clip): the gradient follows clip only where it is the smaller one,
            // i.e. where the stored far equals clip and the selected plane distance was not clamped at 0.
            if (clip <= rec[12] && t_raw > 0.0f) {
                const float inv = 1.0f / dd;
                const float go_a = -gfar * inv, gd_a = -gfar * t_raw * inv;
                if (axis == 0) { go[0] += go_a; gd[0] += gd_a; } else if (axis == 1) { go[1] += go_a; gd[1] += gd_a; } else { go[2] += go_a; gd[2] += gd_a; }
            }
        }
        // d = v/|v|, v = R l
        float v[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) v[r] = T[4 * r] * l[0] + T[4 * r + 1] * l[1] + T[4 * r + 2] * l[2];
        const float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const float dot = d[0] * gd[0] + d[1] * gd[1] + d[2] * gd[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float gv = (gd[r] - d[r] * dot) / nrm;
            acc[4 * r + 0] += gv * l[0];
            acc[4 * r + 1] += gv * l[1];
            acc[4 * r + 2] += gv * l[2];
            acc[4 * r + 3] += go[r] / scale;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 12) d_transform[seg * 12 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

extern "C" int lnr_lidar_rays_backward(const float* d_rays, const float* rays, const int64_t* src_index, const int32_t* seg_start,
                                       int32_t n_seg, const float* const* directions, const int64_t* n_points, const float* transforms,
                                       float scale, float* d_transform, void* stream) {
    LNR_REQUIRE(d_rays && rays && src_index && seg_start && directions && n_points && transforms && d_transform, "lnr_lidar_rays_backward: null argument");
    LNR_REQUIRE(n_seg >= 1 && n_seg <= LNR_MAX_SEG, "lnr_lidar_rays_backward: n_seg must be in [1,%d]", LNR_MAX_SEG);
    DirTable tab;
    for (int s = 0; s < n_seg; ++s) { tab.dirs[s] = directions[s]; tab.n_points[s] = n_points[s]; }
    hipLaunchKernelGGL(lidar_rays_backward_kernel, dim3(n_seg), dim3(256), 0, (hipStream_t)stream, d_rays, rays, src_index, seg_start, tab,
                       transforms, scale, 0.0f, d_transform);
    LNR_CHECK_LAUNCH("lnr_lidar_rays_backward");
    return LNR_OK;
}
