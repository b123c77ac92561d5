// Ray samplers and occupancy lookup (gfx950).  Compile with -ffp-contract=off: the sample
// indices must be bit-identical to the reference's torch CPU path, so every float operation is
// rounded separately, sums follow ATen's cascade order and the cdf is a sequential float64
// running sum (oracle/torch_rounding.py is the CPU statement of the rules).
//
// Replaces  OccGridRaySampler.get_samples   src/models/ray_sampling.py:53-92
//           UniformRaySampler.get_samples   src/models/ray_sampling.py:22-43
//           sample_pdf                      src/models/rendering_tcnn.py:18-67
//           OccupancyGridModel.interpolate  src/models/model_tcnn.py:122-131
//
// One workgroup (256 threads) per ray; per-ray state (coarse depths, occupancy probabilities,
// pdf/cdf, importance samples, the merge buffer) lives in LDS and never touches HBM.
#include "lnr_common.h"

#define SAMPLER_BLOCK 256

// grid_sample(mode='bilinear', align_corners=False, padding zeros) on a [V,V,V] (z,y,x) volume,
// with torch's exact association order.
__device__ __forceinline__ float trilinear_zero_pad(const float* __restrict__ grid, int V, float x, float y, float z) {
    const float fV = (float)V;
    const float ix = ((x + 1.0f) * fV - 1.0f) / 2.0f;
    const float iy = ((y + 1.0f) * fV - 1.0f) / 2.0f;
    const float iz = ((z + 1.0f) * fV - 1.0f) / 2.0f;
    const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f, z1 = z0 + 1.0f;
    const float wx0 = x1 - ix, wx1 = ix - x0;
    const float wy0 = y1 - iy, wy1 = iy - y0;
    const float wz0 = z1 - iz, wz1 = iz - z0;
    float out = 0.0f;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const float xc = (corner & 1) ? x1 : x0, yc = (corner & 2) ? y1 : y0, zc = (corner & 4) ? z1 : z0;
        const float w = (((corner & 1) ? wx1 : wx0) * ((corner & 2) ? wy1 : wy0)) * ((corner & 4) ? wz1 : wz0);
        if (xc >= 0.0f && xc < fV && yc >= 0.0f && yc < fV && zc >= 0.0f && zc < fV) {
            const float v = grid[((int)zc * V + (int)yc) * V + (int)xc];
            out = out + v * w;
        }
    }
    return out;
}

__global__ void occ_interpolate_kernel(const float* __restrict__ grid, int V, const float* __restrict__ pts, int64_t n,
                                       float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = trilinear_zero_pad(grid, V, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
}

extern "C" int lnr_occ_interpolate(const float* grid, int32_t V, const float* pts, int64_t n, float* out, void* stream) {
    LNR_REQUIRE(grid && pts && out && V > 0 && n >= 0, "lnr_occ_interpolate: bad argument");
    if (n == 0) return LNR_OK;
    hipLaunchKernelGGL(occ_interpolate_kernel, dim3(lnr_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, grid, V, pts, n, out);
    LNR_CHECK_LAUNCH("lnr_occ_interpolate");
    return LNR_OK;
}

// ------------------------------------------------------------------------------------------------
// stratified coarse depths (shared by both samplers): writes zc[0..H)
// ------------------------------------------------------------------------------------------------
__device__ void stratified_depths(float near, float far, int H, float perturb, const float* __restrict__ steps,
                                  const float* __restrict__ u_row, uint64_t seed, uint64_t ray,
                                  float* zc, float* tmp) {
    for (int j = threadIdx.x; j < H; j += SAMPLER_BLOCK) {
        const float s = steps[j];
        tmp[j] = near * (1.0f - s) + far * s;
    }
    __syncthreads();
    if (perturb > 0.0f) {
        for (int j = threadIdx.x; j < H; j += SAMPLER_BLOCK) {
            const float zj = tmp[j];
            const float upper = (j < H - 1) ? 0.5f * (zj + tmp[j + 1]) : zj;
            const float lower = (j > 0) ? 0.5f * (tmp[j - 1] + zj) : zj;
            const float u = u_row ? u_row[j] : lnr_rand_uniform(seed, LNR_STREAM_JITTER, ray, (uint32_t)j);
            zc[j] = lower + (upper - lower) * (perturb * u);
        }
    } else {
        for (int j = threadIdx.x; j < H; j += SAMPLER_BLOCK) zc[j] = tmp[j];
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// torch.sum(x, -1) with ATen's cascade order, x[0..K) in LDS.  Result broadcast through *result.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ceil_log2_i(int n) { return n <= 1 ? 0 : 32 - __clz(n - 1); }

__device__ void aten_row_sum(const float* x, int K, float* part /*[32]*/, float* result) {
    const int V = K < 8 ? 1 : 8;        // vector lanes
    const int M = K / V;                // vectors
    const int groups = M / 4;           // rows of 4 interleaved vectors
    const int t = threadIdx.x;
    if (t < 4 * V) {
        const int k = t / V, lane = t % V;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        int power = ceil_log2_i(groups) / 4;
        if (power < 4) power = 4;
        const int step = 1 << power, mask = step - 1;
        int i = 0;
        while (i + step <= groups) {
            for (int j = 0; j < step; ++j, ++i) a0 = a0 + x[(4 * i + k) * V + lane];
            // cascade flush
            a1 = a1 + a0; a0 = 0.0f;
            if ((i & (mask << power)) == 0) {
                a2 = a2 + a1; a1 = 0.0f;
                if ((i & (mask << (2 * power))) == 0) { a3 = a3 + a2; a2 = 0.0f; }
            }
        }
        for (; i < groups; ++i) a0 = a0 + x[(4 * i + k) * V + lane];
        a0 = a0 + a1; a0 = a0 + a2; a0 = a0 + a3;
        part[t] = a0;
    }
    __syncthreads();
    if (t == 0) {
        float out[8];
        for (int lane = 0; lane < V; ++lane) {
            float p0 = part[0 * V + lane];
            for (int v = 4 * groups; v < M; ++v) p0 = p0 + x[v * V + lane];   // leftover vectors -> accumulator 0
            p0 = p0 + part[1 * V + lane];
            p0 = p0 + part[2 * V + lane];
            p0 = p0 + part[3 * V + lane];
            out[lane] = p0;
        }
        float total;
        if (V == 1) {
            total = out[0];
        } else {
            total = 0.0f;
            for (int j = M * V; j < K; ++j) total = total + x[j];
            for (int lane = 0; lane < V; ++lane) total = total + out[lane];
        }
        *result = total;
    }
    __syncthreads();
}

// bitonic sort of n_pow2 floats in LDS (ascending).  Each of the four waves owns a contiguous quarter of the array: a pass whose
// partner distance j is at most an eighth of the array exchanges inside the quarters, so it needs no workgroup barrier (a wave's LDS
// operations complete in order) - of the 45 passes of a 512-element sort (78 of a 4096-element one) only the three with j >= n / 4 do.
// descending: the mirrored network (every comparison flipped).  k_first: the first stage to run - n_pow2 runs only the last stage,
// the bitonic MERGE of an array whose first half ascends and whose second half descends.
__device__ void bitonic_sort_lds(float* a, int n_pow2, bool descending = false, int k_first = 2) {
    auto exchange = [&](int t, int j, int k) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));       // lower index of the pair
        const int p = i | j;
        const bool up = ((i & k) == 0) != descending;
        const float x = a[i], y = a[p];
        if ((x > y) == up) { a[i] = y; a[p] = x; }
    };
    if (n_pow2 < 256) {                                             // (a quarter of fewer than 64 elements: not worth the wave-local form)
        for (int k = k_first; k <= n_pow2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < n_pow2 / 2; t += SAMPLER_BLOCK) exchange(t, j, k);
                __syncthreads();
            }
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int quarter_pairs = n_pow2 / 8;                          // pairs inside a wave's quarter (n / 4 elements)
    bool local_before = false;
    for (int k = k_first; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (8 * j <= n_pow2) {                                  // partners inside the wave's quarter
                for (int t = wave * quarter_pairs + lane; t < (wave + 1) * quarter_pairs; t += 64) exchange(t, j, k);
                __builtin_amdgcn_wave_barrier();
                local_before = true;
            } else {
                if (local_before) __syncthreads();
                for (int t = threadIdx.x; t < n_pow2 / 2; t += SAMPLER_BLOCK) exchange(t, j, k);
                __syncthreads();
                local_before = false;
            }
        }
    }
    __syncthreads();
}

// LDS (floats): zc[H] | tmp[H] (reused as probs) | pdf[H] | cdf[H] | merged[P2] | part[32] | scalars[4]
__global__ void __launch_bounds__(SAMPLER_BLOCK)
sample_occ_kernel(const float* __restrict__ rays, int n_rays, const int32_t* __restrict__ n_rays_dev,
                  const float* __restrict__ grid, int V, int S, float perturb, const float* __restrict__ steps,
                  const float* __restrict__ u_jitter, const float* __restrict__ u_pdf, uint64_t seed,
                  float* __restrict__ z_out, int64_t* __restrict__ dbg_inds, float* __restrict__ dbg_probs,
                  float* __restrict__ dbg_cdf, int P2) {
    extern __shared__ float lds[];
    const int ray = blockIdx.x;
    if (ray >= lnr_live_rays(n_rays, n_rays_dev)) return;
    const int H = S / 2;
    const int K = H - 2;               // pdf bins
    float* zc = lds;
    float* probs = zc + H;
    float* pdf = probs + H;
    float* cdf = pdf + H;
    float* merged = cdf + H;
    float* part = merged + P2;
    float* scal = part + 32;

    const float* r = rays + (size_t)ray * LNR_RAY_STRIDE;
    const float ox = r[0], oy = r[1], oz = r[2], dx = r[3], dy = r[4], dz = r[5];
    const float near = r[11], far = r[12];

    stratified_depths(near, far, H, perturb, steps, u_jitter ? u_jitter + (size_t)ray * H : nullptr, seed, (uint64_t)ray, zc, probs);

    // occupancy probability at the coarse samples (ray_sampling.py:77-81)
    for (int j = threadIdx.x; j < H; j += SAMPLER_BLOCK) {
        const float zv = zc[j];
        const float logit = trilinear_zero_pad(grid, V, ox + dx * zv, oy + dy * zv, oz + dz * zv);
        const float e = (float)exp((double)(-logit));            // correctly rounded float32 exp
        float p = 1.0f / (1.0f + e);
        p = fminf(fmaxf(p, 0.5f), 1.0f);
        p = 2.0f * (p - 0.5f);
        probs[j] = p;
        if (dbg_probs) dbg_probs[(size_t)ray * H + j] = p;
    }
    __syncthreads();

    // pdf over the K interior intervals: w_k = probs[k+1] + 1e-5  (rendering_tcnn.py:33-35)
    for (int k = threadIdx.x; k < K; k += SAMPLER_BLOCK) pdf[k] = probs[k + 1] + 1e-5f;
    __syncthreads();
    aten_row_sum(pdf, K, part, &scal[0]);
    const float total = scal[0];
    for (int k = threadIdx.x; k < K; k += SAMPLER_BLOCK) pdf[k] = pdf[k] / total;
    __syncthreads();
    // cdf = [0, cumsum(pdf)] with the running sum in float64 (torch.cumsum on CPU).  The float64 sums are EXACT here - every pdf
    // value is a float32 >= 1e-5 / (K (1 + 1e-5)) > 2^-29 for K <= 4094, i.e. a multiple of 2^-52, and every partial sum is below 2 -
    // so the order of the additions does not matter: the first wave sums 64 chunks side by side and combines them with an exclusive
    // scan, instead of one thread walking K dependent additions (with K = 1022 at test time, ~40 k cycles per ray).
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, chunk = (K + 63) / 64;
        const int k0 = lane * chunk, k1 = min(K, k0 + chunk);
        double own = 0.0;
        for (int k = k0; k < k1; ++k) own = own + (double)pdf[k];
        double incl = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(incl, o, 64);
            if (lane >= o) incl = incl + up;
        }
        double run = incl - own;                                   // exclusive prefix (exact)
        if (lane == 0) cdf[0] = 0.0f;
        for (int k = k0; k < k1; ++k) {
            run = run + (double)pdf[k];
            cdf[k + 1] = (float)run;
        }
    }
    __syncthreads();
    if (dbg_cdf) for (int k = threadIdx.x; k <= K; k += SAMPLER_BLOCK) dbg_cdf[(size_t)ray * (K + 1) + k] = cdf[k];

    // inverse-cdf samples (rendering_tcnn.py:50-67); bins[k] = 0.5*(zc[k]+zc[k+1]), k = 0..K
    bool unsorted = false;
    for (int j = threadIdx.x; j < H; j += SAMPLER_BLOCK) {
        const float u = u_pdf ? u_pdf[(size_t)ray * H + j] : lnr_rand_uniform(seed, LNR_STREAM_PDF, (uint64_t)ray, (uint32_t)j);
        // searchsorted(right=True): number of cdf entries <= u (cdf has K+1 entries, ascending)
        int lo = 0, hi = K + 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int ind = lo;
        const int below = ind - 1 > 0 ? ind - 1 : 0;
        const int above = ind < K ? ind : K;
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = 0.5f * (zc[below] + zc[below + 1]);
        const float b1 = 0.5f * (zc[above] + zc[above + 1]);
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.0f;
        const float smp = b0 + (u - c0) / denom * (b1 - b0);
        merged[H + j] = smp;
        merged[j] = zc[j];
        unsorted |= (j + 1 < H) && (zc[j] > zc[j + 1]);
        if (dbg_inds) dbg_inds[(size_t)ray * H + j] = ind;
    }
    for (int j = S + threadIdx.x; j < P2; j += SAMPLER_BLOCK) merged[j] = __builtin_inff();
    // sort(cat(coarse, importance)) (ray_sampling.py:90; only the values matter).  The coarse depths already ascend (a jittered depth
    // stays inside its stratum), so when S is a power of two the full network is not needed: the importance half is sorted DESCENDING
    // by the four waves (a network of half the size), and the last stage of the full network merges the two halves - 36 + 9 instead
    // of 45 passes at S = 512, the 36 of them over half the pairs.  Same values out, whatever the route; any inversion among the coarse
    // depths (none has been seen) sends the ray down the full sort.
    if (__syncthreads_or(unsorted ? 1 : 0) == 0 && P2 == S && H >= 64) {
        bitonic_sort_lds(merged + H, H, true);
        bitonic_sort_lds(merged, P2, false, P2);
    } else {
        bitonic_sort_lds(merged, P2);
    }
    for (int j = threadIdx.x; j < S; j += SAMPLER_BLOCK) z_out[(size_t)ray * S + j] = merged[j];
}

__global__ void __launch_bounds__(SAMPLER_BLOCK)
sample_uniform_kernel(const float* __restrict__ rays, int n_rays, const int32_t* __restrict__ n_rays_dev, int S, float perturb,
                      const float* __restrict__ steps, const float* __restrict__ u_jitter, uint64_t seed,
                      float* __restrict__ z_out) {
    extern __shared__ float lds[];
    const int ray = blockIdx.x;
    if (ray >= lnr_live_rays(n_rays, n_rays_dev)) return;
    float* zc = lds;
    float* tmp = lds + S;
    const float* r = rays + (size_t)ray * LNR_RAY_STRIDE;
    stratified_depths(r[11], r[12], S, perturb, steps, u_jitter ? u_jitter + (size_t)ray * S : nullptr, seed, (uint64_t)ray, zc, tmp);
    for (int j = threadIdx.x; j < S; j += SAMPLER_BLOCK) z_out[(size_t)ray * S + j] = zc[j];
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

extern "C" int lnr_sample_rays_occ(const float* rays, int32_t n_rays, const int32_t* n_rays_dev, const float* grid, int32_t V,
                                   int32_t n_samples, float perturb, const float* steps, const float* u_jitter,
                                   const float* u_pdf, uint64_t seed, float* z_out, int64_t* dbg_inds, float* dbg_probs,
                                   float* dbg_cdf, void* stream) {
    LNR_REQUIRE(rays && grid && steps && z_out, "lnr_sample_rays_occ: null argument");
    LNR_REQUIRE(V > 0 && n_rays >= 0, "lnr_sample_rays_occ: bad V/n_rays");
    LNR_REQUIRE(n_samples >= 8 && n_samples % 2 == 0 && n_samples <= 8192, "lnr_sample_rays_occ: n_samples must be even, in [8, 8192] (got %d)", n_samples);
    if (n_rays == 0) return LNR_OK;
    const int H = n_samples / 2, P2 = next_pow2(n_samples);
    const size_t lds = (size_t)(4 * H + P2 + 32 + 4) * sizeof(float);
    hipLaunchKernelGGL(sample_occ_kernel, dim3(n_rays), dim3(SAMPLER_BLOCK), lds, (hipStream_t)stream, rays, n_rays, n_rays_dev,
                       grid, V, n_samples, perturb, steps, u_jitter, u_pdf, seed, z_out, dbg_inds, dbg_probs, dbg_cdf, P2);
    LNR_CHECK_LAUNCH("lnr_sample_rays_occ");
    return LNR_OK;
}

extern "C" int lnr_sample_rays_uniform(const float* rays, int32_t n_rays, const int32_t* n_rays_dev, int32_t n_samples,
                                       float perturb, const float* steps, const float* u_jitter, uint64_t seed, float* z_out,
                                       void* stream) {
    LNR_REQUIRE(rays && steps && z_out, "lnr_sample_rays_uniform: null argument");
    LNR_REQUIRE(n_samples >= 2 && n_samples <= 8192 && n_rays >= 0, "lnr_sample_rays_uniform: bad n_samples/n_rays");
    if (n_rays == 0) return LNR_OK;
    const size_t lds = (size_t)(2 * n_samples) * sizeof(float);
    hipLaunchKernelGGL(sample_uniform_kernel, dim3(n_rays), dim3(SAMPLER_BLOCK), lds, (hipStream_t)stream, rays, n_rays, n_rays_dev,
                       n_samples, perturb, steps, u_jitter, seed, z_out);
    LNR_CHECK_LAUNCH("lnr_sample_rays_uniform");
    return LNR_OK;
}


// ------------------------------------------------------------------------------------------------ diagnostics
// the draws of the in-kernel generator as tensors (include/loner_hip.h): the SAME device functions, indexed as the kernels index them
__global__ void rng_draws_kernel(int which, uint64_t seed, int n_rays, int n_per_ray, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_rays * n_per_ray) return;
    const uint64_t ray = (uint64_t)(i / n_per_ray);
    const uint32_t j = (uint32_t)(i % n_per_ray);
    float v;
    if (which == LNR_DRAW_JITTER) v = lnr_rand_uniform(seed, LNR_STREAM_JITTER, ray, j);
    else if (which == LNR_DRAW_PDF) v = lnr_rand_uniform(seed, LNR_STREAM_PDF, ray, j);
    else v = lnr_rand_uniform(seed, 0x44ull + (uint64_t)(which - LNR_DRAW_RAY_INDEX), (uint64_t)i >> 2, (uint32_t)i & 3u);   // build_window_rays_kernel
    out[i] = v;
}

extern "C" int lnr_rng_draws(int32_t which, uint64_t seed, int32_t n_rays, int32_t n_per_ray, float* out, void* stream) {
    LNR_REQUIRE(out != nullptr && n_rays >= 0 && n_per_ray > 0, "lnr_rng_draws: bad argument");
    LNR_REQUIRE(which == LNR_DRAW_JITTER || which == LNR_DRAW_PDF || which == LNR_DRAW_NOISE || which >= LNR_DRAW_RAY_INDEX, "lnr_rng_draws: unknown draw %d", which);
    const int64_t n = (int64_t)n_rays * n_per_ray;
    if (n == 0) return LNR_OK;
    // the density noise is evaluated in the translation unit of the kernels that use it (lnr_render.hip: this file is compiled with
    // -ffp-contract=off, that one is not, and the libm calls of the Box-Muller step inline differently under the two)
    if (which == LNR_DRAW_NOISE) return lnr_render_noise_draws(seed, n_rays, n_per_ray, out, (hipStream_t)stream);
    hipLaunchKernelGGL(rng_draws_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, which, seed, n_rays, n_per_ray, out);
    LNR_CHECK_LAUNCH("lnr_rng_draws");
    return LNR_OK;
}
