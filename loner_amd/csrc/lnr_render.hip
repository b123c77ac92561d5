// Volume rendering, line-of-sight loss and their analytic backward (gfx950).
//
// Replaces  raw2outputs                 src/models/rendering_tcnn.py:71-147
//           Optimizer.compute_loss      src/mapping/optimizer.py:437-595 (lidar branch)
//           get_weights_gt              src/models/losses.py:29-51
//           get_logits_grad             src/models/losses.py:54-62
//           calculate_JS/KL_divergence  src/mapping/optimizer.py:614-626
// and the autograd backward of all of them (loss.backward(), optimizer.py:366).
//
// One wavefront (64 lanes) per ray; lane L owns the C consecutive samples [L*C, L*C+C).  The
// transmittance product and the reverse "sum of G*w behind me" are wave-level scans (shuffles),
// per-ray reductions (opacity, depth, weighted mean / variance, target normaliser) are wave
// reductions.  All per-sample state stays in registers; HBM traffic is sigma, z (+noise) in and
// d_sigma (+optional weights) out.
#include "lnr_common.h"

#define RENDER_BLOCK 256
#define RAYS_PER_BLOCK (RENDER_BLOCK / 64)
static_assert(RAYS_PER_BLOCK == LNR_LOSS_RAYS_PER_BLOCK, "header constant out of date");

__device__ __forceinline__ float wave_excl_suffix_sum(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_down(inc, o, 64);
        if (lane + o < 64) inc += t;
    }
    float ex = __shfl_down(inc, 1, 64);
    return lane == 63 ? 0.0f : ex;
}

template <int C>
struct RayState {
    float z[C], e[C], T[C], w[C], r[C], dl[C];   // depth, exp(-delta*relu), transmittance, weight, relu(dens), delta*|d|
    bool pos[C];                                  // dens > 0
    float opacity, depth, variance, dnorm;
};

// forward volume rendering of one ray held by one wave
// A lane's C consecutive values of a ray's row through LDS: the wave loads the row in 1 KB instructions (lane l of instruction k takes
// elements 256 k + 4 l ..) and each lane reads its own chunk back.  Loaded directly, a lane's chunk of C = 32 floats starts 128 bytes
// after its neighbour's: every load instruction touches 64 cache lines for 16 bytes each, and the 8 instructions that walk a chunk
// evict each other's lines (four waves x two 8 KB rows against a 32 KB L1) - the compositing of a 2048-sample scan ran at 7 TB/s of L2
// traffic for 1 GB of input.  Rows of 32 floats are padded by 4 in LDS (lane stride 144 bytes: conflict-free ds_read_b128).
template <int C>
__device__ __forceinline__ void load_chunk_staged(const float* __restrict__ src_row, float* stage, int lane, float (&out)[C]) {
    static_assert(C % 4 == 0, "float4 pieces");
#pragma unroll
    for (int k = 0; k < C / 4; ++k) {
        const int i = 256 * k + 4 * lane;
        *reinterpret_cast<float4*>(stage + i + 4 * (i / C)) = *reinterpret_cast<const float4*>(src_row + i);
    }
    __builtin_amdgcn_wave_barrier();                            // a wave's LDS operations complete in order
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(stage + lane * (C + 4) + 4 * q);
        out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
    }
    __builtin_amdgcn_wave_barrier();
}

template <int C>
__device__ __forceinline__ void render_ray(RayState<C>& st, const float* __restrict__ sigma, const float* __restrict__ z,
                                           const float* __restrict__ noise, float noise_std, uint64_t seed, int ray,
                                           int S, int lane, const float* __restrict__ rayrec, float* stage = nullptr) {
    const int base = lane * C;
    const size_t row = (size_t)ray * S;
    float dens[C];
    bool staged = false;
    if constexpr (C >= 16) {
        if (stage != nullptr && S == 64 * C) {                 // wave-uniform: whole rows
            load_chunk_staged<C>(z + row, stage, lane, st.z);
            load_chunk_staged<C>(sigma + row, stage, lane, dens);
            if (noise) {
                float nz[C];
                load_chunk_staged<C>(noise + row, stage, lane, nz);
#pragma unroll
                for (int t = 0; t < C; ++t) dens[t] += nz[t];
            } else if (noise_std > 0.0f) {
#pragma unroll
                for (int t = 0; t < C; ++t) dens[t] += lnr_rand_normal(seed, (uint64_t)ray, (uint32_t)(base + t)) * noise_std;
            }
            staged = true;
        }
    }
    if (!staged) {
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int i = base + t;
        if (i < S) {
            st.z[t] = z[row + i];
            float n = 0.0f;
            if (noise) n = noise[row + i];
            else if (noise_std > 0.0f) n = lnr_rand_normal(seed, (uint64_t)ray, (uint32_t)i) * noise_std;
            dens[t] = sigma[row + i] + n;
        } else {
            st.z[t] = 0.0f;
            dens[t] = 0.0f;
        }
    }
    }
    const float dx = rayrec[3], dy = rayrec[4], dz = rayrec[5];
    st.dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    // z of the sample after my last one comes from the next lane
    const float z_next_lane = __shfl_down(st.z[0], 1, 64);
    float tprod = 1.0f;
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int i = base + t;
        float delta;
        if (i < S - 1) delta = ((t + 1 < C) ? st.z[(t + 1 < C) ? t + 1 : t] : z_next_lane) - st.z[t];
        else delta = 1e10f;
        delta *= st.dnorm;
        st.dl[t] = delta;
        st.pos[t] = dens[t] > 0.0f;
        st.r[t] = st.pos[t] ? dens[t] : 0.0f;
        st.e[t] = (i < S) ? expf(-delta * st.r[t]) : 1.0f;
        const float alpha = 1.0f - st.e[t];
        st.T[t] = tprod;                           // local exclusive product, fixed up below
        tprod *= (i < S) ? (1.0f - alpha + 1e-10f) : 1.0f;
    }
    const float lane_prefix = wave_excl_prod(tprod, lane);
    float o_part = 0.0f, d_part = 0.0f;
#pragma unroll
    for (int t = 0; t < C; ++t) {
        st.T[t] *= lane_prefix;
        st.w[t] = (base + t < S) ? (1.0f - st.e[t]) * st.T[t] : 0.0f;
        o_part += st.w[t];
        d_part += st.w[t] * st.z[t];
    }
    st.opacity = wave_sum(o_part);
    st.depth = wave_sum(d_part) + (1.0f - st.opacity) * rayrec[12];
    float v_part = 0.0f;
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const float q = st.depth - st.z[t];
        v_part += st.w[t] * q * q;
    }
    st.variance = wave_sum(v_part);
}

// backward of render_ray given G[t] = dL/dw (already including the depth/opacity/variance paths)
// and g_far = dL/dfar.  Writes d_sigma and the direct ray-record gradient.
template <int C>
__device__ __forceinline__ void render_ray_backward(const RayState<C>& st, const float G[C], float g_far, int ray, int S, int lane,
                                                    const float* __restrict__ rayrec, float* __restrict__ d_sigma,
                                                    float* __restrict__ d_rays) {
    const int base = lane * C;
    float gw[C];
    float local = 0.0f;
#pragma unroll
    for (int t = 0; t < C; ++t) { gw[t] = G[t] * st.w[t]; local += gw[t]; }
    const float behind_lanes = wave_excl_suffix_sum(local, lane);   // sum over lanes after mine
    float suffix = behind_lanes;                                     // running sum over k > i
    float dnorm_part = 0.0f;
#pragma unroll
    for (int t = C - 1; t >= 0; --t) {
        const int i = base + t;
        if (i < S) {
            const float alpha = 1.0f - st.e[t];
            const float tt = 1.0f - alpha + 1e-10f;     // the factor the forward pass multiplied by
            const float d_alpha = G[t] * st.T[t] - suffix / tt;
            const float d_x = d_alpha * st.e[t];        // x = delta' * relu(dens)
            d_sigma[(size_t)ray * S + i] = st.pos[t] ? d_x * st.dl[t] : 0.0f;
            // delta' = delta * |d| -> gradient to |d| (delta itself carries no gradient: z is detached)
            if (st.dnorm > 0.0f) dnorm_part += d_x * st.r[t] * (st.dl[t] / st.dnorm);
        }
        suffix += gw[t];
    }
    const float d_norm = wave_sum(dnorm_part);
    if (lane < LNR_RAY_STRIDE) {
        float v = 0.0f;
        if (lane >= 3 && lane < 6 && st.dnorm > 0.0f) v = d_norm * rayrec[lane] / st.dnorm;
        if (lane == 12) v = g_far;
        d_rays[(size_t)ray * LNR_RAY_STRIDE + lane] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// plain rendering forward / backward (API-parity path: Model.forward + autograd)
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(RENDER_BLOCK)
render_forward_kernel(const float* __restrict__ sigma, const float* __restrict__ z, const float* __restrict__ rays, int n_rays,
                      const int32_t* __restrict__ n_rays_dev, int S, const float* __restrict__ noise, float noise_std,
                      uint64_t seed, float* __restrict__ depth, float* __restrict__ weights, float* __restrict__ opacity,
                      float* __restrict__ variance) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= lnr_live_rays(n_rays, n_rays_dev)) return;
    const float* rr = rays + (size_t)ray * LNR_RAY_STRIDE;
    RayState<C> st;
    extern __shared__ __attribute__((aligned(16))) float render_stage[];       // C >= 16: 64 (C + 4) floats per wave (the host sizes it)
    float* stage = C >= 16 ? render_stage + (threadIdx.x >> 6) * 64 * (C + 4) : nullptr;
    render_ray<C>(st, sigma, z, noise, noise_std, seed, ray, S, lane, rr, stage);
    if (weights) {
#pragma unroll
        for (int t = 0; t < C; ++t) if (lane * C + t < S) weights[(size_t)ray * S + lane * C + t] = st.w[t];
    }
    if (lane == 0) {
        if (depth) depth[ray] = st.depth;
        if (opacity) opacity[ray] = st.opacity;
        if (variance) variance[ray] = st.variance;
    }
}

template <int C>
__global__ void __launch_bounds__(RENDER_BLOCK)
render_backward_kernel(const float* __restrict__ sigma, const float* __restrict__ z, const float* __restrict__ rays, int n_rays,
                       const int32_t* __restrict__ n_rays_dev, int S, const float* __restrict__ noise, float noise_std,
                       uint64_t seed, const float* __restrict__ g_depth, const float* __restrict__ g_weights,
                       const float* __restrict__ g_opacity, const float* __restrict__ g_variance,
                       float* __restrict__ d_sigma, float* __restrict__ d_rays) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= lnr_live_rays(n_rays, n_rays_dev)) return;
    const float* rr = rays + (size_t)ray * LNR_RAY_STRIDE;
    RayState<C> st;
    render_ray<C>(st, sigma, z, noise, noise_std, seed, ray, S, lane, rr);
    const float gD = g_depth ? g_depth[ray] : 0.0f;
    const float gO = g_opacity ? g_opacity[ray] : 0.0f;
    const float gV = g_variance ? g_variance[ray] : 0.0f;
    // variance depends on depth: dV/dD = 2 * sum_i w_i (D - z_i)
    float q_part = 0.0f;
#pragma unroll
    for (int t = 0; t < C; ++t) q_part += st.w[t] * (st.depth - st.z[t]);
    const float gD_tot = gD + gV * 2.0f * wave_sum(q_part);
    const float far = rr[12];
    float G[C];
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int i = lane * C + t;
        const float q = st.depth - st.z[t];
        float g = gO + gD_tot * (st.z[t] - far) + gV * q * q;
        if (g_weights && i < S) g += g_weights[(size_t)ray * S + i];
        G[t] = (i < S) ? g : 0.0f;
    }
    render_ray_backward<C>(st, G, gD_tot * (1.0f - st.opacity), ray, S, lane, rr, d_sigma, d_rays);
}

// ------------------------------------------------------------------------------------------------
// target weights (losses.py:29-51) for one ray, shared by the fused loss and lnr_weights_gt
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float std_normal_cdf(float x) { return 0.5f * (1.0f + erff(x / 1.41421356237309515f)); }

// un-normalised truncated-Gaussian density at sample depth s (metres)
__device__ __forceinline__ float target_raw(float s, float g, float eps, float sd, float clip_norm) {
    const bool inside = (s - (g - eps) > 0.0f) && ((g + eps) - s > 0.0f);    // heaviside(., 0)
    if (!inside) return 0.0f;
    const float t = (s - g) / sd;
    return 0.39894228040143270f * expf(-0.5f * (t * t)) / sd / clip_norm;
}

__device__ __forceinline__ void target_params(float g, float eps, float* sd, float* clip_norm) {
    *sd = eps / 3.0f;
    const float lo = (g - eps - g) / *sd;
    const float hi = (g + eps - g) / *sd;
    *clip_norm = std_normal_cdf(hi) - std_normal_cdf(lo);
}

__device__ __forceinline__ float gaussian_kl(float m1, float s1, float m2, float s2) {
    const float d = m1 - m2;
    return logf(s2 / s1) + (s1 * s1 + d * d) / (2.0f * (s2 * s2)) - 0.5f;
}
__device__ __forceinline__ float gaussian_js(float m1, float s1, float m2, float s2) {
    const float mm = 0.5f * (m1 + m2);
    const float sm = 0.5f * sqrtf(s1 * s1 + s2 * s2);
    return 0.5f * gaussian_kl(m1, s1, mm, sm) + 0.5f * gaussian_kl(m2, s2, mm, sm);
}

// one ray of Optimizer.compute_loss held by one wave; terms[0..4] += {total, depth, los, opacity, eps}
template <int C>
__device__ __forceinline__ void los_loss_ray(const float* __restrict__ sigma, const float* __restrict__ z, const float* __restrict__ rays,
                                             const float* __restrict__ depth_gt, int S, const float* __restrict__ noise, float noise_std,
                                             uint64_t seed, float scale, const LnrLossConfig& cfg, const int32_t* __restrict__ counts,
                                             float* terms, bool terms_atomic, float* __restrict__ d_sigma, float* __restrict__ d_rays,
                                             float* __restrict__ ray_stats, float* __restrict__ weights_out, int ray, int lane,
                                             const float* __restrict__ far0_dev) {
    const float* rr = rays + (size_t)ray * LNR_RAY_STRIDE;
    RayState<C> st;
    render_ray<C>(st, sigma, z, noise, noise_std, seed, ray, S, lane, rr);

    // masks, incl. the reference's broadcast quirk: every depth is compared with far of ray 0 - of the WHOLE batch, which a
    // sharded caller passes in far0_dev (rank 0's first ray); a single-GPU batch reads its own first ray
    const float dgt = depth_gt[ray];
    const float far0 = far0_dev ? far0_dev[0] : rays[12];
    const bool opaque = (dgt > 0.0f) && !(dgt > far0);
    const float g = dgt * scale;                       // metres
    const float n_all = (float)counts[0] * (float)S;
    const float n_op = (float)counts[1];

    // weighted mean / variance of the sample depths (metres) under the rendered weights
    float m_part = 0.0f;
#pragma unroll
    for (int t = 0; t < C; ++t) m_part += (st.z[t] * scale) * st.w[t];
    const float wden = st.opacity + 1e-10f;
    const float mean = wave_sum(m_part) / wden;
    float v_part = 0.0f;
#pragma unroll
    for (int t = 0; t < C; ++t) { const float q = st.z[t] * scale - mean; v_part += q * q * st.w[t]; }
    const float var = wave_sum(v_part) / wden + 1e-10f;
    const float sd_pred = sqrtf(var);
    const float js = gaussian_js(g, cfg.min_eps / 3.0f, mean, sd_pred);

    float eps;
    if (cfg.selection <= 1) {           // L1_JS / L2_JS: dynamic margin
        float score = js;
        if (score < cfg.min_js) score = 0.0f;
        if (score > cfg.max_js) score = cfg.max_js;
        eps = cfg.min_eps * (1.0f + cfg.js_alpha * score);
    } else {
        eps = cfg.fixed_eps;
    }
    float sd_t, clip_norm;
    target_params(g, eps, &sd_t, &clip_norm);
    float raw[C];
    float r_part = 0.0f;
#pragma unroll
    for (int t = 0; t < C; ++t) {
        raw[t] = (lane * C + t < S) ? target_raw(st.z[t] * scale, g, eps, sd_t, clip_norm) : 0.0f;
        r_part += raw[t];
    }
    const float rden = wave_sum(r_part) + 1e-6f;

    const bool l1 = (cfg.selection == 0 || cfg.selection == 2);
    const float depth_m = st.depth * scale;
    const float gO = opaque ? ((st.opacity - 1.0f > 0.0f) ? 1.0f : (st.opacity - 1.0f < 0.0f ? -1.0f : 0.0f)) / n_op : 0.0f;
    const float gD = opaque ? cfg.depth_lambda * 2.0f * (depth_m - g) * scale / n_op : 0.0f;
    const float far = rr[12];
    float G[C];
    float los_part = 0.0f;
#pragma unroll
    for (int t = 0; t < C; ++t) {
        const int i = lane * C + t;
        const float target = opaque ? raw[t] / rden : 0.0f;
        const float diff = st.w[t] - target;
        float gw;
        if (l1) { los_part += fabsf(diff); gw = (diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f)); }
        else { los_part += diff * diff; gw = 2.0f * diff; }
        G[t] = (i < S) ? cfg.los_lambda * gw / n_all + gO + gD * (st.z[t] - far) : 0.0f;
        if (weights_out && i < S) weights_out[(size_t)ray * S + i] = st.w[t];
    }
    const float los_sum = wave_sum(los_part);
    render_ray_backward<C>(st, G, gD * (1.0f - st.opacity), ray, S, lane, rr, d_sigma, d_rays);

    if (lane == 0) {
        const float l_depth = opaque ? cfg.depth_lambda * (depth_m - g) * (depth_m - g) / n_op : 0.0f;
        const float l_los = cfg.los_lambda * los_sum / n_all;
        const float l_op = opaque ? fabsf(st.opacity - 1.0f) / n_op : 0.0f;
        const float t5[5] = {l_depth + l_los + l_op, l_depth, l_los, l_op, eps};     // eps: sum of the per-ray margins (-> _depth_eps = mean)
#pragma unroll
        for (int q = 0; q < 5; ++q) { if (terms_atomic) atomicAdd(terms + q, t5[q]); else terms[q] = t5[q]; }
        if (ray_stats) {
            float* o = ray_stats + (size_t)ray * 8;
            o[0] = st.depth; o[1] = st.opacity; o[2] = st.variance; o[3] = mean; o[4] = sd_pred; o[5] = js; o[6] = eps;
            o[7] = opaque ? 1.0f : 0.0f;
        }
    }
}


template <int C>
__global__ void __launch_bounds__(RENDER_BLOCK)
los_loss_fused_kernel(const float* __restrict__ sigma, const float* __restrict__ z, const float* __restrict__ rays,
                      const float* __restrict__ depth_gt, int n_rays, const int32_t* __restrict__ n_rays_dev, int S,
                      const float* __restrict__ noise, float noise_std, uint64_t seed, float scale, const LnrLossConfig cfg,
                      const int32_t* __restrict__ counts, float* __restrict__ loss_out, float* __restrict__ d_sigma,
                      float* __restrict__ d_rays, float* __restrict__ ray_stats, float* __restrict__ weights_out,
                      float* __restrict__ block_partials, const float* __restrict__ far0_dev) {
    // Loss terms: 20 k same-address float atomics (5 per ray) serialise in one L2 channel and cost ~0.25 ms - more than
    // the rest of the kernel.  With block_partials each workgroup stores its 5 sums and loss_reduce_kernel adds them up.
    __shared__ float s_terms[RAYS_PER_BLOCK][5];
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    const bool alive = ray < lnr_live_rays(n_rays, n_rays_dev);         // wave-uniform
    if (block_partials) {
        if (lane < 5) s_terms[threadIdx.x >> 6][lane] = 0.0f;
        if (alive) los_loss_ray<C>(sigma, z, rays, depth_gt, S, noise, noise_std, seed, scale, cfg, counts, s_terms[threadIdx.x >> 6], false,
                                   d_sigma, d_rays, ray_stats, weights_out, ray, lane, far0_dev);
        __syncthreads();
        if (threadIdx.x < 8) {
            float v = 0.0f;
            if (threadIdx.x < 5) for (int r = 0; r < RAYS_PER_BLOCK; ++r) v += s_terms[r][threadIdx.x];
            block_partials[(size_t)blockIdx.x * 8 + threadIdx.x] = v;
        }
    } else if (alive) {
        los_loss_ray<C>(sigma, z, rays, depth_gt, S, noise, noise_std, seed, scale, cfg, counts, loss_out, true,
                        d_sigma, d_rays, ray_stats, weights_out, ray, lane, far0_dev);
    }
}

// poison (nullable, int32[2] = {code, tag}): a NaN total marks the run as failed at iteration `tag` (LNR_POISON_NAN_LOSS) - the
// reference asserts "NaN Loss Encountered" inside compute_loss, before backward and step (optimizer.py:590); here the steps that
// follow read the word and become no-ops (lnr_adam_step, lnr_occ_grid_apply), and the host raises at the end of the phase.
__global__ void loss_reduce_kernel(const float* __restrict__ block_partials, int n_blocks, float* __restrict__ loss_out,
                                   int32_t* __restrict__ poison, int32_t poison_tag) {
    __shared__ float part[4][8];
    const int term = threadIdx.x & 7, slice = threadIdx.x >> 3;          // 256 threads = 32 slices x 8 terms
    float v = 0.0f;
    for (int b = slice; b < n_blocks; b += 32) v += block_partials[(size_t)b * 8 + term];
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if ((threadIdx.x & 63) < 8) part[threadIdx.x >> 6][term] = v;
    __syncthreads();
    if (threadIdx.x < 5) {
        const float t = loss_out[threadIdx.x] + (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
        loss_out[threadIdx.x] = t;
        if (threadIdx.x == 0 && poison != nullptr && t != t && atomicCAS(poison, 0, LNR_POISON_NAN_LOSS) == 0) poison[1] = poison_tag;
    }
}

__global__ void count_opaque_kernel(const float* __restrict__ rays, const float* __restrict__ depth_gt, int n_rays,
                                    const int32_t* __restrict__ n_rays_dev, int32_t* __restrict__ counts,
                                    const float* __restrict__ far0_dev) {
    __shared__ int partial[16];
    const int n = lnr_live_rays(n_rays, n_rays_dev);
    const float far0 = far0_dev ? far0_dev[0] : (n > 0 ? rays[12] : 0.0f);
    int c = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float d = depth_gt[i];
        c += ((d > 0.0f) && !(d > far0)) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += partial[w];
        counts[0] = n;
        counts[1] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// standalone elementwise pieces of the reference API
// ------------------------------------------------------------------------------------------------
__global__ void weights_gt_kernel(const float* __restrict__ s, const float* __restrict__ g, const float* __restrict__ eps_ray,
                                  float eps_scalar, int normalise, int n_rays, int S, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float gg = g[ray];
    const float eps = eps_ray ? eps_ray[ray] : eps_scalar;
    float sd, cn;
    target_params(gg, eps, &sd, &cn);
    float part = 0.0f;
    for (int i = lane; i < S; i += 64) part += target_raw(s[(size_t)ray * S + i], gg, eps, sd, cn);
    const float den = wave_sum(part) + 1e-6f;
    for (int i = lane; i < S; i += 64) {
        const float v = target_raw(s[(size_t)ray * S + i], gg, eps, sd, cn);
        out[(size_t)ray * S + i] = normalise ? v / den : v;
    }
}

__global__ void logits_grad_kernel(const float* __restrict__ s, const float* __restrict__ g, int64_t total, int S, float margin,
                                   float l_free, float l_occ, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float x = s[i] - g[i / S];
    const float free_side = (-x - margin > 0.0f) ? 1.0f : 0.0f;
    const float near_surface = ((x + margin > 0.0f) && (margin - x > 0.0f)) ? 1.0f : 0.0f;
    out[i] = l_free * free_side - l_occ * near_surface;
}

__global__ void points_grad_to_rays_kernel(const float* __restrict__ d_pts, const float* __restrict__ z, int n_rays,
                                           const int32_t* __restrict__ n_rays_dev, int S, float* __restrict__ d_rays) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (ray >= lnr_live_rays(n_rays, n_rays_dev)) return;
    float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = lane; i < S; i += 64) {
        const size_t m = (size_t)ray * S + i;
        const float zv = z[m];
        const float gx = d_pts[3 * m], gy = d_pts[3 * m + 1], gz = d_pts[3 * m + 2];
        a[0] += gx; a[1] += gy; a[2] += gz;
        a[3] += zv * gx; a[4] += zv * gy; a[5] += zv * gz;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k] = wave_sum(a[k]);
    if (lane < 6) {
        float v = lane == 0 ? a[0] : lane == 1 ? a[1] : lane == 2 ? a[2] : lane == 3 ? a[3] : lane == 4 ? a[4] : a[5];
        d_rays[(size_t)ray * LNR_RAY_STRIDE + lane] += v;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int chunk_for(int S) {
    int c = (S + 63) / 64;
    int p = 1;
    while (p < c) p <<= 1;
    return p;
}

#define DISPATCH_C(S, CALL)                                    \
    switch (chunk_for(S)) {                                    \
        case 1: { constexpr int C = 1; CALL; } break;          \
        case 2: { constexpr int C = 2; CALL; } break;          \
        case 4: { constexpr int C = 4; CALL; } break;          \
        case 8: { constexpr int C = 8; CALL; } break;          \
        case 16: { constexpr int C = 16; CALL; } break;        \
        case 32: { constexpr int C = 32; CALL; } break;        \
        default:                                               \
            lnr_set_error("n_samples=%d not supported (max 2048)", S); \
            return LNR_ERR_UNSUPPORTED;                        \
    }

extern "C" int lnr_render_forward(const float* sigma, const float* z, const float* rays, int32_t n_rays, const int32_t* n_rays_dev,
                                  int32_t n_samples, const float* noise, float noise_std, uint64_t seed, float* depth,
                                  float* weights, float* opacity, float* variance, void* stream) {
    LNR_REQUIRE(sigma && z && rays && n_rays >= 0 && n_samples >= 2, "lnr_render_forward: bad argument");
    if (n_rays == 0) return LNR_OK;
    const dim3 grid(lnr_div_up(n_rays, RAYS_PER_BLOCK)), block(RENDER_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_C(n_samples, hipLaunchKernelGGL(render_forward_kernel<C>, grid, block, C >= 16 ? RAYS_PER_BLOCK * 64 * (C + 4) * sizeof(float) : 0, st,
                                             sigma, z, rays, n_rays, n_rays_dev, n_samples, noise, noise_std, seed, depth, weights, opacity, variance));
    LNR_CHECK_LAUNCH("lnr_render_forward");
    return LNR_OK;
}

// ------------------------------------------------------------------------------------------------
// Front-to-back inference (opt-in route of Model.render_depth; the reference composites every sample of every ray,
// src/models/model_tcnn.py:73-105 -> rendering_tcnn.py:71-147).  A front-to-back composite stops contributing once the transmittance is
// gone: with T < 2^-24 in front of a sample every remaining weight is below the fp32 resolution of the rendered depth.  The ray's 2048
// sorted depths are still drawn in full (the sampler's indices stay the reference's); the density network is then evaluated block by
// block of B samples ALONG the ray, on the rays that are still alive:
//   ftb_gather     alive rays' records and their depths of block b -> compact [n_alive, 13] / [n_alive, B] arrays (the density forward's
//                  rays form), and the NEXT block's alive counter set to zero
//   density forward on the compact arrays (live count on the device: no host round trip)
//   ftb_composite  one wave per alive ray: alpha, transmittance and weights of the block exactly as render_ray forms them (same noise
//                  draw per (ray, sample)), accumulated into per-ray depth / opacity sums; a ray whose transmittance stays >= 2^-24
//                  appends itself to the next block's list (order arbitrary: rays are independent)
// On a trained map about half of a scan's 134 M samples lie behind the first surface (profiles/r06_render_dead.txt).
// ------------------------------------------------------------------------------------------------
#define FTB_T_MIN 5.9604644775390625e-08f          /* 2^-24 */

__global__ void __launch_bounds__(RENDER_BLOCK)
ftb_gather_kernel(const float* __restrict__ rays, const float* __restrict__ z, int S, const int32_t* __restrict__ idx,
                  const int32_t* __restrict__ n_alive, int cap, int b0, int B, float* __restrict__ rays_c, float* __restrict__ z_c,
                  int32_t* __restrict__ next_count) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *next_count = 0;
    const int lane = threadIdx.x & 63;
    const int n = min(*n_alive, cap);
    for (int j = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6); j < n; j += gridDim.x * RAYS_PER_BLOCK) {      // one wave per alive ray
        const int ray = idx[j];
        if (lane < LNR_RAY_STRIDE) rays_c[(size_t)j * LNR_RAY_STRIDE + lane] = rays[(size_t)ray * LNR_RAY_STRIDE + lane];
        const float* src = z + (size_t)ray * S + b0;
        float* dst = z_c + (size_t)j * B;
        for (int i = 4 * lane; i < B; i += 256) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
    }
}

__global__ void __launch_bounds__(RENDER_BLOCK)
ftb_composite_kernel(const float* __restrict__ sigma_c, const float* __restrict__ z, const float* __restrict__ rays, int S,
                     const int32_t* __restrict__ idx, const int32_t* __restrict__ n_alive, int cap, int b0, const float* __restrict__ noise,
                     float noise_std, uint64_t seed, float* __restrict__ T_ray, float* __restrict__ depth_acc, float* __restrict__ opac_acc,
                     int32_t* __restrict__ next_idx, int32_t* __restrict__ next_count, int last) {
    constexpr int C = 4, B = 64 * C;                              // 256 samples per block, four per lane
    const int lane = threadIdx.x & 63;
    const int n = min(*n_alive, cap);
    for (int j = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6); j < n; j += gridDim.x * RAYS_PER_BLOCK) {
        const int ray = idx[j];
        const float* rec = rays + (size_t)ray * LNR_RAY_STRIDE;
        const float dnorm = sqrtf(rec[3] * rec[3] + rec[4] * rec[4] + rec[5] * rec[5]);
        const int i0 = b0 + C * lane;
        const float4 zv = *reinterpret_cast<const float4*>(z + (size_t)ray * S + i0);
        const float4 sv = *reinterpret_cast<const float4*>(sigma_c + (size_t)j * B + C * lane);
        const float zl[C] = {zv.x, zv.y, zv.z, zv.w};
        float dens[C] = {sv.x, sv.y, sv.z, sv.w};
        if (noise != nullptr) {                                    // explicit draws [n_rays, S] (already scaled), as lnr_render_forward takes them
            const float4 nv = *reinterpret_cast<const float4*>(noise + (size_t)ray * S + i0);
            dens[0] += nv.x; dens[1] += nv.y; dens[2] += nv.z; dens[3] += nv.w;
        } else if (noise_std > 0.0f) {
#pragma unroll
            for (int t = 0; t < C; ++t) dens[t] += lnr_rand_normal(seed, (uint64_t)ray, (uint32_t)(i0 + t)) * noise_std;
        }
        // the depth behind my last sample: the next lane's first, or - for the block's last sample - the next block's first
        float z_after = __shfl_down(zl[0], 1, 64);
        if (lane == 63) z_after = (b0 + B < S) ? z[(size_t)ray * S + b0 + B] : 0.0f;
        float e[C], Tl[C], tprod = 1.0f;
#pragma unroll
        for (int t = 0; t < C; ++t) {
            const int i = i0 + t;
            float delta = (i < S - 1) ? ((t + 1 < C) ? zl[(t + 1 < C) ? t + 1 : t] : z_after) - zl[t] : 1e10f;
            delta *= dnorm;
            const float r = dens[t] > 0.0f ? dens[t] : 0.0f;
            e[t] = expf(-delta * r);
            Tl[t] = tprod;
            tprod *= (1.0f - (1.0f - e[t]) + 1e-10f);                 // (1 - alpha + 1e-10 with alpha = 1 - e, as render_ray)
        }
        const float prefix = wave_excl_prod(tprod, lane) * T_ray[ray];
        float o_part = 0.0f, d_part = 0.0f;
#pragma unroll
        for (int t = 0; t < C; ++t) {
            const float w = (1.0f - e[t]) * (Tl[t] * prefix);
            o_part += w;
            d_part += w * zl[t];
        }
        const float o_sum = wave_sum(o_part), d_sum = wave_sum(d_part);
        const float T_out = __shfl(prefix * tprod, 63, 64);
        if (lane == 0) {
            depth_acc[ray] += d_sum;
            opac_acc[ray] += o_sum;
            T_ray[ray] = T_out;
            if (!last && T_out >= FTB_T_MIN) next_idx[atomicAdd(next_count, 1)] = ray;
        }
    }
}

extern "C" int lnr_render_ftb_gather(const float* rays, const float* z, int32_t n_samples, const int32_t* idx, const int32_t* n_alive_dev,
                                     int32_t cap, int32_t b0, int32_t block_samples, float* rays_c, float* z_c, int32_t* next_count_dev,
                                     void* stream) {
    LNR_REQUIRE(rays && z && idx && n_alive_dev && rays_c && z_c && next_count_dev, "lnr_render_ftb_gather: null argument");
    LNR_REQUIRE(cap >= 0 && block_samples > 0 && block_samples % 4 == 0 && b0 >= 0 && b0 % 4 == 0 && b0 + block_samples <= n_samples && n_samples % 4 == 0,
                "lnr_render_ftb_gather: a block must lie inside the ray's samples (multiples of four)");
    if (cap == 0) return LNR_OK;
    const int blocks = lnr_div_up(cap, RAYS_PER_BLOCK);
    hipLaunchKernelGGL(ftb_gather_kernel, dim3(blocks > 4096 ? 4096 : blocks), dim3(RENDER_BLOCK), 0, (hipStream_t)stream, rays, z, n_samples, idx,
                       n_alive_dev, cap, b0, block_samples, rays_c, z_c, next_count_dev);
    LNR_CHECK_LAUNCH("lnr_render_ftb_gather");
    return LNR_OK;
}

extern "C" int lnr_render_ftb_composite(const float* sigma_c, const float* z, const float* rays, int32_t n_samples, const int32_t* idx,
                                        const int32_t* n_alive_dev, int32_t cap, int32_t b0, int32_t block_samples, const float* noise, float noise_std,
                                        uint64_t seed, float* transmittance, float* depth_acc, float* opacity_acc, int32_t* next_idx, int32_t* next_count_dev,
                                        int32_t last, void* stream) {
    LNR_REQUIRE(sigma_c && z && rays && idx && n_alive_dev && transmittance && depth_acc && opacity_acc && next_idx && next_count_dev,
                "lnr_render_ftb_composite: null argument");
    LNR_REQUIRE(block_samples == 256, "lnr_render_ftb_composite: blocks of 256 samples (one wave per ray, four samples per lane)");
    LNR_REQUIRE(cap >= 0 && b0 >= 0 && b0 % 4 == 0 && b0 + block_samples <= n_samples && n_samples % 4 == 0, "lnr_render_ftb_composite: a block must lie inside the ray's samples");
    if (cap == 0) return LNR_OK;
    const int blocks = lnr_div_up(cap, RAYS_PER_BLOCK);
    hipLaunchKernelGGL(ftb_composite_kernel, dim3(blocks > 4096 ? 4096 : blocks), dim3(RENDER_BLOCK), 0, (hipStream_t)stream, sigma_c, z, rays, n_samples,
                       idx, n_alive_dev, cap, b0, noise, noise_std, seed, transmittance, depth_acc, opacity_acc, next_idx, next_count_dev, last);
    LNR_CHECK_LAUNCH("lnr_render_ftb_composite");
    return LNR_OK;
}

extern "C" int lnr_render_backward(const float* sigma, const float* z, const float* rays, int32_t n_rays, const int32_t* n_rays_dev,
                                   int32_t n_samples, const float* noise, float noise_std, uint64_t seed, const float* g_depth,
                                   const float* g_weights, const float* g_opacity, const float* g_variance, float* d_sigma,
                                   float* d_rays, void* stream) {
    LNR_REQUIRE(sigma && z && rays && d_sigma && d_rays && n_rays >= 0 && n_samples >= 2, "lnr_render_backward: bad argument");
    if (n_rays == 0) return LNR_OK;
    const dim3 grid(lnr_div_up(n_rays, RAYS_PER_BLOCK)), block(RENDER_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_C(n_samples, hipLaunchKernelGGL(render_backward_kernel<C>, grid, block, 0, st, sigma, z, rays, n_rays, n_rays_dev, n_samples,
                                             noise, noise_std, seed, g_depth, g_weights, g_opacity, g_variance, d_sigma, d_rays));
    LNR_CHECK_LAUNCH("lnr_render_backward");
    return LNR_OK;
}

extern "C" int lnr_count_opaque(const float* rays, const float* depth_gt, int32_t n_rays, const int32_t* n_rays_dev,
                                const float* far0_dev, int32_t* counts_dev, void* stream) {
    LNR_REQUIRE(counts_dev && n_rays >= 0 && (n_rays == 0 || (rays && depth_gt)), "lnr_count_opaque: bad argument");
    hipLaunchKernelGGL(count_opaque_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rays, depth_gt, n_rays, n_rays_dev, counts_dev,
                       far0_dev);
    LNR_CHECK_LAUNCH("lnr_count_opaque");
    return LNR_OK;
}

extern "C" int lnr_los_loss_fused(const float* sigma, const float* z, const float* rays, const float* depth_gt, int32_t n_rays,
                                  const int32_t* n_rays_dev, int32_t n_samples, const float* noise, float noise_std, uint64_t seed,
                                  float scale, const LnrLossConfig* cfg, const int32_t* counts_dev, const float* far0_dev, float* loss_out,
                                  float* d_sigma, float* d_rays, float* ray_stats, float* weights_out, float* block_partials,
                                  int32_t* poison_dev, int32_t poison_tag, void* stream) {
    LNR_REQUIRE(sigma && z && rays && depth_gt && cfg && counts_dev && loss_out && d_sigma && d_rays, "lnr_los_loss_fused: null argument");
    LNR_REQUIRE(poison_dev == nullptr || block_partials != nullptr, "lnr_los_loss_fused: the failure guard needs block_partials (the deterministic reduction)");
    LNR_REQUIRE(cfg->selection >= 0 && cfg->selection <= 3, "lnr_los_loss_fused: unknown loss selection %d", cfg->selection);
    LNR_REQUIRE(n_rays >= 0 && n_samples >= 2, "lnr_los_loss_fused: bad sizes");
    if (n_rays == 0) return LNR_OK;
    const dim3 grid(lnr_div_up(n_rays, RAYS_PER_BLOCK)), block(RENDER_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_C(n_samples, hipLaunchKernelGGL(los_loss_fused_kernel<C>, grid, block, 0, st, sigma, z, rays, depth_gt, n_rays, n_rays_dev,
                                             n_samples, noise, noise_std, seed, scale, *cfg, counts_dev, loss_out, d_sigma, d_rays,
                                             ray_stats, weights_out, block_partials, far0_dev));
    LNR_CHECK_LAUNCH("lnr_los_loss_fused");
    if (block_partials) {
        hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, block_partials, (int)grid.x, loss_out, poison_dev, poison_tag);
        LNR_CHECK_LAUNCH("lnr_los_loss_fused(reduce)");
    }
    return LNR_OK;
}

extern "C" int lnr_weights_gt(const float* s, const float* g, const float* eps_ray, float eps_scalar, int32_t normalise,
                              int32_t n_rays, int32_t n_samples, float* out, void* stream) {
    LNR_REQUIRE(s && g && out && n_rays >= 0 && n_samples > 0, "lnr_weights_gt: bad argument");
    if (n_rays == 0) return LNR_OK;
    hipLaunchKernelGGL(weights_gt_kernel, dim3(lnr_div_up(n_rays, RAYS_PER_BLOCK)), dim3(RENDER_BLOCK), 0, (hipStream_t)stream, s, g,
                       eps_ray, eps_scalar, normalise, n_rays, n_samples, out);
    LNR_CHECK_LAUNCH("lnr_weights_gt");
    return LNR_OK;
}

extern "C" int lnr_logits_grad(const float* s, const float* g, int32_t n_rays, int32_t n_samples, float margin, float l_free,
                               float l_occ, float* out, void* stream) {
    LNR_REQUIRE(s && g && out && n_rays >= 0 && n_samples > 0, "lnr_logits_grad: bad argument");
    const int64_t total = (int64_t)n_rays * n_samples;
    if (total == 0) return LNR_OK;
    hipLaunchKernelGGL(logits_grad_kernel, dim3(lnr_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, s, g, total, n_samples,
                       margin, l_free, l_occ, out);
    LNR_CHECK_LAUNCH("lnr_logits_grad");
    return LNR_OK;
}

extern "C" int lnr_points_grad_to_rays(const float* d_pts, const float* z, int32_t n_rays, const int32_t* n_rays_dev,
                                       int32_t n_samples, float* d_rays, void* stream) {
    LNR_REQUIRE(d_pts && z && d_rays && n_rays >= 0 && n_samples > 0, "lnr_points_grad_to_rays: bad argument");
    if (n_rays == 0) return LNR_OK;
    hipLaunchKernelGGL(points_grad_to_rays_kernel, dim3(lnr_div_up(n_rays, RAYS_PER_BLOCK)), dim3(RENDER_BLOCK), 0, (hipStream_t)stream,
                       d_pts, z, n_rays, n_rays_dev, n_samples, d_rays);
    LNR_CHECK_LAUNCH("lnr_points_grad_to_rays");
    return LNR_OK;
}


// ------------------------------------------------------------------------------------------------ diagnostics
// lnr_rng_draws(LNR_DRAW_NOISE) (include/loner_hip.h): lnr_rand_normal exactly as render_ray evaluates it
__global__ void noise_draws_kernel(uint64_t seed, int n_rays, int n_per_ray, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_rays * n_per_ray) return;
    out[i] = lnr_rand_normal(seed, (uint64_t)(i / n_per_ray), (uint32_t)(i % n_per_ray));
}

int lnr_render_noise_draws(uint64_t seed, int n_rays, int n_per_ray, float* out, hipStream_t st) {
    const int64_t n = (int64_t)n_rays * n_per_ray;
    hipLaunchKernelGGL(noise_draws_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seed, n_rays, n_per_ray, out);
    LNR_CHECK_LAUNCH("lnr_rng_draws");
    return LNR_OK;
}
