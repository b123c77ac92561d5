// Optimiser steps (gfx950): dense Adam over the flat parameter vector and the occupancy-grid
// pseudo-gradient step.
//
// Replaces  torch.optim.Adam(...).step() / zero_grad   src/mapping/optimizer.py:257-269,376-380
//           Optimizer._step_occupancy_grid             src/mapping/optimizer.py:598-609
//           get_logits_grad                            src/models/losses.py:54-62
//
// Adam is a pure HBM stream: per parameter read p,g,m,v and write p,m,v (+g=0) = 32 B; 16-byte
// vector accesses, grid-stride, one pass.
#include "lnr_common.h"

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float lr_over_c1, float b1, float b2, float eps,
                                         float inv_c2, float grad_scale) {
    const float gg = g * grad_scale;
    m = m + (gg - m) * (1.0f - b1);                  // exp_avg.lerp_(grad, 1-beta1)
    v = v * b2 + (1.0f - b2) * gg * gg;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    const float denom = sqrtf(v) * inv_c2 + eps;     // sqrt(v)/sqrt(bias_correction2) + eps
    p = p - lr_over_c1 * (m / denom);
}

__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
            float lr_over_c1, float b1, float b2, float eps, float inv_c2, float grad_scale, int zero_grad, const int32_t* __restrict__ poison) {
    // failure guard (see lnr_los_loss_fused / lnr_pose_backward): once the run is marked failed no parameter moves any more,
    // as in the reference, where the exception leaves optimizer.step() unreached (optimizer.py:368-376)
    if (poison != nullptr && *poison != 0) return;
    const int64_t n4 = n / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adam_one(pp.x, gg.x, mm.x, vv.x, lr_over_c1, b1, b2, eps, inv_c2, grad_scale);
        adam_one(pp.y, gg.y, mm.y, vv.y, lr_over_c1, b1, b2, eps, inv_c2, grad_scale);
        adam_one(pp.z, gg.z, mm.z, vv.z, lr_over_c1, b1, b2, eps, inv_c2, grad_scale);
        adam_one(pp.w, gg.w, mm.w, vv.w, lr_over_c1, b1, b2, eps, inv_c2, grad_scale);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (zero_grad) g4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    // tail (n not a multiple of 4)
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        adam_one(pp, gg, mm, vv, lr_over_c1, b1, b2, eps, inv_c2, grad_scale);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (zero_grad) g[i] = 0.0f;
    }
}

extern "C" int lnr_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t count, float lr, float beta1,
                             float beta2, float eps, int32_t step, float grad_scale, int32_t zero_grad, const int32_t* poison_dev, void* stream) {
    LNR_REQUIRE(params && grads && exp_avg && exp_avg_sq, "lnr_adam_step: null argument");
    LNR_REQUIRE(count >= 0 && step >= 1, "lnr_adam_step: bad count/step");
    LNR_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15u) == 0, "lnr_adam_step: buffers must be 16-byte aligned");
    if (count == 0) return LNR_OK;
    const double c1 = 1.0 - pow((double)beta1, (double)step);
    const double c2 = sqrt(1.0 - pow((double)beta2, (double)step));
    int64_t blocks = (count / 4 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, count,
                       (float)((double)lr / c1), beta1, beta2, eps, (float)(1.0 / c2), grad_scale, zero_grad, poison_dev);
    LNR_CHECK_LAUNCH("lnr_adam_step");
    return LNR_OK;
}

// ------------------------------------------------------------------------------------------------
// occupancy grid step.  A wave walks 64 consecutive samples of a ray; at V=100 about 20 consecutive
// samples fall into the same voxel, and a float atomic costs one L2 transaction per touched 64-byte line
// per instruction (profiles/r01_scatter_transactions.txt), so each of the 8 corner contributions is first
// summed over runs of equal voxel index with segmented shuffles and only run heads issue atomics.
// The pseudo-gradient is 0 more than `margin` behind the surface, so most runs are skipped altogether.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
occ_grid_step_kernel(float* __restrict__ grid, int V, const float* __restrict__ rays, const float* __restrict__ z,
                     const float* __restrict__ depth_gt, int n_rays, const int32_t* __restrict__ n_rays_dev, int S, float scale,
                     float lr, float margin, float l_free, float l_occ, long long* __restrict__ grad_acc) {
    const int lane = threadIdx.x & 63;
    const int64_t total = (int64_t)lnr_live_rays(n_rays, n_rays_dev) * S;
    const int64_t n_chunks = (total + 63) / 64;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float fV = (float)V;
    for (int64_t chunk = wave_id; chunk < n_chunks; chunk += n_waves) {
        const int64_t m = chunk * 64 + lane;
        const bool live = m < total;
        const int64_t mm = live ? m : total - 1;
        const int ray = (int)(mm / S);
        const float zv = z[mm];
        const float x = zv * scale - depth_gt[ray] * scale;
        float gval = 0.0f;
        if (-x - margin > 0.0f) gval = l_free;
        else if (x + margin > 0.0f && margin - x > 0.0f) gval = -l_occ;
        if (!live) gval = 0.0f;
        if (__ballot(gval != 0.0f) == 0ull) continue;
        const float* r = rays + (size_t)ray * LNR_RAY_STRIDE;
        const float px = lnr_add_rn(r[0], lnr_mul_rn(r[3], zv)), py = lnr_add_rn(r[1], lnr_mul_rn(r[4], zv)), pz = lnr_add_rn(r[2], lnr_mul_rn(r[5], zv));
        const float ix = ((px + 1.0f) * fV - 1.0f) * 0.5f, iy = ((py + 1.0f) * fV - 1.0f) * 0.5f, iz = ((pz + 1.0f) * fV - 1.0f) * 0.5f;
        const float x0 = floorf(ix), y0 = floorf(iy), z0 = floorf(iz);
        const float fx = ix - x0, fy = iy - y0, fz = iz - z0;
        // run structure: consecutive lanes with the same base voxel AND the same ray
        const int cell = ((int)z0 * 1024 + (int)y0) * 1024 + (int)x0;
        const int prev_cell = __shfl_up(cell, 1, 64), prev_ray = __shfl_up(ray, 1, 64);
        const bool head = (lane == 0) || (prev_cell != cell) || (prev_ray != ray);
        int seg = head ? 1 : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(seg, o, 64); if (lane >= o) seg += t; }
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const float xc = x0 + (float)(corner & 1), yc = y0 + (float)((corner >> 1) & 1), zc = z0 + (float)((corner >> 2) & 1);
            const bool inside = !(xc < 0.0f || xc >= fV || yc < 0.0f || yc >= fV || zc < 0.0f || zc >= fV);
            const float w = ((corner & 1) ? fx : 1.0f - fx) * ((corner & 2) ? fy : 1.0f - fy) * ((corner & 4) ? fz : 1.0f - fz);
            float v = inside ? gval * w : 0.0f;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int s2 = __shfl_down(seg, o, 64);
                const float t = __shfl_down(v, o, 64);
                if (lane + o < 64 && s2 == seg) v += t;
            }
            if (head && inside && v != 0.0f) {
                const size_t idx = ((size_t)(int)zc * V + (int)yc) * V + (int)xc;
                // 64-bit fixed point (2^-42): integer atomics add exactly, so the step does not depend on the order of the waves
                if (grad_acc) atomicAdd(reinterpret_cast<unsigned long long*>(grad_acc) + idx, (unsigned long long)__float2ll_rn(v * 4398046511104.0f));
                else atomicAdd(grid + idx, -lr * v);
            }
        }
    }
}

__global__ void occ_grid_apply_kernel(float* __restrict__ grid, long long* __restrict__ grad, int64_t n, float lr, int zero_grad,
                                      const int32_t* __restrict__ poison) {
    if (poison != nullptr && *poison != 0) return;          // failed run: the grid stays as it was (optimizer.py:382-384 is never reached)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long q = grad[i];
        if (q != 0ll) {
            grid[i] = grid[i] - lr * (float)((double)q * (1.0 / 4398046511104.0));
            if (zero_grad) grad[i] = 0ll;
        }
    }
}

extern "C" int lnr_occ_grid_step(float* grid, int32_t V, const float* rays, const float* z, const float* depth_gt, int32_t n_rays,
                                 const int32_t* n_rays_dev, int32_t n_samples, float scale, float lr, float margin, float l_free,
                                 float l_occ, int64_t* grad_acc, void* stream) {
    LNR_REQUIRE(grid && rays && z && depth_gt && V > 0 && n_rays >= 0 && n_samples > 0, "lnr_occ_grid_step: bad argument");
    const int64_t total = (int64_t)n_rays * n_samples;
    if (total == 0) return LNR_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(occ_grid_step_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, grid, V, rays, z, depth_gt, n_rays,
                       n_rays_dev, n_samples, scale, lr, margin, l_free, l_occ, reinterpret_cast<long long*>(grad_acc));
    LNR_CHECK_LAUNCH("lnr_occ_grid_step");
    return LNR_OK;
}

extern "C" int lnr_occ_grid_apply(float* grid, int64_t* grad_acc, int64_t count, float lr, int32_t zero_grad, const int32_t* poison_dev, void* stream) {
    LNR_REQUIRE(grid && grad_acc && count >= 0, "lnr_occ_grid_apply: bad argument");
    if (count == 0) return LNR_OK;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(occ_grid_apply_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, grid, reinterpret_cast<long long*>(grad_acc), count, lr, zero_grad, poison_dev);
    LNR_CHECK_LAUNCH("lnr_occ_grid_apply");
    return LNR_OK;
}
