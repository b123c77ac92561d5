// 6-vector pose [t, axis-angle] -> [R|t] and its backward, batched over the keyframes of a window (gfx950).
//
// Replaces  tensor_to_transform  src/common/pose_utils.py:288-302  (pytorch3d axis_angle_to_matrix: axis-angle ->
// unit quaternion with the small-angle series below 1e-6 rad -> matrix) and its autograd backward, i.e. the last
// step of the pose-Jacobian tail of loss.backward() (src/mapping/optimizer.py:366).  A window has <= 8 poses:
// one thread per pose; the point is to replace ~100 tiny framework kernels per iteration by two launches.
#include "lnr_common.h"

struct QuatParts { float w, x, y, z, k, theta, half; bool small; };

__device__ __forceinline__ QuatParts pose_quat(const float* __restrict__ p) {
    QuatParts q;
    const float a = p[3], b = p[4], c = p[5];
    q.theta = sqrtf(a * a + b * b + c * c);
    q.half = 0.5f * q.theta;
    q.small = fabsf(q.theta) < 1e-6f;
    q.k = q.small ? 0.5f - q.theta * q.theta / 48.0f : sinf(q.half) / q.theta;
    q.w = cosf(q.half); q.x = a * q.k; q.y = b * q.k; q.z = c * q.k;
    return q;
}

__global__ void pose_forward_kernel(const float* __restrict__ pose6, int n, float* __restrict__ T12) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = pose6 + 6 * i;
    const QuatParts q = pose_quat(p);
    const float w = q.w, x = q.x, y = q.y, z = q.z;
    const float s2 = 2.0f / (w * w + x * x + y * y + z * z);
    float* T = T12 + 12 * i;
    T[0] = 1.0f - s2 * (y * y + z * z); T[1] = s2 * (x * y - z * w);        T[2] = s2 * (x * z + y * w);        T[3] = p[0];
    T[4] = s2 * (x * y + z * w);        T[5] = 1.0f - s2 * (x * x + z * z); T[6] = s2 * (y * z - x * w);        T[7] = p[1];
    T[8] = s2 * (x * z - y * w);        T[9] = s2 * (y * z + x * w);        T[10] = 1.0f - s2 * (x * x + y * y); T[11] = p[2];
}

// d_pose6[i] = mask[i] * J^T dT12[i]   (mask nullable: 1 = pose is optimised, 0 = fixed / anchored: gradient exactly 0)
// poison (nullable, int32[2]): the reference raises "invalid gradient in pose" / "invalid pose tensor" after backward and BEFORE
// the optimiser step of the same iteration (optimizer.py:368-374); here the first non-finite pose gradient (of an optimised pose)
// or pose tensor marks the word, and the steps that follow read it and do nothing.
__global__ void pose_backward_kernel(const float* __restrict__ pose6, const float* __restrict__ dT12, const uint8_t* __restrict__ mask,
                                     int n, float* __restrict__ d_pose6, int accumulate, int32_t* __restrict__ poison, int32_t poison_tag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = pose6 + 6 * i;
    const float* G = dT12 + 12 * i;
    const QuatParts q = pose_quat(p);
    const float w = q.w, x = q.x, y = q.y, z = q.z;
    const float nn = w * w + x * x + y * y + z * z;
    const float s2 = 2.0f / nn;
    const float G00 = G[0], G01 = G[1], G02 = G[2], G10 = G[4], G11 = G[5], G12 = G[6], G20 = G[8], G21 = G[9], G22 = G[10];
    const float d_s2 = -G00 * (y * y + z * z) + G01 * (x * y - z * w) + G02 * (x * z + y * w) + G10 * (x * y + z * w) - G11 * (x * x + z * z) +
                       G12 * (y * z - x * w) + G20 * (x * z - y * w) + G21 * (y * z + x * w) - G22 * (x * x + y * y);
    float dw = s2 * (-z * G01 + y * G02 + z * G10 - x * G12 - y * G20 + x * G21);
    float dx = s2 * (y * G01 + z * G02 + y * G10 - 2.0f * x * G11 - w * G12 + z * G20 + w * G21 - 2.0f * x * G22);
    float dy = s2 * (-2.0f * y * G00 + x * G01 + w * G02 + x * G10 + z * G12 - w * G20 + z * G21 - 2.0f * y * G22);
    float dz = s2 * (-2.0f * z * G00 - w * G01 + x * G02 + w * G10 - 2.0f * z * G11 + y * G12 + x * G20 + y * G21);
    const float ds2_dq = -0.5f * s2 * s2 * 2.0f;          // d s2 / d q_i = -s2^2 * q_i
    dw += d_s2 * ds2_dq * w; dx += d_s2 * ds2_dq * x; dy += d_s2 * ds2_dq * y; dz += d_s2 * ds2_dq * z;
    // q = (cos h, aa * k(theta))
    const float a = p[3], b = p[4], c = p[5];
    const float dk = q.small ? -q.theta / 24.0f : (0.5f * cosf(q.half) * q.theta - sinf(q.half)) / (q.theta * q.theta);
    const float d_theta = dk * (a * dx + b * dy + c * dz) - 0.5f * sinf(q.half) * dw;
    const float inv_theta = q.theta > 0.0f ? 1.0f / q.theta : 0.0f;       // torch: d|v|/dv = 0 at v = 0
    const float m = mask ? (mask[i] ? 1.0f : 0.0f) : 1.0f;
    float out[6];
    out[0] = G[3]; out[1] = G[7]; out[2] = G[11];
    out[3] = q.k * dx + d_theta * a * inv_theta;
    out[4] = q.k * dy + d_theta * b * inv_theta;
    out[5] = q.k * dz + d_theta * c * inv_theta;
    float* o = d_pose6 + 6 * i;
    bool bad_grad = false, bad_pose = false;
    for (int j = 0; j < 6; ++j) {
        const float v = (accumulate ? o[j] : 0.0f) + (m != 0.0f ? out[j] : 0.0f);
        o[j] = v;
        bad_grad |= !isfinite(v);
        bad_pose |= !isfinite(p[j]);
    }
    if (poison != nullptr && (bad_grad || bad_pose) && atomicCAS(poison, 0, bad_grad ? LNR_POISON_POSE_GRAD : LNR_POISON_POSE) == 0)
        poison[1] = poison_tag;
}

extern "C" int lnr_pose_forward(const float* pose6, int32_t n, float* transforms, void* stream) {
    LNR_REQUIRE(pose6 && transforms && n >= 0, "lnr_pose_forward: bad argument");
    if (n == 0) return LNR_OK;
    hipLaunchKernelGGL(pose_forward_kernel, dim3(lnr_div_up(n, 64)), dim3(64), 0, (hipStream_t)stream, pose6, n, transforms);
    LNR_CHECK_LAUNCH("lnr_pose_forward");
    return LNR_OK;
}

extern "C" int lnr_pose_backward(const float* pose6, const float* d_transforms, const uint8_t* mask, int32_t n, float* d_pose6,
                                 int32_t accumulate, int32_t* poison_dev, int32_t poison_tag, void* stream) {
    LNR_REQUIRE(pose6 && d_transforms && d_pose6 && n >= 0, "lnr_pose_backward: bad argument");
    if (n == 0) return LNR_OK;
    hipLaunchKernelGGL(pose_backward_kernel, dim3(lnr_div_up(n, 64)), dim3(64), 0, (hipStream_t)stream, pose6, d_transforms, mask, n,
                       d_pose6, accumulate, poison_dev, poison_tag);
    LNR_CHECK_LAUNCH("lnr_pose_backward");
    return LNR_OK;
}
