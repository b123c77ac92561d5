// Micro-benchmark 2: how does the memory system price scattered atomics / stores as a function of how many
// lanes of a wave fall into the same 64-byte line?   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// G = lanes per group sharing one random line-aligned block of G*4 bytes
template <int G, int KIND>   // KIND 0: float atomic, 1: 4-byte store, 2: 8-byte store (G counts 8-byte slots), 3: 16-byte store
__global__ void k(float* __restrict__ tab, uint32_t n_blocks_of_16f, int per_thread, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = tid / G, sub = tid % G;
    uint32_t s = hash32(grp * 2654435761u + seed);
    for (int i = 0; i < per_thread; ++i) {
        s = hash32(s + i);
        const size_t line = (size_t)(s % n_blocks_of_16f) * 16;       // 64-byte aligned block of 16 floats
        if (KIND == 0) atomicAdd(tab + line + sub, 1.0f);
        else if (KIND == 1) tab[line + sub] = 1.0f;
        else if (KIND == 2) reinterpret_cast<float2*>(tab + line)[sub] = make_float2(1.0f, 2.0f);
        else reinterpret_cast<float4*>(tab + line)[sub] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
    }
}

__global__ void lds_atomics(float* out, int per_thread, uint32_t seed, int mode) {
    __shared__ float acc[16384];
    __shared__ int cur[512];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) acc[i] = 0;
    for (int i = threadIdx.x; i < 512; i += blockDim.x) cur[i] = 0;
    __syncthreads();
    uint32_t s = hash32((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed);
    int sink = 0;
    for (int i = 0; i < per_thread; ++i) {
        s = hash32(s + i);
        if (mode == 0) atomicAdd(&acc[s & 16383], 1.0f);
        else sink += atomicAdd(&cur[s & 511], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0] + sink + cur[1];
}

template <int G, int KIND>
void run(const char* name, float* tab, uint32_t nb) {
    const int blocks = 2048, threads = 256, per = 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<G, KIND><<<blocks, threads>>>(tab, nb, 4, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<G, KIND><<<blocks, threads>>>(tab, nb, per, 7u);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    const double lane_ops = (double)blocks * threads * per;
    printf("%-52s %8.3f ms  %8.2f G lane-ops/s  %8.2f G line-transactions/s\n", name, ms, lane_ops / ms / 1e6, lane_ops / G / ms / 1e6);
}

int main() {
    const uint32_t nb = 7413760 / 16;          // 29.7 MB table in 64-byte lines
    float* tab; hipMalloc(&tab, (size_t)nb * 64); hipMemset(tab, 0, (size_t)nb * 64);
    run<1, 0>("atomic f32, 1 lane per line", tab, nb);
    run<2, 0>("atomic f32, 2 adjacent lanes per line", tab, nb);
    run<4, 0>("atomic f32, 4 adjacent lanes per line", tab, nb);
    run<8, 0>("atomic f32, 8 adjacent lanes per line", tab, nb);
    run<16, 0>("atomic f32, 16 adjacent lanes per line (full line)", tab, nb);
    run<1, 1>("store 4 B, 1 lane per line", tab, nb);
    run<1, 2>("store 8 B, 1 lane per line", tab, nb);
    run<1, 3>("store 16 B, 1 lane per line", tab, nb);
    run<4, 3>("store 16 B, 4 lanes per line (full line)", tab, nb);
    run<8, 2>("store 8 B, 8 lanes per line (full line)", tab, nb);
    run<16, 1>("store 4 B, 16 lanes per line (full line)", tab, nb);
    float* out; hipMalloc(&out, 4096 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        lds_atomics<<<1024, 1024>>>(out, 8, 1u, mode); hipDeviceSynchronize();
        hipEventRecord(a); lds_atomics<<<1024, 1024>>>(out, 256, 7u, mode); hipEventRecord(b); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-52s %8.3f ms  %8.2f G lane-ops/s\n", mode == 0 ? "LDS ds_add_f32 random over 64 KB" : "LDS ds_add_rtn_u32 random over 512 counters", ms, 1024.0 * 1024 * 256 / ms / 1e6);
    }
    return 0;
}
