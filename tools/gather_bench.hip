// Micro-benchmark 3: what prices a random table gather on MI355X when the table is L2-resident (2 MB, the size of one
// hash-grid level)?  Varies the element width, how many lanes of a wave share a 64-byte line, the cache policy of
// the load, and compares with LDS gathers and LDS atomics.     hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// W = bytes per lane (4, 8, 16); G = adjacent lanes sharing one aligned 64-byte line; U = independent loads in flight
template <int W, int G, int U>
__global__ void __launch_bounds__(256) gather_k(const float* __restrict__ tab, uint32_t n_lines, int iters, uint32_t seed, float* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = tid / G, sub = tid % G;
    uint32_t s = hash32(grp * 2654435761u + seed);
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        uint32_t line[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { s = hash32(s + u + 1); line[u] = (s % n_lines) * 16u; }
        float part[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float* p = tab + line[u] + sub * (W / 4);
            if (W == 4) part[u] = *p;
            else if (W == 8) { const float2 t = *reinterpret_cast<const float2*>(p); part[u] = t.x + t.y; }
            else { const float4 t = *reinterpret_cast<const float4*>(p); part[u] = t.x + t.y + t.z + t.w; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += part[u];
    }
    if (acc == 12345.678f) out[0] = acc;
}

// Is a gather priced per wave instruction or per active lane?  Only every HALF-th lane loads (the others sit the kernel out).
template <int HALF, int U>
__global__ void __launch_bounds__(256) gather_masked_k(const float* __restrict__ tab, uint32_t n_lines, int iters, uint32_t seed, float* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = hash32(tid * 2654435761u + seed);
    float acc = 0.0f;
    if ((tid % HALF) == 0) {
        for (int i = 0; i < iters; ++i) {
            float2 part[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { s = hash32(s + u + 1); part[u] = *reinterpret_cast<const float2*>(tab + (size_t)(s % n_lines) * 16u); }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += part[u].x + part[u].y;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// sc1 / nt variants of the 8-byte gather, through the builtins that lower to those policy bits (relaxed agent-scope atomic
// load = sc1, system scope = sc0 sc1, nontemporal = nt); hand-written asm loads are invisible to the register allocator's
// view of what is still in flight.
template <int POLICY, int U>
__global__ void __launch_bounds__(256) gather8_policy_k(const float* __restrict__ tab, uint32_t n_lines, int iters, uint32_t seed, float* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = hash32(tid * 2654435761u + seed);
    unsigned long long acc = 0ull;
    for (int i = 0; i < iters; ++i) {
        unsigned long long part[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s = hash32(s + u + 1);
            unsigned long long* p = reinterpret_cast<unsigned long long*>(const_cast<float*>(tab) + (size_t)(s % n_lines) * 16u);
            if (POLICY == 1) part[u] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (POLICY == 2) part[u] = __builtin_nontemporal_load(p);
            else if (POLICY == 3) part[u] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else part[u] = *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += part[u];
    }
    if (acc == 0x123456789ull) out[0] = 1.0f;
}

// LDS: MODE 0 random ds_read_b64 from 32 KB; 1 random ds_add_u64 over 64 KB; 2 random ds_add_u32 over 32 KB;
// 3 ds_add_rtn_u32 on 64 wave-private counters; 4 random ds_write_b64 into a wave-private 4 KB buffer; 5 ds_add_f32
template <int MODE>
__global__ void __launch_bounds__(1024) lds_k(float* out, int iters, uint32_t seed) {
    extern __shared__ unsigned long long lds64[];
    uint32_t* lds32 = reinterpret_cast<uint32_t*>(lds64);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds64[i] = 0ull;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    uint32_t s = hash32((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed);
    unsigned long long sink = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = hash32(s + u + 1);
            if (MODE == 0) sink += lds64[s & 4095];
            else if (MODE == 1) atomicAdd(&lds64[s & 8191], (unsigned long long)s);
            else if (MODE == 2) atomicAdd(&lds32[s & 8191], s);
            else if (MODE == 3) sink += atomicAdd(&lds32[wave * 64 + (s & 63)], 1u);
            else if (MODE == 4) lds64[wave * 512 + (s & 511)] = s;
            else atomicAdd(reinterpret_cast<float*>(&lds32[s & 8191]), 1.0f);
        }
    }
    __syncthreads();
    if (sink == 0x1234567ull) out[0] = (float)lds64[0];
    if (threadIdx.x == 0) out[1 + blockIdx.x] = (float)lds64[1];
}

static float timeit(void (*launch)(int), int warm, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(warm); hipDeviceSynchronize();
    hipEventRecord(a); launch(iters); hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

static float* g_tab; static float* g_out; static uint32_t g_lines;
#define BLOCKS 4096
template <int W, int G, int U> void lg(int it) { gather_k<W, G, U><<<BLOCKS, 256>>>(g_tab, g_lines, it, 7u, g_out); }
template <int P, int U> void lp(int it) { gather8_policy_k<P, U><<<BLOCKS, 256>>>(g_tab, g_lines, it, 7u, g_out); }
template <int H, int U> void lm(int it) { gather_masked_k<H, U><<<BLOCKS, 256>>>(g_tab, g_lines, it, 7u, g_out); }
template <int M> void ll(int it) { lds_k<M><<<1024, 1024, 65536>>>(g_out, it, 3u); }

template <int W, int G, int U> void rg(const char* name) {
    const int it = 32;
    const float ms = timeit(lg<W, G, U>, 2, it);
    const double lanes = (double)BLOCKS * 256 * it * U;
    printf("%-64s %7.3f ms %8.1f G lane-gathers/s %8.1f G lines/s  %6.2f lines/clk/CU@2.1GHz\n", name, ms, lanes / ms / 1e6, lanes / G / ms / 1e6,
           lanes / G / ms / 1e6 / (256 * 2.1));
}
template <int H, int U> void rm(const char* name) {
    const int it = 32;
    const float ms = timeit(lm<H, U>, 2, it);
    const double instr = (double)BLOCKS * 4 * it * U;
    printf("%-64s %7.3f ms %8.2f clk per wave instruction and CU @2.1GHz\n", name, ms, ms * 1e-3 * 2.1e9 * 256 / instr);
}
template <int P, int U> void rp(const char* name) {
    const int it = 32;
    const float ms = timeit(lp<P, U>, 2, it);
    const double lanes = (double)BLOCKS * 256 * it * U;
    printf("%-64s %7.3f ms %8.1f G lane-gathers/s\n", name, ms, lanes / ms / 1e6);
}
template <int M> void rl(const char* name) {
    const int it = 64;
    const float ms = timeit(ll<M>, 2, it);
    const double lanes = 1024.0 * 1024 * it * 8;
    printf("%-64s %7.3f ms %8.1f G lane-ops/s   %6.2f lanes/clk/CU@2.1GHz\n", name, ms, lanes / ms / 1e6, lanes / ms / 1e6 / (256 * 2.1));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    for (int pass = 0; pass < 3; ++pass) {
        const size_t bytes = pass == 0 ? (2u << 20) : pass == 1 ? (1u << 20) : (30u << 20);
        g_lines = (uint32_t)(bytes / 64);
        hipMalloc(&g_tab, bytes); hipMemset(g_tab, 0, bytes);
        hipMalloc(&g_out, 8192 * 4);
        printf("---- table %zu KB\n", bytes >> 10);
        rg<8, 1, 8>("8 B, 1 lane/line, 8 in flight");
        rg<8, 1, 4>("8 B, 1 lane/line, 4 in flight");
        rg<4, 1, 8>("4 B, 1 lane/line, 8 in flight");
        rg<16, 1, 8>("16 B, 1 lane/line, 8 in flight");
        rg<8, 2, 8>("8 B, 2 lanes/line");
        rg<8, 4, 8>("8 B, 4 lanes/line");
        rg<8, 8, 8>("8 B, 8 lanes/line (full line)");
        rg<4, 16, 8>("4 B, 16 lanes/line (full line)");
        rm<1, 8>("8 B, all 64 lanes of a wave load");
        rm<2, 8>("8 B, every 2nd lane loads");
        rm<4, 8>("8 B, every 4th lane loads");
        rp<0, 8>("8 B plain (u64)");
        rp<1, 8>("8 B sc1 (agent relaxed atomic load)");
        rp<2, 8>("8 B nt");
        rp<3, 8>("8 B sc0 sc1 (system relaxed atomic load)");
        hipFree(g_tab);
    }
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_k<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_k<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_k<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    printf("---- LDS (1024 blocks x 1024 threads, 64 KB each)\n");
    rl<0>("ds_read_b64 random over 32 KB");
    rl<1>("ds_add_u64 random over 64 KB");
    rl<2>("ds_add_u32 random over 32 KB");
    rl<3>("ds_add_rtn_u32, 64 wave-private counters");
    rl<4>("ds_write_b64 random into wave-private 4 KB");
    rl<5>("ds_add_f32 random over 32 KB");
    return 0;
}
