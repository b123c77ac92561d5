// Micro-benchmark: float atomic-add throughput on MI355X for the hash-grid gradient scatter.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_bench.hip -o /tmp/atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF; }

// MODE 0: device scope, one table. MODE 1: workgroup scope into the XCD-private copy. MODE 2: device scope, private copy.
template <int MODE, int PAIR>
__global__ void scatter(float* __restrict__ tab, size_t entries, size_t copy_stride, int per_thread, uint32_t seed, uint32_t mask_local) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float* base = tab;
    if (MODE >= 1) base = tab + (size_t)xcc_id() * copy_stride;
    uint32_t s = hash32(tid * 2654435761u + seed);
    for (int i = 0; i < per_thread; ++i) {
        s = hash32(s + i);
        size_t e = (size_t)(s & mask_local) % entries;
        float* p = base + e * (PAIR ? 2 : 1);
        if (MODE == 1) {
            __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (PAIR) __hip_atomic_fetch_add(p + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            atomicAdd(p, 1.0f);
            if (PAIR) atomicAdd(p + 1, 1.0f);
        }
    }
}

__global__ void census(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

__global__ void sum_all(const float* t, size_t n, double* out) {
    double s = 0; for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += t[i];
    atomicAdd(out, s);
}

template <int MODE, int PAIR>
int run(const char* name, float* tab, size_t entries, size_t stride, uint32_t mask, double* dsum) {
    const int blocks = 2048, threads = 256, per = 128;
    const size_t total_floats = stride * 8;
    CHECK(hipMemset(tab, 0, total_floats * sizeof(float)));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    scatter<MODE, PAIR><<<blocks, threads>>>(tab, entries, stride, 8, 1u, mask);     // warm-up
    CHECK(hipMemset(tab, 0, total_floats * sizeof(float)));
    CHECK(hipDeviceSynchronize());
    hipEventRecord(a);
    scatter<MODE, PAIR><<<blocks, threads>>>(tab, entries, stride, per, 7u, mask);
    hipEventRecord(b);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * threads * per * (PAIR ? 2 : 1);
    CHECK(hipMemset(dsum, 0, 8));
    sum_all<<<1024, 256>>>(tab, total_floats, dsum);
    double h; CHECK(hipMemcpy(&h, dsum, 8, hipMemcpyDeviceToHost));
    printf("%-44s %8.3f ms  %8.2f G atomics/s   sum %s (%.0f / %.0f)\n", name, ms, ops / ms / 1e6, h == ops ? "OK" : "MISMATCH", h, ops);
    return 0;
}

int main() {
    const size_t entries = 3706880;            // default hash grid: entries (x2 floats)
    const size_t stride = entries * 2;         // floats per copy
    float* tab; double* dsum; int* cen;
    CHECK(hipMalloc(&tab, stride * 8 * sizeof(float)));
    CHECK(hipMalloc(&dsum, 8)); CHECK(hipMalloc(&cen, 64 * sizeof(int)));
    census<<<64, 64>>>(cen);
    int h[64]; CHECK(hipMemcpy(h, cen, sizeof(h), hipMemcpyDeviceToHost));
    printf("xcc id of blocks 0..31:"); for (int i = 0; i < 32; ++i) printf(" %d", h[i]); printf("\n");
    run<0, 0>("device scope, random over 14.8 MB", tab, entries, stride, 0xFFFFFFFFu, dsum);
    run<0, 1>("device scope, random pairs over 29.7 MB", tab, entries, stride, 0xFFFFFFFFu, dsum);
    run<2, 1>("device scope, XCD-private copy, pairs", tab, entries, stride, 0xFFFFFFFFu, dsum);
    run<1, 0>("workgroup scope, XCD-private copy", tab, entries, stride, 0xFFFFFFFFu, dsum);
    run<1, 1>("workgroup scope, XCD-private copy, pairs", tab, entries, stride, 0xFFFFFFFFu, dsum);
    run<0, 1>("device scope, hot 4096 entries, pairs", tab, 4096, stride, 0xFFFu, dsum);
    run<1, 1>("workgroup scope, hot 4096 entries, pairs", tab, 4096, stride, 0xFFFu, dsum);
    run<0, 1>("device scope, 262144 entries (2 MB), pairs", tab, 262144, stride, 0x3FFFFu, dsum);
    run<1, 1>("workgroup scope, 262144 entries, pairs", tab, 262144, stride, 0x3FFFFu, dsum);
    return 0;
}
