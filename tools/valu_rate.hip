// Development micro-benchmark (round 6): what ONE wave64 VALU instruction costs a SIMD on gfx950, and how good v_sin_f32 / v_cos_f32 are.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate.bin tools/valu_rate.hip && tools/valu_rate.bin
// Why: the SQ counters give the table-gradient partition 177 M VALU wave-instructions per launch; whether that is 72 % or 36 % of the
// SIMDs' issue capacity depends on whether such an instruction occupies the SIMD for 4 cycles or 2 (profiles/r06_valu_rate.txt).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { OP_FMA = 0, OP_ADD_U32, OP_XOR, OP_FMA_DPP, OP_SIN, OP_CVT_PK, OP_MUL_LO, OP_FMA64, OP_PK_FMA, OP_CNDMASK,
       OP_CNDMASK_SET, OP_CNDMASK_SGPR, OP_CNDMASK_3OP, OP_CNDMASK_MIX, OP_CMP, OP_CMP_SGPR, OP_BFI, OP_MOV, OP_MOV_DPP, OP_READLANE, OP_WRITELANE, OP_MAX, OP_CMP_CND, OP_MUL, OP_CMP_4CND, OP_SCMP_CND, OP_CMP_FMA_CND };

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(float* out, unsigned long long* cyc, int iters, float seed) {
    float a[8];
    unsigned u[8];
    double d[4];
    for (int i = 0; i < 8; ++i) { a[i] = seed + (float)(threadIdx.x + i); u[i] = (unsigned)threadIdx.x * 7u + (unsigned)i; }
    for (int i = 0; i < 4; ++i) d[i] = (double)a[i];
    const float b = seed * 0.5f, c = seed * 0.25f;
    unsigned long long msk = 0x5555AAAA0F0FF0F0ull + (unsigned long long)iters, msk2 = 0;
    int sr = iters;
    if (OP == OP_CNDMASK_SET) asm volatile("s_mov_b64 vcc, %0" :: "s"(msk) : "vcc");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(it));
                if (OP == OP_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(it));
                if (OP == OP_FMA_DPP) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
                if (OP == OP_CVT_PK) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(it));
                if (OP == OP_FMA64) { if (i < 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"((double)b), "v"((double)c)); }
                if (OP == OP_PK_FMA) { if (i < 4) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {a[2 * i], a[2 * i + 1]}; f2 bb = {b, b}, cc = {c, c};
                                                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(bb), "v"(cc)); a[2 * i] = v.x; a[2 * i + 1] = v.y; } }
                if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                if (OP == OP_CNDMASK_SET) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));          // vcc initialised in front of the loop
                if (OP == OP_CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(msk));
                if (OP == OP_CNDMASK_3OP) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b), "v"(c), "s"(msk));
                if (OP == OP_CNDMASK_MIX) { if ((i & 3) == 0) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(msk)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
                if (OP == OP_CMP) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
                if (OP == OP_CMP_SGPR) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(msk2) : "v"(a[i]), "v"(b));
                if (OP == OP_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(it), "v"(u[(i + 1) & 7]));
                if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
                if (OP == OP_MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == OP_READLANE) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sr) : "v"(a[i]));
                if (OP == OP_WRITELANE) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(a[i]) : "s"(sr));
                if (OP == OP_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == OP_CMP_CND) { asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(c), "v"(b) : "vcc"); }
                if (OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                // one compare, then four selects on its vcc (what a compiler emits for `cond ? a[k] : b` over several values)
                if (OP == OP_CMP_4CND) { if ((i & 3) == 0) asm volatile("v_cmp_lt_f32 vcc, %4, %5\n\tv_cndmask_b32 %0, %0, %5, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc\n\tv_cndmask_b32 %2, %2, %5, vcc\n\tv_cndmask_b32 %3, %3, %5, vcc"
                                                                      : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]) : "v"(c), "v"(b) : "vcc"); }
                // vcc written by the SCALAR unit (s_mov), then selects
                if (OP == OP_SCMP_CND) { if ((i & 3) == 0) asm volatile("s_mov_b64 vcc, %4\n\tv_cndmask_b32 %0, %0, %5, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc\n\tv_cndmask_b32 %2, %2, %5, vcc\n\tv_cndmask_b32 %3, %3, %5, vcc"
                                                                      : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]) : "s"(msk), "v"(b) : "vcc"); }
                // compare, three unrelated VALU instructions, then one select
                if (OP == OP_CMP_FMA_CND) { if ((i & 3) == 0) asm volatile("v_cmp_lt_f32 vcc, %4, %5\n\tv_fma_f32 %1, %1, %5, %4\n\tv_fma_f32 %2, %2, %5, %4\n\tv_fma_f32 %3, %3, %5, %4\n\tv_cndmask_b32 %0, %0, %5, vcc"
                                                                        : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]) : "v"(c), "v"(b) : "vcc"); }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i];
    for (int i = 0; i < 4; ++i) s += (float)d[i];
    s += (float)(msk2 & 1ull) + (float)sr;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);
}

template <int OP>
static void run(const char* name, int per_iter, int waves_per_simd) {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const int blocks = cus * waves_per_simd;          // 256 threads = one wave per SIMD
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMalloc(&cyc, 8));
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(cyc, 0, 8));
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h;
    CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    const double n_inst = (double)iters * per_iter;                 // per wave
    // s_memtime counts at a constant 100 MHz reference on some parts: report both the tick count and the wall-clock figure
    const double ns = 1e6 * ms / (n_inst * waves_per_simd);
    printf("%-34s %d wave(s)/SIMD  %8.3f ms  %7.3f ns per wave-instruction and SIMD  (= %5.2f cycles at 2.4 GHz, %5.2f at 2.1 GHz); per wave: %.2f ns\n",
           name, waves_per_simd, ms, ns, 2.4 * ns, 2.1 * ns, 1e6 * ms / n_inst);
    (void)h;
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

__global__ void sin_accuracy_kernel(const float* __restrict__ x, float* __restrict__ s, float* __restrict__ c, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sv, cv;
    const float r = x[i];
    asm volatile("v_sin_f32 %0, %1" : "=v"(sv) : "v"(r));
    asm volatile("v_cos_f32 %0, %1" : "=v"(cv) : "v"(r));
    s[i] = sv; c[i] = cv;
}

int main() {
    for (int w : {1, 4, 8}) {
        run<OP_FMA>("v_fma_f32", 32, w);
        run<OP_ADD_U32>("v_add_u32", 32, w);
        run<OP_XOR>("v_xor_b32", 32, w);
        run<OP_CNDMASK>("v_cndmask_b32", 32, w);
        run<OP_FMA_DPP>("v_fmac_f32_dpp row_shr:1", 32, w);
        run<OP_CVT_PK>("v_cvt_pkrtz_f16_f32", 32, w);
        run<OP_MUL_LO>("v_mul_lo_u32", 32, w);
        run<OP_SIN>("v_sin_f32", 32, w);
        run<OP_FMA64>("v_fma_f64", 16, w);
        run<OP_PK_FMA>("v_pk_fma_f32", 16, w);
        run<OP_CNDMASK_SET>("v_cndmask_b32 vcc (vcc set)", 32, w);
        run<OP_CNDMASK_SGPR>("v_cndmask_b32_e64 sgpr mask", 32, w);
        run<OP_CNDMASK_3OP>("v_cndmask_b32_e64 d, b, c, sgpr", 32, w);
        run<OP_CNDMASK_MIX>("1 cndmask_e64 : 3 v_fma", 32, w);
        run<OP_CMP>("v_cmp_lt_f32 vcc", 32, w);
        run<OP_CMP_SGPR>("v_cmp_lt_f32_e64 sgpr", 32, w);
        run<OP_CMP_CND>("v_cmp + v_cndmask (pair = 2)", 64, w);
        run<OP_BFI>("v_bfi_b32", 32, w);
        run<OP_MOV>("v_mov_b32", 32, w);
        run<OP_MOV_DPP>("v_mov_b32_dpp row_shr:1", 32, w);
        run<OP_READLANE>("v_readlane_b32", 32, w);
        run<OP_WRITELANE>("v_writelane_b32", 32, w);
        run<OP_MAX>("v_max_f32", 32, w);
        run<OP_MUL>("v_mul_f32", 32, w);
        run<OP_CMP_4CND>("v_cmp, 4 x v_cndmask vcc (5 per group)", 40, w);
        run<OP_SCMP_CND>("s_mov vcc, 4 x v_cndmask vcc (4 VALU)", 32, w);
        run<OP_CMP_FMA_CND>("v_cmp, 3 v_fma, v_cndmask vcc (5)", 40, w);
    }
    // accuracy of the hardware sine / cosine (argument in revolutions) over [-1, 1]
    const int n = 1 << 22;
    std::vector<float> hx(n), hs(n), hc(n);
    for (int i = 0; i < n; ++i) hx[i] = -1.0f + 2.0f * (float)i / (float)n + 1e-7f * (float)(i % 7);
    float *dx, *dsn, *dcs;
    CHECK(hipMalloc(&dx, n * 4)); CHECK(hipMalloc(&dsn, n * 4)); CHECK(hipMalloc(&dcs, n * 4));
    CHECK(hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sin_accuracy_kernel, dim3(n / 256), dim3(256), 0, 0, dx, dsn, dcs, n);
    CHECK(hipMemcpy(hs.data(), dsn, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hc.data(), dcs, n * 4, hipMemcpyDeviceToHost));
    double es = 0.0, ec = 0.0, es_small = 0.0;
    for (int i = 0; i < n; ++i) {
        const double t = 6.283185307179586476925 * (double)hx[i];
        es = fmax(es, fabs((double)hs[i] - sin(t)));
        ec = fmax(ec, fabs((double)hc[i] - cos(t)));
        if (fabs(hx[i]) < 1e-3) es_small = fmax(es_small, fabs((double)hs[i] - sin(t)) / fmax(fabs(sin(t)), 1e-30));
    }
    printf("v_sin_f32 over [-1, 1] revolutions: max abs error %.3e; v_cos_f32: %.3e; v_sin_f32 relative error for |x| < 1e-3 rev: %.3e\n", es, ec, es_small);
    return 0;
}
