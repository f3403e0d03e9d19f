// VALU issue-rate / latency micro-benchmark for gfx950: cycles per wave64 instruction for the packed-FP32 ops the FFT
// kernels are made of, next to their scalar-lane forms, at 1, 2 and 4 waves per SIMD, as independent streams
// (16 accumulators) and as one dependent chain.
//
//   hipcc --offload-arch=gfx950 -O3 -o profiles/_bin/valu_rates profiles/valu_rates.hip && profiles/_bin/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                   \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

enum { FMA = 0, PK_FMA, ADD, PK_ADD, PK_MUL, MOV64, PK_FMA_DEP, FMA_DEP, PK_ADD_DEP, PK_ADD_OPSEL, NKIND };
static const char* kNames[NKIND] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_pk_add_f32", "v_pk_mul_f32",
                                    "v_mov_b64", "v_pk_fma_f32 (dependent chain)", "v_fma_f32 (dependent chain)",
                                    "v_pk_add_f32 (dependent chain)", "v_pk_add_f32 op_sel/neg (add_negi)"};

template <int KIND>
__global__ void k(long long* cycles, float* sink, int iters) {
    f2 a[16];
    const f2 b = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (f2){(float)threadIdx.x * 1e-3f + i, (float)i};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define S_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
#define S_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define S_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
#define S_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define S_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define S_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
#define S_PKFMA_DEP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
#define S_FMA_DEP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0].x) : "v"(b.x), "v"(c.x));
#define S_PKADD_DEP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
#define S_PKADD_OPSEL(i) \
    asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
        if (KIND == FMA) { REP16(S_FMA) REP16(S_FMA) REP16(S_FMA) REP16(S_FMA) }
        if (KIND == PK_FMA) { REP16(S_PKFMA) REP16(S_PKFMA) REP16(S_PKFMA) REP16(S_PKFMA) }
        if (KIND == ADD) { REP16(S_ADD) REP16(S_ADD) REP16(S_ADD) REP16(S_ADD) }
        if (KIND == PK_ADD) { REP16(S_PKADD) REP16(S_PKADD) REP16(S_PKADD) REP16(S_PKADD) }
        if (KIND == PK_MUL) { REP16(S_PKMUL) REP16(S_PKMUL) REP16(S_PKMUL) REP16(S_PKMUL) }
        if (KIND == MOV64) { REP16(S_MOV64) REP16(S_MOV64) REP16(S_MOV64) REP16(S_MOV64) }
        if (KIND == PK_FMA_DEP) { REP16(S_PKFMA_DEP) REP16(S_PKFMA_DEP) REP16(S_PKFMA_DEP) REP16(S_PKFMA_DEP) }
        if (KIND == FMA_DEP) { REP16(S_FMA_DEP) REP16(S_FMA_DEP) REP16(S_FMA_DEP) REP16(S_FMA_DEP) }
        if (KIND == PK_ADD_DEP) { REP16(S_PKADD_DEP) REP16(S_PKADD_DEP) REP16(S_PKADD_DEP) REP16(S_PKADD_DEP) }
        if (KIND == PK_ADD_OPSEL) { REP16(S_PKADD_OPSEL) REP16(S_PKADD_OPSEL) REP16(S_PKADD_OPSEL) REP16(S_PKADD_OPSEL) }
    }
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    sink[threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}

template <int KIND>
static void run(long long* d_cyc, float* d_sink) {
    const int iters = 2000;
    printf("  \"%s\": {", kNames[KIND]);
    const int wps[3] = {1, 2, 4};
    for (int j = 0; j < 3; ++j) {
        const int threads = 256 * wps[j];  // one block on one CU: wps[j] waves per SIMD
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d_cyc, d_sink, iters);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, d_cyc, d_sink, 10 * iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        long long cyc = 0;
        CHECK(hipMemcpy(&cyc, d_cyc, sizeof(cyc), hipMemcpyDeviceToHost));
        // [counter ticks, nanoseconds by HIP events] per instruction per resident wave of a SIMD
        const double n = (double)(10 * iters) * 64.0 * wps[j];
        printf("\"%d_waves_per_simd\": [%.2f, %.3f]%s", wps[j], (double)cyc / n, ms * 1e6 / n, j < 2 ? ", " : "");
    }
    printf("}%s\n", KIND + 1 < NKIND ? "," : "");
}

int main() {
    long long* d_cyc;
    float* d_sink;
    CHECK(hipMalloc(&d_cyc, 64));
    CHECK(hipMalloc(&d_sink, 4096 * 4));
    int clk = 0;
    CHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    int wall = 0;
    CHECK(hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0));
    printf("{\"clock_khz\": %d, \"wall_clock_khz\": %d, \"unit\": \"counter ticks per instruction per resident wave of a SIMD "
           "(64 instructions x iters per wave)\",\n", clk, wall);
    run<FMA>(d_cyc, d_sink);
    run<PK_FMA>(d_cyc, d_sink);
    run<ADD>(d_cyc, d_sink);
    run<PK_ADD>(d_cyc, d_sink);
    run<PK_MUL>(d_cyc, d_sink);
    run<MOV64>(d_cyc, d_sink);
    run<PK_FMA_DEP>(d_cyc, d_sink);
    run<FMA_DEP>(d_cyc, d_sink);
    run<PK_ADD_DEP>(d_cyc, d_sink);
    run<PK_ADD_OPSEL>(d_cyc, d_sink);
    printf("}\n");
    return 0;
}
