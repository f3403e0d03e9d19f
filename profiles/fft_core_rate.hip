// Row-transform core in isolation: how long does ONE length-4096 transform of the mid pass take per block when nothing
// but the transform runs (no global memory in the loop), at 1, 2, 3 and 4 blocks per CU, and what do its parts cost?
// Variants (timing only; the reduced ones compute garbage):
//   full        fft_regs<4096> as the mid pass uses it (LDS-only barriers) + the sixteen multiply-accumulates of an item
//   no_lds      the three register stages only (no scatter / gather / barriers)
//   no_valu     the two exchanges only (scatter, barriers, gather)
//   no_barrier  everything but the barriers
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I ffsubsync_amd/csrc -o profiles/_bin/fft_core_rate \
//         profiles/fft_core_rate.hip && profiles/_bin/fft_core_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "ffs_fft.h"

using namespace ffsa;

#define CHECK(x)                                                   \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

enum { FULL = 0, NO_LDS, NO_VALU, NO_BARRIER };

template <int VARIANT>
__global__ __launch_bounds__(256) void k_core(const cf* __restrict__ tw, cf* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int L = 4096;
    typedef Shape<L> S;
    const int u = threadIdx.x;
    RowAddr<L> addr(0, u);
    TwRegs<L> twr;
    twr.load(tw, u);
    cf v[16], acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        v[q] = mk(1e-3f * (float)(u + q), 1e-3f * (float)(u - q));
        acc[q] = mk(0.f, 0.f);
    }
    for (int it = 0; it < iters; ++it) {
        if constexpr (VARIANT == FULL) {
            fft_regs<L, RowAddr<L>, true>(v, lds, u, addr, twr);
        } else {
            if constexpr (VARIANT != NO_VALU) stage_first(v);
            if constexpr (VARIANT != NO_LDS) {
                if constexpr (VARIANT != NO_BARRIER) lds_barrier();
                stage_scatter<L, 16, 1>(v, lds, u, addr);
                if constexpr (VARIANT != NO_BARRIER) lds_barrier();
                stage_gather<L>(v, lds, u, addr);
            }
            if constexpr (VARIANT != NO_VALU) stage_compute<L, S::R1, 16>(v, twr.s1);
            if constexpr (VARIANT != NO_LDS) {
                if constexpr (VARIANT != NO_BARRIER) lds_barrier();
                stage_scatter<L, S::R1, 16>(v, lds, u, addr);
                if constexpr (VARIANT != NO_BARRIER) lds_barrier();
                stage_gather<L>(v, lds, u, addr);
            }
            if constexpr (VARIANT != NO_VALU) stage_compute<L, S::R2, 256>(v, twr.s2);
        }
        if constexpr (VARIANT != NO_VALU) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = cmac(acc[q], v[q], twr.s1.w[q % 6]);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(v[q]));
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) out[((size_t)blockIdx.x * 256 + u) * 16 + q] = acc[q] + v[q];
}

template <int VARIANT>
static void run(const char* name, const cf* tw, cf* out, int n_cu, bool last) {
    const size_t lds = (size_t)RowAddr<4096>::ROW_ELEMS * sizeof(cf);
    CHECK(hipFuncSetAttribute((const void*)k_core<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    printf("  \"%s\": {", name);
    const int iters = 400;
    for (int bpc = 1; bpc <= 4; ++bpc) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_core<VARIANT>, dim3(n_cu * bpc), dim3(256), lds, 0, tw, out, iters);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_core<VARIANT>, dim3(n_cu * bpc), dim3(256), lds, 0, tw, out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // nanoseconds per transform per block, and per transform per CU (= the former / blocks per CU)
        printf("\"%d_blocks_per_cu\": [%.0f, %.0f]%s", bpc, ms * 1e6 / iters, ms * 1e6 / iters / bpc, bpc < 4 ? ", " : "");
    }
    printf("}%s\n", last ? "" : ",");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    cf *tw, *out;
    CHECK(hipMalloc(&tw, 1 << 20));
    CHECK(hipMemset(tw, 0x3c, 1 << 20));  // small finite floats: timing only
    CHECK(hipMalloc(&out, (size_t)n_cu * 4 * 256 * 16 * sizeof(cf)));
    printf("{\"cus\": %d, \"unit\": \"[ns per transform per block, ns per transform per CU] for one 4096-point row transform + 16 cmacs\",\n", n_cu);
    run<FULL>("full", tw, out, n_cu, false);
    run<NO_LDS>("no_lds", tw, out, n_cu, false);
    run<NO_VALU>("no_valu", tw, out, n_cu, false);
    run<NO_BARRIER>("no_barrier", tw, out, n_cu, true);
    printf("}\n");
    return 0;
}
