// Feasibility rate of the on-chip (overlap-save) alternative costed in DESIGN.md section 8: ONE 512-thread block per
// CU keeps a 16384-point chunk transform (32 values per thread) and two frequency-domain accumulators (2 x 32 values
// per thread) in registers -- no intermediate in HBM.  Per chunk step:
//   1. synthesise the chunk's 16384 complex inputs (two VALU instructions per value: the cost of the bit decode)
//   2. four 4096-point sub-transforms (x[4j + g]): two teams of 256 threads, two register sets each, fft_regs<4096>
//      as in the mid pass (own row buffer per team, LDS-only barriers)
//   3. radix-4 combine across the four sub-transforms through a 4 x 4096 LDS tile (aliases the row buffers):
//      X[k' + 4096 r] = sum_g W_16384^(g k') W_4^(g r) F_g[k'],  eight k' per thread
//   4. acc_w[i] += X[i] * conj(R_w[i]) for the two lag-window halves; R_w streams from global memory in four groups of
//      sixteen values (VARIANT 1: an 8 MB region, L2 resident; VARIANT 2: a 2 GB region; VARIANT 0: register constants)
// Timing only (tables hold constants: the data is garbage, the instruction and LDS streams are the real ones).
// Budget: DESIGN.md prices a seven-ratio pair at 322 such transforms, i.e. 6.4 us/pair needs <= 5.1 us per chunk step
// per CU; today's three passes take 13.4 us/pair.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I ffsubsync_amd/csrc -o profiles/_bin/onchip_chunk_rate \
//         profiles/onchip_chunk_rate.hip && profiles/_bin/onchip_chunk_rate
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

constexpr int SUB = 4096;                       // sub-transform length
constexpr int SUB_PAD = SUB + SUB / 16;         // one pad element per 16 (as RowAddr)
typedef const __attribute__((address_space(1))) cf* gcf;

template <int VARIANT, bool WITH_FFT, bool WITH_COMBINE, bool WITH_ACC>
__global__ __launch_bounds__(512, 1) void k_chunk_step(const cf* __restrict__ tw, const cf* __restrict__ twc,
                                                       const cf* __restrict__ rspec, size_t rspec_elems,
                                                       cf* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    const int tid = threadIdx.x, team = tid >> 8, u = tid & 255;
    cf* rowbuf = lds + team * RowAddr<SUB>::ROW_ELEMS;
    RowAddr<SUB> addr(0, u);
    TwRegs<SUB> twr;
    twr.load(tw, u);
    cf v[2][16];      // register sets: sub-transform g = 2*team + s holds x[4*(u + 256 q) + g]
    cf acc[2][32];    // two lag-window halves x the thread's 32 output bins
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[w][i] = mk(0.f, 0.f);
    // combine twiddles W_16384^k' for this thread's eight k' = tid + 512 j
    cf w1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w1[j] = twc[tid + 512 * j];
    const size_t chunk_elems = 2 * 16384;  // both windows' spectra of one chunk
    const size_t n_chunks = rspec_elems / chunk_elems;
    size_t chunk = (size_t)blockIdx.x * 7 % n_chunks;
    for (int it = 0; it < iters; ++it) {
        // 1. inputs: two VALU instructions per value
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const unsigned b = (unsigned)(it * 2654435761u) >> ((q + 16 * s) & 31);
                v[s][q] = mk((b & 1u) ? 1.0f : -1.0f, (b & 2u) ? 0.96f : 0.0f);
            }
        // 2. four sub-transforms, two per team one after the other
        if constexpr (WITH_FFT) {
#pragma unroll
            for (int s = 0; s < 2; ++s) fft_regs<SUB, RowAddr<SUB>, true>(v[s], rowbuf, u, addr, twr);
        }
        // 3. radix-4 combine across g through the LDS tile [g][k' + k'/16]
        cf y[8][4];
        if constexpr (WITH_COMBINE) {
            lds_barrier();  // the sub-transforms' last gathers are done (the tile aliases the row buffers)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int k = u + 256 * q;
                    lds[(2 * team + s) * SUB_PAD + k + (k >> 4)] = v[s][q];
                }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = tid + 512 * j;
#pragma unroll
                for (int g = 0; g < 4; ++g) y[j][g] = lds[g * SUB_PAD + k + (k >> 4)];
            }
            lds_barrier();  // the next step's scatter may overwrite the tile
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const cf w2 = cmul(w1[j], w1[j]), w3 = cmul(w2, w1[j]);
                y[j][1] = cmul(y[j][1], w1[j]);
                y[j][2] = cmul(y[j][2], w2);
                y[j][3] = cmul(y[j][3], w3);
                dft4(y[j][0], y[j][1], y[j][2], y[j][3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) y[j][g] = v[g & 1][(2 * j + (g >> 1)) & 15];
        }
        // 4. products with the two reference spectra of this chunk, accumulated
        if constexpr (WITH_ACC) {
            if constexpr (VARIANT == 0) {
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[w][4 * j + r] = cmac(acc[w][4 * j + r], y[j][r], w1[(j + r + w) & 7]);
            } else {
                gcf rs = (gcf)(rspec + chunk * chunk_elems);
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        cf rr[16];  // sixteen values in flight: bins k' + 4096 r for j = 4*half .. 4*half + 3
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                rr[4 * jj + r] = rs[(size_t)w * 16384 + (size_t)(tid + 512 * (4 * half + jj)) + 4096 * r];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int i = 4 * (4 * half + jj) + r;
                                acc[w][i] = cmac(acc[w][i], y[4 * half + jj][r], rr[4 * jj + r]);
                            }
                    }
                chunk = chunk + 1 < n_chunks ? chunk + 1 : 0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(y[j][r]));
        }
    }
    cf total = mk(0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int i = 0; i < 32; ++i) total = cadd(total, acc[w][i]);
    out[(size_t)blockIdx.x * 512 + tid] = total;
}

template <int VARIANT, bool F, bool C, bool A>
static void run(const char* name, const cf* tw, const cf* twc, const cf* rspec, size_t rspec_elems, cf* out, int n_cu, bool last) {
    const size_t lds = (size_t)4 * SUB_PAD * sizeof(cf);
    CHECK(hipFuncSetAttribute((const void*)k_chunk_step<VARIANT, F, C, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int iters = 200;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_chunk_step<VARIANT, F, C, A>), dim3(n_cu), dim3(512), lds, 0, tw, twc, rspec, rspec_elems, out, iters);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chunk_step<VARIANT, F, C, A>), dim3(n_cu), dim3(512), lds, 0, tw, twc, rspec, rspec_elems, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    printf("  \"%s\": {\"us_per_chunk_step_per_cu\": %.3f, \"us_per_pair_at_322_steps\": %.2f}%s\n", name, us, us * 322.0 / n_cu,
           last ? "" : ",");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    cf *tw, *twc, *out, *rspec;
    CHECK(hipMalloc(&tw, 1 << 20));
    CHECK(hipMemset(tw, 0x3c, 1 << 20));  // small finite floats: timing only
    CHECK(hipMalloc(&twc, 4096 * sizeof(cf)));
    CHECK(hipMemset(twc, 0x3c, 4096 * sizeof(cf)));
    CHECK(hipMalloc(&out, (size_t)n_cu * 512 * sizeof(cf)));
    const size_t big = (size_t)1 << 28;  // 2 GB of reference spectra (HBM); the first 8 MB double as the L2-resident set
    CHECK(hipMalloc(&rspec, big * sizeof(cf)));
    CHECK(hipMemset(rspec, 0x3c, big * sizeof(cf)));
    printf("{\"cus\": %d, \"what\": \"one 512-thread block per CU: 16384-point chunk transform (4 x 4096 + radix-4 combine) + two accumulators in registers\",\n", n_cu);
    run<0, true, false, false>("sub_transforms_only", tw, twc, rspec, big, out, n_cu, false);
    run<0, true, true, false>("transform_16k", tw, twc, rspec, big, out, n_cu, false);
    run<0, true, true, true>("step_reference_in_registers", tw, twc, rspec, big, out, n_cu, false);
    run<1, true, true, true>("step_reference_from_l2", tw, twc, rspec, (size_t)1 << 20, out, n_cu, false);
    run<2, true, true, true>("step_reference_from_hbm", tw, twc, rspec, big, out, n_cu, true);
    printf("}\n");
    return 0;
}
