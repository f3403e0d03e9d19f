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

// ---- experimental exchange / schedule variants (measured here, NOT part of the product header) ----------------------
namespace ffsa {
// The two halves are separate functions because the phase-staggered transform (fft4096_stag) puts a barrier between them.
FFS_DEV void bfly16_tw_front(cf* t, const cf* w /* w1 w2 w3 w4 w8 w12 */) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        t[4 + b] = cmul(t[4 + b], w[3]);
        t[8 + b] = cmul(t[8 + b], w[4]);
        t[12 + b] = cmul(t[12 + b], w[5]);
        dft4(t[b], t[4 + b], t[8 + b], t[12 + b]);
    }
}
FFS_DEV void bfly16_tw_back(cf* t, const cf* w) {
    // slot b + 4*k1 holds A_b[k1]
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        t[4 * k1 + 1] = cmul(t[4 * k1 + 1], w[0]);
        t[4 * k1 + 2] = cmul(t[4 * k1 + 2], w[1]);
        t[4 * k1 + 3] = cmul(t[4 * k1 + 3], w[2]);
    }
    t[5] = cmul_k(t[5], FFS_COS_PI_8, -FFS_SIN_PI_8);
    t[6] = cmul_k(t[6], FFS_SQRT_HALF, -FFS_SQRT_HALF);
    t[7] = cmul_k(t[7], FFS_SIN_PI_8, -FFS_COS_PI_8);
    t[9] = cmul_k(t[9], FFS_SQRT_HALF, -FFS_SQRT_HALF);
    t[10] = cmul_negi(t[10]);
    t[11] = cmul_k(t[11], -FFS_SQRT_HALF, -FFS_SQRT_HALF);
    t[13] = cmul_k(t[13], FFS_SIN_PI_8, -FFS_COS_PI_8);
    t[14] = cmul_k(t[14], -FFS_SQRT_HALF, -FFS_SQRT_HALF);
    t[15] = cmul_k(t[15], -FFS_COS_PI_8, FFS_SIN_PI_8);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(t[4 * k1], t[4 * k1 + 1], t[4 * k1 + 2], t[4 * k1 + 3]);
}
// ---- planar row exchange through ds_write_addtid_b32 -------------------------------------------------------------
// A ds_write_b64 moves three dwords per lane (address + two data) from the VGPRs to the LDS at two cycles each: six
// cycles per wave-instruction, whatever the LDS array could do.  ds_write_addtid_b32 carries ONE dword -- its address
// is M0 + immediate + 4*lane -- and takes two, so a complex value costs four cycles instead of six when its real and
// imaginary parts go to two planes laid out [slot][thread] (the only layout "address = lane" allows).  The gathers then
// read the two planes with ds_read_b32 (two cycles each, like one ds_read_b64 with the two-way conflict the padded
// row buffer has); the plane row strides (258 / 272 dwords) make both gathers conflict-free.  Registers, butterflies and
// barriers are those of fft_regs<4096>; only the sixteen-value exchanges differ.
struct RowPlanar {
    static constexpr int S1 = 258, S2 = 272;          // dwords between the slot rows of exchange 1 / exchange 2
    static constexpr int PLANE = 16 * S2;             // dwords per plane (the larger of the two layouts)
    static constexpr int ROW_DWORDS = 2 * PLANE;      // real plane + imaginary plane
    unsigned m0;      // byte address of this wave's first thread in the buffer: the addtid base
    int g1, g2;       // gather bases (dwords)
    // lds_byte_base: byte address of the buffer in the LDS; u = thread of the transform (0..255), wave-contiguous
    FFS_DEV RowPlanar(unsigned lds_byte_base, int u) {
        m0 = __builtin_amdgcn_readfirstlane(lds_byte_base + 4u * (unsigned)(u & ~63));
        g1 = (u & 15) * S1 + (u >> 4);   // exchange 1: element p = 16*w + r sits at [r][w]; thread u wants p = u + 256 q
        g2 = (u >> 4) * S2 + (u & 15);   // exchange 2: element p = 256*(w>>4) + (w&15) + 16 r sits at [r][w]
    }
};
template <int OFF>
FFS_DEV void addtid_store(float x) {
    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(x), "i"(OFF) : "memory");
}
template <int S, int... Q>
FFS_DEV void planar_scatter_impl(const cf (&v)[16], unsigned m0, std::integer_sequence<int, Q...>) {
    // volatile statements keep their order; nothing the compiler emits in these kernels touches M0 in between
    asm volatile("s_mov_b32 m0, %0" ::"s"(m0) : "memory", "m0");
    (addtid_store<4 * Q * S>(v[Q].x), ...);
    (addtid_store<4 * (Q * S + RowPlanar::PLANE)>(v[Q].y), ...);
}
template <int S, int... Q>
FFS_DEV void planar_gather_impl(cf (&v)[16], const float* lds, int g, std::integer_sequence<int, Q...>) {
    ((v[Q].x = lds[g + 16 * Q]), ...);
    ((v[Q].y = lds[g + 16 * Q + RowPlanar::PLANE]), ...);
}
// fft_regs<4096> with the planar exchanges.  `lds` = the buffer as floats (RowPlanar::ROW_DWORDS of them).
template <bool LB>
FFS_DEV void fft4096_planar(cf (&v)[16], float* lds, const RowPlanar& ad, const TwRegs<4096>& tw) {
    constexpr int L = 4096;
    typedef Shape<L> S;
    const auto seq = std::make_integer_sequence<int, 16>{};
    stage_first(v);
    block_sync<LB>();
    planar_scatter_impl<RowPlanar::S1>(v, ad.m0, seq);
    block_sync<LB>();
    planar_gather_impl<RowPlanar::S1>(v, lds, ad.g1, seq);
    stage_compute<L, S::R1, 16, false>(v, tw.s1);
    block_sync<LB>();
    planar_scatter_impl<RowPlanar::S2>(v, ad.m0, seq);
    block_sync<LB>();
    planar_gather_impl<RowPlanar::S2>(v, lds, ad.g2, seq);
    stage_compute<L, S::R2, 256, false>(v, tw.s2);
}

// Phase-staggered 4096-point transform for blocks made of TWO teams of 256 threads (k_mid_seg_duo).  Same arithmetic
// and LDS traffic as fft_regs<4096>, but the barrier that protects the exchange buffer against the next scatter (all
// gathers done) sits in the MIDDLE of the following butterfly instead of at its end.  The workgroup barrier counts
// arrivals of all eight waves, so a team that starts one barrier later than the other stays exactly one interval
// behind: while one team waits for its exchange (scatter drained -> barrier -> gather returned) the other team's waves
// own the VALUs, and the other way round in the next interval.  Two independent 256-thread blocks on a CU do not settle
// into that alternation by themselves (both stations are shared: measured 1097 ns per transform per CU against 729 if
// perfect, profiles/fft_core_rate.hip).  Every call executes exactly FOUR barriers; an idle team calls stag_idle().
template <class Addr>
FFS_DEV void fft4096_stag(cf (&v)[16], cf* lds, int u, Addr& addr, const TwRegs<4096>& tw) {
    constexpr int L = 4096;
    stage_first(v);
    stage_scatter<L, 16, 1>(v, lds, u, addr);
    lds_barrier();  // the team's scatter is complete
    stage_gather<L>(v, lds, u, addr);
    {
        cf t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = v[r];
        bfly16_tw_front(t, tw.s1.w);
        lds_barrier();  // every wave of the team has its gathered values: the buffer may be overwritten
        bfly16_tw_back(t, tw.s1.w);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = t[Bfly<16>::slot_of(r)];
    }
    stage_scatter<L, 16, 16>(v, lds, u, addr);
    lds_barrier();
    stage_gather<L>(v, lds, u, addr);
    {
        cf t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = v[r];
        bfly16_tw_front(t, tw.s2.w);
        lds_barrier();
        bfly16_tw_back(t, tw.s2.w);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = t[Bfly<16>::slot_of(r)];
    }
}
FFS_DEV void stag_idle() {
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_barrier();
}

}  // namespace ffsa

#define CHECK(x)                                                   \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

enum { FULL = 0, NO_LDS, NO_VALU, NO_BARRIER, PLANAR, PLANAR_NO_VALU };

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
    RowPlanar pad((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem, u);
    float* ldsf = reinterpret_cast<float*>(smem);
    const auto seq = std::make_integer_sequence<int, 16>{};
    for (int it = 0; it < iters; ++it) {
        if constexpr (VARIANT == FULL) {
            fft_regs<L, RowAddr<L>, true>(v, lds, u, addr, twr);
        } else if constexpr (VARIANT == PLANAR) {
            fft4096_planar<true>(v, ldsf, pad, twr);
        } else if constexpr (VARIANT == PLANAR_NO_VALU) {
            lds_barrier();
            planar_scatter_impl<RowPlanar::S1>(v, pad.m0, seq);
            lds_barrier();
            planar_gather_impl<RowPlanar::S1>(v, ldsf, pad.g1, seq);
            lds_barrier();
            planar_scatter_impl<RowPlanar::S2>(v, pad.m0, seq);
            lds_barrier();
            planar_gather_impl<RowPlanar::S2>(v, ldsf, pad.g2, seq);
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
        if constexpr (VARIANT != NO_VALU && VARIANT != PLANAR_NO_VALU) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = cmac(acc[q], v[q], twr.s1.w[q % 6]);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(v[q]));
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) out[((size_t)blockIdx.x * 256 + u) * 16 + q] = acc[q] + v[q];
}

// Two teams of 256 threads in ONE 512-thread block, each with its own row buffer, running fft4096_stag: the barrier
// that frees the exchange buffer sits in the middle of the next butterfly.  OFFSET: team 1 starts one barrier later,
// so the teams alternate between "waiting for the exchange" and "owning the VALUs" (k_mid_seg_duo's schedule);
// without it the eight waves move in lockstep.
template <bool OFFSET>
__global__ __launch_bounds__(512, 1) void k_core_duo(const cf* __restrict__ tw, cf* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int L = 4096;
    const int team = threadIdx.x >> 8, u = threadIdx.x & 255;
    cf* lds = reinterpret_cast<cf*>(smem) + team * RowAddr<L>::ROW_ELEMS;
    RowAddr<L> addr(0, u);
    TwRegs<L> twr;
    twr.load(tw, u);
    cf v[16], acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        v[q] = mk(1e-3f * (float)(u + q), 1e-3f * (float)(u - q));
        acc[q] = mk(0.f, 0.f);
    }
    if (OFFSET && team == 1) lds_barrier();
    for (int it = 0; it < iters; ++it) {
        fft4096_stag(v, lds, u, addr, twr);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = cmac(acc[q], v[q], twr.s1.w[q % 6]);
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(v[q]));
    }
    if (OFFSET && team == 0) lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) out[((size_t)blockIdx.x * 512 + threadIdx.x) * 16 + q] = acc[q] + v[q];
}

// ---- wave-local 1024-point rows: ONE wave per row transform (64 lanes x 16 values, 16*16*4), both exchanges inside
// the wave's own LDS region, no workgroup barrier at all (LDS operations of a wave execute in order).  The question:
// what would the row pass cost per POINT if rows were 1024 instead of 4096 points long (N = 256 x 1024 instead of
// 64 x 4096), i.e. with every wave of the CU an independent unit of overlap?
template <int L, class Addr>
FFS_DEV void fft_wave(cf (&v)[16], cf* lds, int u, Addr& addr, const TwRegs<L>& tw) {
    typedef Shape<L> S;
    stage_first(v);
    stage_scatter<L, 16, 1>(v, lds, u, addr);
    stage_gather<L>(v, lds, u, addr);
    stage_compute<L, S::R1, 16, false>(v, tw.s1);
    if constexpr (S::R2 > 1) {
        stage_scatter<L, S::R1, 16>(v, lds, u, addr);
        stage_gather<L>(v, lds, u, addr);
        stage_compute<L, S::R2, 256, false>(v, tw.s2);
    }
}
__global__ __launch_bounds__(256) void k_core_wave1024(const cf* __restrict__ tw, cf* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int L = 1024;
    const int wave = threadIdx.x >> 6, u = threadIdx.x & 63;
    cf* lds = reinterpret_cast<cf*>(smem) + wave * RowAddr<L>::ROW_ELEMS;
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
        fft_wave<L>(v, lds, u, addr, twr);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = cmac(acc[q], v[q], twr.s1.w[q % 6]);
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(v[q]));
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) out[((size_t)blockIdx.x * 256 + threadIdx.x) * 16 + q] = acc[q] + v[q];
}
static void run_wave1024(const cf* tw, cf* out, int n_cu, bool last) {
    const size_t lds = 4 * (size_t)RowAddr<1024>::ROW_ELEMS * sizeof(cf);
    printf("  \"wave_local_1024\": {");
    const int iters = 1600;
    for (int bpc = 1; bpc <= 4; ++bpc) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_core_wave1024, dim3(n_cu * bpc), dim3(256), lds, 0, tw, out, iters);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_core_wave1024, dim3(n_cu * bpc), dim3(256), lds, 0, tw, out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // ns per 1024-point transform per wave, and ns per 4096 POINTS per CU (comparable with the other rows)
        printf("\"%d_blocks_per_cu\": [%.0f, %.0f]%s", bpc, ms * 1e6 / iters, ms * 1e6 / iters / (4 * bpc) * 4, bpc < 4 ? ", " : "");
    }
    printf("}%s\n", last ? "" : ",");
}

// ---- the product transform + the mid pass's load pattern: sixteen 8-byte global loads per thread per item, requested
// one item ahead into a staging buffer (copied when its turn comes), from a buffer small enough to stay in L2 (MODE 1)
// or striding through 2 GB (MODE 2: HBM); MODE 0 = no loads.  66.8 KB of LDS per block: exactly two resident blocks.
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_core_loads(const cf* __restrict__ tw, const cf* __restrict__ src, size_t src_elems,
                                                        cf* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int L = 4096;
    const int u = threadIdx.x;
    RowAddr<L> addr(0, u);
    TwRegs<L> twr;
    twr.load(tw, u);
    cf v[16], xl[16], acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        v[q] = mk(1e-3f * (float)(u + q), 1e-3f * (float)(u - q));
        xl[q] = v[q];
        acc[q] = mk(0.f, 0.f);
    }
    // a row = sixteen 2 KB pieces (one per q) at a stride of 128 KB, as in the tile layout of the 64-row plan
    const size_t row_elems = 16 * 16384;
    size_t row = ((size_t)blockIdx.x * 977) % (src_elems / row_elems);
    auto request = [&]() {
        if constexpr (MODE != 0) {
            const cf* p = src + row * row_elems + u;
#pragma unroll
            for (int q = 0; q < 16; ++q) xl[q] = p[(size_t)q * 16384];
            row = (row + (MODE == 2 ? 1031 : 1)) % (MODE == 2 ? (src_elems / row_elems) : 8);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    request();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE != 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = cadd(v[q], xl[q]);
            request();
        }
        fft_regs<L, RowAddr<L>, true>(v, lds, u, addr, twr);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = cmac(acc[q], v[q], twr.s1.w[q % 6]);
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(v[q]));
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) out[((size_t)blockIdx.x * 256 + u) * 16 + q] = acc[q] + v[q];
}
template <int MODE>
static void run_loads(const char* name, const cf* tw, const cf* src, size_t src_elems, cf* out, int n_cu, bool last) {
    const size_t lds = (size_t)RowAddr<4096>::ROW_ELEMS * sizeof(cf) + 32768;
    CHECK(hipFuncSetAttribute((const void*)k_core_loads<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int iters = 100, bpc = 16;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_core_loads<MODE>, dim3(n_cu * bpc), dim3(256), lds, 0, tw, src, src_elems, out, iters);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_core_loads<MODE>, dim3(n_cu * bpc), dim3(256), lds, 0, tw, src, src_elems, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("  \"%s\": {\"ns_per_transform_per_block\": %.0f, \"ns_per_transform_per_cu\": %.0f, \"load_TBps\": %.2f}%s\n", name,
           ms * 1e6 / iters / (bpc / 2), ms * 1e6 / iters / bpc,
           MODE ? (double)n_cu * bpc * iters * 32768.0 / (ms * 1e-3) / 1e12 : 0.0, last ? "" : ",");
}

// ---- the same transform under the mid pass's REGISTER pressure: four live accumulator rows (128 VGPRs) next to the
// row being transformed, as in k_mid_seg_one (250 VGPRs there).  Does the compiler still schedule the exchanges as
// "sixteen reads, one wait, butterflies", or does it trickle them through the few free registers?
__global__ __launch_bounds__(256, 2) void k_core_pressure(const cf* __restrict__ tw, cf* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int L = 4096;
    const int u = threadIdx.x;
    RowAddr<L> addr(0, u);
    TwRegs<L> twr;
    twr.load(tw, u);
    cf v[16], acc[4][16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        v[q] = mk(1e-3f * (float)(u + q), 1e-3f * (float)(u - q));
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][q] = mk(0.f, (float)a);
    }
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            fft_regs<L, RowAddr<L>, true>(v, lds, u, addr, twr);
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][q] = cmac(acc[a][q], v[q], twr.s1.w[q % 6]);
#pragma unroll
            for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(v[q]));
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q)
        out[((size_t)blockIdx.x * 256 + u) * 16 + q] = cadd(cadd(acc[0][q], acc[1][q]), cadd(cadd(acc[2][q], acc[3][q]), v[q]));
}

// ---- the same loads through LDS-DMA (global_load_lds_dwordx4: global -> LDS without touching a VGPR), read back with
// ds_read_b64 when the item's turn comes: does the transform keep its pace next to loads that bypass the register file?
// LDS: row buffer + 32 KB of staging (8 KB per wave: sixteen 512-byte pieces), two resident blocks.
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_core_dma(const cf* __restrict__ tw, const cf* __restrict__ src, size_t src_elems,
                                                      cf* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int L = 4096;
    const int u = threadIdx.x, lane = u & 63, wv = u >> 6;
    cf* stage = lds + RowAddr<L>::ROW_ELEMS + wv * 1024;  // this wave's 8 KB
    RowAddr<L> addr(0, u);
    TwRegs<L> twr;
    twr.load(tw, u);
    cf v[16], acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        v[q] = mk(1e-3f * (float)(u + q), 1e-3f * (float)(u - q));
        acc[q] = mk(0.f, 0.f);
    }
    const size_t row_elems = 16 * 16384;
    size_t row = ((size_t)blockIdx.x * 977) % (src_elems / row_elems);
    auto request = [&]() {
        // one instruction = 1 KB = two 512-byte pieces (q, q + 1): lanes 0..31 piece q, lanes 32..63 piece q + 1, 16 B each
        const char* p = reinterpret_cast<const char*>(src + row * row_elems + wv * 64) + (lane & 31) * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const char* g = p + (size_t)(2 * j + (lane >> 5)) * 16384 * sizeof(cf);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(stage + j * 128), 16, 0, 0);
        }
        row = (row + (MODE == 2 ? 1031 : 1)) % (MODE == 2 ? (src_elems / row_elems) : 8);
        __builtin_amdgcn_sched_barrier(0);
    };
    request();
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA has landed (it reads only its own pieces)
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = cadd(v[q], stage[q * 64 + lane]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        request();
        fft_regs<L, RowAddr<L>, true>(v, lds, u, addr, twr);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = cmac(acc[q], v[q], twr.s1.w[q % 6]);
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(v[q]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 16; ++q) out[((size_t)blockIdx.x * 256 + u) * 16 + q] = acc[q] + v[q];
}
template <int MODE>
static void run_dma(const char* name, const cf* tw, const cf* src, size_t src_elems, cf* out, int n_cu, bool last) {
    const size_t lds = (size_t)RowAddr<4096>::ROW_ELEMS * sizeof(cf) + 32768;
    CHECK(hipFuncSetAttribute((const void*)k_core_dma<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int iters = 100, bpc = 16;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_core_dma<MODE>, dim3(n_cu * bpc), dim3(256), lds, 0, tw, src, src_elems, out, iters);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_core_dma<MODE>, dim3(n_cu * bpc), dim3(256), lds, 0, tw, src, src_elems, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("  \"%s\": {\"ns_per_transform_per_block\": %.0f, \"ns_per_transform_per_cu\": %.0f, \"load_TBps\": %.2f}%s\n", name,
           ms * 1e6 / iters / (bpc / 2), ms * 1e6 / iters / bpc, (double)n_cu * bpc * iters * 32768.0 / (ms * 1e-3) / 1e12,
           last ? "" : ",");
}

template <bool OFFSET>
static void run_duo(const char* name, const cf* tw, cf* out, int n_cu, bool last) {
    const size_t lds = 2 * (size_t)RowAddr<4096>::ROW_ELEMS * sizeof(cf);
    CHECK(hipFuncSetAttribute((const void*)k_core_duo<OFFSET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    printf("  \"%s\": {", name);
    const int iters = 400;
    for (int bpc = 1; bpc <= 2; ++bpc) {  // one 512-thread block per CU = the occupancy of two 256-thread blocks
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_core_duo<OFFSET>, dim3(n_cu * bpc), dim3(512), lds, 0, tw, out, iters);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_core_duo<OFFSET>, dim3(n_cu * bpc), dim3(512), lds, 0, tw, out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // a block does two transforms per iteration
        printf("\"%d_duo_blocks_per_cu\": [%.0f, %.0f]%s", bpc, ms * 1e6 / iters, ms * 1e6 / iters / (2 * bpc), bpc < 2 ? ", " : "");
    }
    printf("}%s\n", last ? "" : ",");
}

template <int VARIANT>
static void run(const char* name, const cf* tw, cf* out, int n_cu, bool last) {
    const size_t lds = (size_t)RowPlanar::ROW_DWORDS * 4 > (size_t)RowAddr<4096>::ROW_ELEMS * sizeof(cf) ? (size_t)RowPlanar::ROW_DWORDS * 4 : (size_t)RowAddr<4096>::ROW_ELEMS * sizeof(cf);
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
    CHECK(hipMalloc(&out, (size_t)n_cu * 16 * 512 * 16 * sizeof(cf)));
    printf("{\"cus\": %d, \"unit\": \"[ns per transform per block, ns per transform per CU] for one 4096-point row transform + 16 cmacs\",\n", n_cu);
    run<FULL>("full", tw, out, n_cu, false);
    run<NO_LDS>("no_lds", tw, out, n_cu, false);
    run<NO_VALU>("no_valu", tw, out, n_cu, false);
    run<NO_BARRIER>("no_barrier", tw, out, n_cu, false);
    run<PLANAR>("planar_addtid_full", tw, out, n_cu, false);
    run<PLANAR_NO_VALU>("planar_addtid_no_valu", tw, out, n_cu, false);
    run_duo<true>("duo_staggered", tw, out, n_cu, false);
    run_duo<false>("duo_lockstep", tw, out, n_cu, false);
    run_wave1024(tw, out, n_cu, false);
    {   // the product transform with the mid pass's footprint: 66.8 KB of LDS per block caps the CU at two resident
        // blocks whatever the dispatcher does; 16 blocks per CU in the grid keep both slots filled
        const size_t lds = (size_t)RowAddr<4096>::ROW_ELEMS * sizeof(cf) + 32768;
        CHECK(hipFuncSetAttribute((const void*)k_core<FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int iters = 100, bpc = 16;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_core<FULL>, dim3(n_cu * bpc), dim3(256), lds, 0, tw, out, iters);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_core<FULL>, dim3(n_cu * bpc), dim3(256), lds, 0, tw, out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // bpc blocks per CU run two at a time: bpc/2 rounds of `iters` transforms each
        printf("  \"full_exactly_two_resident_blocks\": {\"ns_per_transform_per_block\": %.0f, \"ns_per_transform_per_cu\": %.0f},\n",
               ms * 1e6 / iters / (bpc / 2), ms * 1e6 / iters / bpc);
    }
    {
        cf* src;
        const size_t src_elems = (size_t)1 << 28;  // 2 GB
        CHECK(hipMalloc(&src, src_elems * sizeof(cf)));
        CHECK(hipMemset(src, 0, src_elems * sizeof(cf)));
        {
            const size_t lds = (size_t)RowAddr<4096>::ROW_ELEMS * sizeof(cf) + 32768;
            CHECK(hipFuncSetAttribute((const void*)k_core_pressure, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int iters = 100, bpc = 16;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(k_core_pressure, dim3(n_cu * bpc), dim3(256), lds, 0, tw, out, iters);
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_core_pressure, dim3(n_cu * bpc), dim3(256), lds, 0, tw, out, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("  \"four_live_accumulator_rows\": {\"ns_per_transform_per_block\": %.0f, \"ns_per_transform_per_cu\": %.0f},\n",
                   ms * 1e6 / iters / (bpc / 2), ms * 1e6 / iters / bpc);
        }
        run_loads<0>("with_loads_none", tw, src, src_elems, out, n_cu, false);
        run_loads<1>("with_loads_l2_resident", tw, src, src_elems, out, n_cu, false);
        run_loads<2>("with_loads_hbm", tw, src, src_elems, out, n_cu, false);
        run_dma<1>("with_lds_dma_l2_resident", tw, src, src_elems, out, n_cu, false);
        run_dma<2>("with_lds_dma_hbm", tw, src, src_elems, out, n_cu, true);
    }
    printf("}\n");
    return 0;
}
