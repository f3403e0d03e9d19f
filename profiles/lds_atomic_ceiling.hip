// LDS scatter-add ceiling on gfx950: the resource k_runs_corr (csrc/ffs_runs.h) lives off -- one `ds_add_u32` per boundary
// coincidence into a 12 288-lag histogram.  Measures lane-adds per clock per CU (and per second on the whole chip) for
//   address patterns : conflict-free (consecutive lanes -> consecutive words), uniformly random words of a 48 KB array
//                      (the kernel's situation), same bank (32-way conflict), one word (broadcast address),
//   active lanes     : 64, 32 (lanes 0..31), 32 (even lanes), 16 (every fourth lane), 8,
//   instructions     : ds_add_u32 (no return), ds_add_rtn_u32 (returned value consumed), ds_write_b32 (for scale),
//   residency        : 4 x 256-thread workgroups per CU (the kernel's) and 2 x 512.
// Every wave issues ITER x 16 DS operations; addresses come from 16 per-lane registers rotated by the iteration, so the
// VALU work per add is one v_add + one v_and.  Prints one JSON object.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o profiles/_bin/lds_atomic_ceiling profiles/lds_atomic_ceiling.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x)                                                   \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

// 256-thread layout: 6144-word histogram (two lags per word) in 39 KB of LDS -> four workgroups per CU (round 4's kernel);
// 512-thread layouts: 12 288-word histogram (one lag per word) in 62 KB -> two workgroups per CU; 6144-word histogram
// (two lags per word, the value shifted by the lag's parity) in 38 KB -> four workgroups per CU (round 5's kernel)
enum { ADD = 0, ADD_RTN = 1, WRITE = 2 };
enum { FREE = 0, RANDOM = 1, SAME_BANK = 2, SAME_WORD = 3, RANDOM_PACKED16 = 4 };

__device__ unsigned hash32(unsigned x) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}

template <int OP, int THREADS>
__global__ __launch_bounds__(THREADS) void k_lds(int pattern, unsigned long long lane_mask, int iters, int words, unsigned* out,
                                                 long long* cycles) {
    extern __shared__ unsigned hist[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < words; i += THREADS) hist[i] = 0;
    unsigned off[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        unsigned w;
        if (pattern == FREE)
            w = (unsigned)(lane + 64 * k + 1024 * (tid >> 6));
        else if (pattern == RANDOM || pattern == RANDOM_PACKED16)
            w = hash32((unsigned)(tid * 16 + k) * 2654435761u + blockIdx.x * 97u);
        else if (pattern == SAME_BANK)
            w = (unsigned)(lane * 32 + k * 2048 + 7);
        else
            w = (unsigned)(k * 37 + 5);
        off[k] = (w % (unsigned)words) * 4u;
    }
    __syncthreads();
    const bool active = (lane_mask >> lane) & 1ull;
    const unsigned wrap = (unsigned)words * 4u;
    unsigned acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    if (active) {
        unsigned base = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                unsigned a = off[k] + base;
                a = a >= wrap ? a - wrap : a;
                unsigned v = 1u;
                if (pattern == RANDOM_PACKED16) v = (a & 4u) ? 0x10000u : 1u;  // the 16-bits-per-lag variant's value logic
                if (OP == ADD)
                    asm volatile("ds_add_u32 %0, %1" ::"v"(a), "v"(v) : "memory");
                else if (OP == ADD_RTN) {
                    unsigned r;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(v) : "memory");
                    acc += r;
                } else
                    asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
            }
            base += 256u * 4u;  // (keeps lane -> bank relations of the pattern: 256 words = 8 x 32 banks)
            base = base >= wrap ? base - wrap : base;
        }
        if (OP == ADD_RTN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    unsigned s = acc;
    for (int i = tid; i < words; i += THREADS) s += hist[i];
    if (s == 0x9e3779b9u) out[blockIdx.x] = s;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP, int THREADS>
static double run(int pattern, unsigned long long mask, int wg_per_cu, int n_cu, unsigned* out, long long* cyc, int iters, int WORDS,
                  size_t lds) {
    const int grid = n_cu * wg_per_cu * 4;  // four rounds of resident workgroups
    CHECK(hipFuncSetAttribute((const void*)k_lds<OP, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_lds<OP, THREADS>), dim3(grid), dim3(THREADS), lds, 0, pattern, mask, iters, WORDS, out, cyc);
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_lds<OP, THREADS>), dim3(grid), dim3(THREADS), lds, 0, pattern, mask, iters, WORDS, out, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    const double lanes = (double)__builtin_popcountll(mask);
    const double adds = (double)grid * (THREADS / 64) * lanes * 16.0 * iters;  // lane-adds per launch
    return adds / (ms / reps * 1e-3);                                           // lane-adds per second, whole chip
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;  // Hz
    unsigned* out;
    long long* cyc;
    CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMalloc(&cyc, 1 << 20));
    const char* pat_names[5] = {"conflict_free", "random_words", "same_bank_32way", "same_word", "random_words_packed16_value"};
    struct {
        const char* name;
        unsigned long long mask;
    } masks[5] = {{"64", ~0ull},
                  {"32_first_half", 0xffffffffull},
                  {"32_even_lanes", 0x5555555555555555ull},
                  {"16_every_4th", 0x1111111111111111ull},
                  {"8_every_8th", 0x0101010101010101ull}};
    const int iters = 256;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_MHz\": %.0f,\n", prop.gcnArchName, n_cu, clk / 1e6);
    printf(" \"unit\": \"lane-adds per clock per CU (whole-chip G lane-adds/s in parentheses fields *_Gps)\",\n");
    printf(" \"ds_add_u32_4x256_per_cu\": {\n");
    for (int p = 0; p < 5; ++p) {
        printf("  \"%s\": {", pat_names[p]);
        for (int m = 0; m < 5; ++m) {
            const double r = run<ADD, 256>(p, masks[m].mask, 4, n_cu, out, cyc, iters, 6144, 39 * 1024);
            printf("\"lanes_%s\": %.3f, \"lanes_%s_Gps\": %.1f%s", masks[m].name, r / clk / n_cu, masks[m].name, r / 1e9, m < 4 ? ", " : "");
        }
        printf("}%s\n", p < 4 ? "," : "");
    }
    printf(" },\n \"ds_add_u32_2x512_per_cu\": {\n");
    for (int p = 0; p < 2; ++p) {
        printf("  \"%s\": {", pat_names[p]);
        for (int m = 0; m < 3; ++m) {
            const double r = run<ADD, 512>(p, masks[m].mask, 2, n_cu, out, cyc, iters, 12288, 62 * 1024);
            printf("\"lanes_%s\": %.3f, \"lanes_%s_Gps\": %.1f%s", masks[m].name, r / clk / n_cu, masks[m].name, r / 1e9, m < 2 ? ", " : "");
        }
        printf("}%s\n", p < 1 ? "," : "");
    }
    printf(" },\n \"ds_add_u32_4x512_per_cu\": {\n");
    for (int p = 0; p < 2; ++p) {
        printf("  \"%s\": {", pat_names[p == 0 ? 0 : 4]);
        for (int m = 0; m < 3; ++m) {
            const double r = run<ADD, 512>(p == 0 ? 0 : 4, masks[m].mask, 4, n_cu, out, cyc, iters, 6144, 38 * 1024);
            printf("\"lanes_%s\": %.3f, \"lanes_%s_Gps\": %.1f%s", masks[m].name, r / clk / n_cu, masks[m].name, r / 1e9, m < 2 ? ", " : "");
        }
        printf("}%s\n", p < 1 ? "," : "");
    }
    printf(" },\n \"ds_add_rtn_u32_4x256_per_cu\": {\n");
    for (int p = 0; p < 2; ++p) {
        const double r = run<ADD_RTN, 256>(p, ~0ull, 4, n_cu, out, cyc, iters, 6144, 39 * 1024);
        printf("  \"%s\": {\"lanes_64\": %.3f, \"lanes_64_Gps\": %.1f}%s\n", pat_names[p], r / clk / n_cu, r / 1e9, p < 1 ? "," : "");
    }
    printf(" },\n \"ds_write_b32_4x256_per_cu\": {\n");
    for (int p = 0; p < 2; ++p) {
        const double r = run<WRITE, 256>(p, ~0ull, 4, n_cu, out, cyc, iters, 6144, 39 * 1024);
        printf("  \"%s\": {\"lanes_64\": %.3f, \"lanes_64_Gps\": %.1f}%s\n", pat_names[p], r / clk / n_cu, r / 1e9, p < 1 ? "," : "");
    }
    printf(" },\n \"note\": \"39 KB of dynamic LDS per 256-thread workgroup (four resident per CU), 62 KB per 512-thread workgroup (two per CU); "
           "rates = lane-adds issued / wall time of the launch (HIP events), 16 x %d DS operations per wave\"}\n", iters);
    return 0;
}
