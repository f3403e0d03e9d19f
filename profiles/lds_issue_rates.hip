// LDS instruction issue rates on gfx950 with NO vector-ALU work between the DS operations (VERDICT r5, weak #4a): the
// round-5 harness (lds_atomic_ceiling.hip) computed every address in the loop (add, compare, select: 4-6 VALU operations
// per DS operation) and read `ds_write_b32` at 6.7 cycles per wave-instruction where the microarchitecture guide's table
// says 4 -- so its `ds_add_u32` figure (7.5 cycles) could have been the harness, not the LDS.  Here every lane's 16
// addresses and the data sit in registers before the timed loop, the loop body is 16 DS instructions with IMMEDIATE
// offsets (offset:0 / offset:8192 alternate so that consecutive instructions do not hit the same words) and one scalar
// loop counter: the only thing issued besides the DS operations is s_sub + s_cbranch per 16 of them.
//
//   instructions : ds_write_b32 (the control the guide tabulates: 4 cycles), ds_add_u32, ds_add_rtn_u32 (result unused but
//                  waited for), ds_read_b32 / ds_read_b64 (controls: 2 cycles)
//   addresses    : conflict-free (lane -> consecutive words), uniformly random words (k_runs_corr's situation),
//                  2-way, 4-way bank conflicts, one word (broadcast)
//   active lanes : 64, 52 (k_runs_corr's measured mean), 32, 16
//   residency    : 16 waves per CU (4 x 256 threads) and 32 waves per CU (4 x 512 threads, the kernel's)
// Prints one JSON object: cycles per wave-instruction per CU = waves_per_CU_in_flight-independent service time
//   = (CU clock cycles of the launch) / (DS wave-instructions issued per CU).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o profiles/_bin/lds_issue_rates profiles/lds_issue_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                   \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

enum { WRITE = 0, ADD = 1, ADD_RTN = 2, READ32 = 3, READ64 = 4 };
enum { FREE = 0, RANDOM = 1, WAY2 = 2, WAY4 = 3, SAME_WORD = 4 };

__device__ unsigned hash32(unsigned x) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}

// 16 KB window of words per address register + 8 KB immediate offset: the LDS allocation is 40 KB
constexpr int WORDS = 4096;

#define DS16(OPSTR, TAIL)                                                                                           \
    asm volatile(OPSTR " %0, %16" TAIL "\n" OPSTR " %1, %16 offset:8192" TAIL "\n" OPSTR " %2, %16" TAIL "\n" OPSTR   \
                       " %3, %16 offset:8192" TAIL "\n" OPSTR " %4, %16" TAIL "\n" OPSTR " %5, %16 offset:8192" TAIL \
                       "\n" OPSTR " %6, %16" TAIL "\n" OPSTR " %7, %16 offset:8192" TAIL "\n" OPSTR " %8, %16" TAIL  \
                       "\n" OPSTR " %9, %16 offset:8192" TAIL "\n" OPSTR " %10, %16" TAIL "\n" OPSTR                  \
                       " %11, %16 offset:8192" TAIL "\n" OPSTR " %12, %16" TAIL "\n" OPSTR " %13, %16 offset:8192" TAIL \
                       "\n" OPSTR " %14, %16" TAIL "\n" OPSTR " %15, %16 offset:8192" TAIL "\n" ::"v"(a[0]),          \
                 "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),    \
                 "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(one)                      \
                 : "memory")

template <int OP, int THREADS>
__global__ __launch_bounds__(THREADS) void k_rate(int pattern, unsigned long long lane_mask, int iters, unsigned* out) {
    __shared__ unsigned lds[WORDS + 2048 + 2048];  // 32 KB: words [0, WORDS) through the registers, + 2048 through the offset
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < WORDS + 4096; i += THREADS) lds[i] = 0;
    unsigned a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        unsigned w;
        if (pattern == FREE)
            w = (unsigned)(lane + 64 * ((k + wave) & 31));
        else if (pattern == RANDOM)
            w = hash32((unsigned)(tid * 16 + k) * 2654435761u + blockIdx.x * 97u);
        else if (pattern == WAY2)  // lanes l and l + 16 of a 32-lane group share a bank (different words)
            w = (unsigned)((lane & 15) + 32 * (lane >> 4) + 16 * (k & 1) + 256 * k);
        else if (pattern == WAY4)  // four lanes of a group per bank
            w = (unsigned)((lane & 7) + 32 * (lane >> 3) + 8 * (k & 3) + 512 * (k & 7));
        else
            w = (unsigned)(k * 37 + 5);
        a[k] = (w % (unsigned)WORDS) * 4u;
        if (OP == READ64) a[k] &= ~7u;
    }
    const unsigned one = 1u;
    __syncthreads();
    const bool active = (lane_mask >> lane) & 1ull;
    if (active) {
        for (int it = 0; it < iters; ++it) {
            if (OP == WRITE)
                DS16("ds_write_b32", "");
            else if (OP == ADD)
                DS16("ds_add_u32", "");
            else if (OP == ADD_RTN) {
                // returning atomics: 16 results into 16 scratch registers, waited for once per 16
                unsigned r[16];
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(r[k]) : "v"(a[k]), "v"(one) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(r[k]));
            } else if (OP == READ32) {
                unsigned r[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("ds_read_b32 %0, %1" : "=v"(r[k]) : "v"(a[k]) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(r[k]));
            } else {
                unsigned long long r[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("ds_read_b64 %0, %1" : "=v"(r[k]) : "v"(a[k]) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(r[k]));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned s = 0;
    for (int i = tid; i < WORDS + 4096; i += THREADS) s += lds[i];
    if (s == 0x9e3779b9u) out[blockIdx.x] = s;
}

template <int OP, int THREADS>
static double cycles_per_instr(int pattern, unsigned long long mask, int n_cu, double clk_hz, unsigned* out, int iters) {
    const int wg_per_cu = 4, rounds = 4;
    const int grid = n_cu * wg_per_cu * rounds;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate<OP, THREADS>), dim3(grid), dim3(THREADS), 0, 0, pattern, mask, iters, out);
    // empty-loop launch (iters = 0) gives the fixed part (launch, zeroing, final sum): subtracted
    float ms_full = 0, ms_empty = 0;
    const int reps = 5;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_rate<OP, THREADS>), dim3(grid), dim3(THREADS), 0, 0, pattern, mask, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&ms_full, e0, e1));
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_rate<OP, THREADS>), dim3(grid), dim3(THREADS), 0, 0, pattern, mask, 0, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&ms_empty, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    const double sec = (ms_full - ms_empty) / reps * 1e-3;
    const double instr_per_cu = (double)wg_per_cu * rounds * (THREADS / 64) * 16.0 * iters;
    return sec * clk_hz / instr_per_cu;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    unsigned* out;
    CHECK(hipMalloc(&out, 1 << 20));
    const char* pat[5] = {"conflict_free", "random_words", "bank_conflict_2way", "bank_conflict_4way", "same_word"};
    const int iters = 512;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_MHz\": %.0f,\n", prop.gcnArchName, n_cu, clk / 1e6);
    printf(" \"unit\": \"CU clock cycles per DS wave-instruction (all 64 lanes unless stated), every CU streaming; loop body = 16 DS "
           "instructions with immediate offsets + one scalar counter, no vector ALU\",\n");
#define ROW(NAME, OP, THREADS)                                                                                       \
    printf(" \"%s\": {", NAME);                                                                                      \
    for (int p = 0; p < 5; ++p)                                                                                      \
        printf("\"%s\": %.2f%s", pat[p], cycles_per_instr<OP, THREADS>(p, ~0ull, n_cu, clk, out, iters), p < 4 ? ", " : ""); \
    printf("},\n");
    ROW("ds_write_b32_16_waves_per_cu", WRITE, 256)
    ROW("ds_write_b32_32_waves_per_cu", WRITE, 512)
    ROW("ds_add_u32_16_waves_per_cu", ADD, 256)
    ROW("ds_add_u32_32_waves_per_cu", ADD, 512)
    ROW("ds_add_rtn_u32_32_waves_per_cu", ADD_RTN, 512)
    ROW("ds_read_b32_32_waves_per_cu", READ32, 512)
    ROW("ds_read_b64_32_waves_per_cu", READ64, 512)
    // active lanes (random words, the kernel's residency): does a partly filled atomic cost less?
    struct {
        const char* name;
        unsigned long long mask;
    } masks[4] = {{"64", ~0ull}, {"52", 0x000fffffffffffffull}, {"32_first_half", 0xffffffffull}, {"16_every_4th", 0x1111111111111111ull}};
    printf(" \"ds_add_u32_random_words_by_active_lanes_32_waves_per_cu\": {");
    for (int m = 0; m < 4; ++m)
        printf("\"lanes_%s\": %.2f%s", masks[m].name, cycles_per_instr<ADD, 512>(RANDOM, masks[m].mask, n_cu, clk, out, iters), m < 3 ? ", " : "");
    printf("},\n \"ds_add_u32_conflict_free_by_active_lanes_32_waves_per_cu\": {");
    for (int m = 0; m < 4; ++m)
        printf("\"lanes_%s\": %.2f%s", masks[m].name, cycles_per_instr<ADD, 512>(FREE, masks[m].mask, n_cu, clk, out, iters), m < 3 ? ", " : "");
    printf("},\n \"guide\": \"MI355X_MICROARCH.md LDS table: ds_write_b32 4 cycles per wave-instruction (2 source dwords x 2 cycles), "
           "ds_read_b32 / b64 2 cycles\"}\n");
    return 0;
}
