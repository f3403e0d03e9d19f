// Workgroup dispatch cost on gfx950 for k_runs_corr's launch shape: 512-thread workgroups with 38 KB of static LDS (four
// resident per CU), N workgroups per launch, each doing (a) nothing, (b) a chain of `depth` dependent global loads
// (descriptor -> header -> list), (c) the chain + zeroing 24 KB of LDS + one barrier.  Prints ns per workgroup (whole
// chip) -- the floor under any per-candidate-workgroup kernel -- for 256- and 512-thread workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o profiles/_bin/wg_dispatch_rate profiles/wg_dispatch_rate.hip
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

template <int THREADS, int LDS_WORDS>
__global__ __launch_bounds__(THREADS) void k_wg(const int* __restrict__ chain, int depth, int work, int* __restrict__ out) {
    __shared__ unsigned lds[LDS_WORDS];
    int idx = blockIdx.x & 4095;
    for (int d = 0; d < depth; ++d) idx = chain[idx];  // dependent loads (uniform: scalar loads)
    if (work) {
        for (int i = threadIdx.x; i < LDS_WORDS; i += THREADS) lds[i] = 0;
        __syncthreads();
        if (lds[(threadIdx.x * 7 + idx) % LDS_WORDS] == 12345u) out[blockIdx.x] = 1;
    }
    if (idx == -7) out[blockIdx.x] = idx;
}

template <int THREADS, int LDS_WORDS>
static double ns_per_wg(int n_wg, int depth, int work, const int* chain, int* out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_wg<THREADS, LDS_WORDS>), dim3(n_wg), dim3(THREADS), 0, 0, chain, depth, work, out);
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_wg<THREADS, LDS_WORDS>), dim3(n_wg), dim3(THREADS), 0, 0, chain, depth, work, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return 1e6 * ms / reps / n_wg;
}

int main() {
    int *chain, *out;
    CHECK(hipMalloc(&chain, 4096 * sizeof(int)));
    CHECK(hipMalloc(&out, 1 << 20));
    int h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (i * 1237 + 11) & 4095;
    CHECK(hipMemcpy(chain, h, sizeof h, hipMemcpyHostToDevice));
    const int n_wg = 57344;  // 8192 pairs x 7 candidates
    printf("{\"workgroups_per_launch\": %d, \"unit\": \"ns per workgroup, whole chip (launch time / workgroups)\",\n", n_wg);
    printf(" \"512_threads_38KB_lds\": {\"empty\": %.2f, \"chain_of_3_loads\": %.2f, \"chain_of_6_loads\": %.2f, \"chain_3_plus_zero_24KB_and_barrier\": %.2f},\n",
           ns_per_wg<512, 9728>(n_wg, 0, 0, chain, out), ns_per_wg<512, 9728>(n_wg, 3, 0, chain, out), ns_per_wg<512, 9728>(n_wg, 6, 0, chain, out),
           ns_per_wg<512, 9728>(n_wg, 3, 1, chain, out));
    printf(" \"512_threads_1KB_lds\": {\"empty\": %.2f, \"chain_of_3_loads\": %.2f},\n", ns_per_wg<512, 256>(n_wg, 0, 0, chain, out),
           ns_per_wg<512, 256>(n_wg, 3, 0, chain, out));
    printf(" \"256_threads_38KB_lds\": {\"empty\": %.2f, \"chain_of_3_loads\": %.2f, \"chain_3_plus_zero_24KB_and_barrier\": %.2f},\n",
           ns_per_wg<256, 9728>(n_wg, 0, 0, chain, out), ns_per_wg<256, 9728>(n_wg, 3, 0, chain, out), ns_per_wg<256, 9728>(n_wg, 3, 1, chain, out));
    printf(" \"1024_threads_76KB_lds\": {\"empty\": %.2f, \"chain_of_3_loads\": %.2f},\n", ns_per_wg<1024, 19456>(n_wg / 2, 0, 0, chain, out),
           ns_per_wg<1024, 19456>(n_wg / 2, 3, 0, chain, out));
    printf(" \"note\": \"k_runs_corr at 0.26 us per seven-candidate pair spends 37 ns of chip time per candidate workgroup\"}\n");
    return 0;
}
