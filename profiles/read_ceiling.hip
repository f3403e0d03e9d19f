// What does a READ-ONLY sweep over one 90-minute PCM file (518.4 MB, larger than the 256 MB Infinity Cache) reach on
// this box?  The ceiling k_vad_energy is measured against: 16-byte loads, U loads in flight per lane, B blocks per CU,
// grid-stride, a trivial integer reduction (one value per block written).  Prints TB/s per configuration.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o profiles/_bin/read_ceiling profiles/read_ceiling.hip
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

typedef int v4i __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const v4i* __restrict__ p, long long nvec, int* __restrict__ out) {
    const long long stride = (long long)gridDim.x * 256 * U;
    int acc = 0;
    for (long long i = (long long)blockIdx.x * 256 * U + threadIdx.x; i < nvec; i += stride) {
        v4i w[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const long long j = i + (long long)k * 256;
            if (NT)
                w[k] = j < nvec ? __builtin_nontemporal_load(p + j) : v4i{0, 0, 0, 0};
            else
                w[k] = j < nvec ? p[j] : v4i{0, 0, 0, 0};
        }
#pragma unroll
        for (int k = 0; k < U; ++k) acc += w[k].x ^ w[k].y ^ w[k].z ^ w[k].w;
    }
    if (acc == 0x12345678) out[blockIdx.x] = acc;
}

template <int U, bool NT>
static void run(const v4i* p, long long nvec, int* out, int n_cu, bool last) {
    printf("  \"loads_in_flight_%d%s\": {", U, NT ? "_nt" : "");
    const int bpcs[4] = {2, 4, 8, 16};
    for (int b = 0; b < 4; ++b) {
        const int grid = n_cu * bpcs[b];
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_read<U, NT>), dim3(grid), dim3(256), 0, 0, p, nvec, out);
        CHECK(hipEventRecord(e0));
        const int reps = 10;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_read<U, NT>), dim3(grid), dim3(256), 0, 0, p, nvec, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("\"%d_blocks_per_cu\": %.3f%s", bpcs[b], (double)nvec * 16 / (ms / reps * 1e-3) / 1e12, b < 3 ? ", " : "");
    }
    printf("}%s\n", last ? "" : ",");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const long long nbytes = 518400000LL, nvec = nbytes / 16;
    v4i* p;
    int* out;
    CHECK(hipMalloc(&p, nbytes));
    CHECK(hipMemset(p, 1, nbytes));
    CHECK(hipMalloc(&out, 1 << 20));
    printf("{\"what\": \"read-only sweep over 518.4 MB, TB/s\", \"cus\": %d,\n", n_cu);
    run<1, false>(p, nvec, out, n_cu, false);
    run<2, false>(p, nvec, out, n_cu, false);
    run<4, false>(p, nvec, out, n_cu, false);
    run<8, false>(p, nvec, out, n_cu, false);
    run<4, true>(p, nvec, out, n_cu, true);
    printf("}\n");
    return 0;
}
