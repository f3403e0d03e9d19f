// Write-only micro-benchmark: which store shape does the HBM write path of gfx950 like?  Context for pass A, whose
// block writes one contiguous 32 KB tile (64 rows x 64 columns of complex fp32) as sixteen 8-byte stores per thread.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_patterns profiles/store_patterns.hip && /tmp/store_patterns
//
// Patterns (all write the same bytes, every 128-byte line completely):
//   fill16     grid-stride, 16 bytes per lane, consecutive lanes consecutive (what torch.fill_ does)
//   tile_8B    pass A today: thread (c = t%64, u = t/64) stores 8 bytes to rows u + 4q (q < 16) of its block's tile
//   tile_16B   two adjacent columns per thread: (cp = t%32, jg = t/32) stores 16 bytes to rows jg + 8i (i < 8)
//   tile_8B_w  as tile_8B but a wave owns 16 consecutive rows (8 KB contiguous per wave)
//   tile128    128-column tile (64 KB), 512 threads, 16-byte stores: one 1 KB row chunk per wave instruction
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__global__ __launch_bounds__(256) void k_fill16(f4* out, size_t n16) {
    const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = v;
}

// XCD-aware tile order of pass A: 64 tiles per transform, tile = (b % 8) * 8 + b / 8
__device__ __forceinline__ size_t tile_of(unsigned b) {
    const unsigned y = b / 64, x = b % 64;
    return (size_t)y * 64 + (x % 8) * 8 + x / 8;
}

__global__ __launch_bounds__(256) void k_tile_8B(f2* out) {
    f2* tile = out + tile_of(blockIdx.x) * 4096;
    const int c = threadIdx.x % 64, u = threadIdx.x / 64;
    const f2 v = {1.f, (float)threadIdx.x};
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[(u + 4 * q) * 64 + c] = v;
}

__global__ __launch_bounds__(256) void k_tile_16B(f2* out) {
    f2* tile = out + tile_of(blockIdx.x) * 4096;
    const int cp = threadIdx.x % 32, jg = threadIdx.x / 32;
    const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f4*>(&tile[(jg + 8 * i) * 64 + 2 * cp]) = v;
}

__global__ __launch_bounds__(256) void k_tile_8B_w(f2* out) {
    f2* tile = out + tile_of(blockIdx.x) * 4096;
    const int c = threadIdx.x % 64, u = threadIdx.x / 64;
    const f2 v = {1.f, (float)threadIdx.x};
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[(16 * u + q) * 64 + c] = v;
}

__global__ __launch_bounds__(512) void k_tile128(f2* out) {
    f2* tile = out + (size_t)blockIdx.x * 8192;  // 64 rows x 128 columns
    const int cp = threadIdx.x % 64, jg = threadIdx.x / 64;
    const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f4*>(&tile[(jg + 8 * i) * 128 + 2 * cp]) = v;
}

template <class F>
static double timed(F launch, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e-3 / reps;
}

int main() {
    const size_t bytes = (size_t)8 << 30;  // 8 GiB: far beyond L2 and the 256 MiB Infinity Cache
    void* buf = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    const unsigned tiles = (unsigned)(bytes / 32768);
    const int reps = 5;
    double t;
    printf("{");
    t = timed([&] { hipLaunchKernelGGL(k_fill16, dim3(256 * 16), dim3(256), 0, 0, (f4*)buf, bytes / 16); }, reps);
    printf("\"fill16_GBps\": %.0f, ", bytes / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL(k_tile_8B, dim3(tiles), dim3(256), 0, 0, (f2*)buf); }, reps);
    printf("\"tile_8B_GBps\": %.0f, ", bytes / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL(k_tile_16B, dim3(tiles), dim3(256), 0, 0, (f2*)buf); }, reps);
    printf("\"tile_16B_GBps\": %.0f, ", bytes / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL(k_tile_8B_w, dim3(tiles), dim3(256), 0, 0, (f2*)buf); }, reps);
    printf("\"tile_8B_wave_rows_GBps\": %.0f, ", bytes / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL(k_tile128, dim3(tiles / 2), dim3(512), 0, 0, (f2*)buf); }, reps);
    printf("\"tile128_16B_GBps\": %.0f, ", bytes / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL(k_fill16, dim3(256 * 16), dim3(256), 0, 0, (f4*)buf, bytes / 16); }, reps);
    printf("\"fill16_again_GBps\": %.0f}\n", bytes / t / 1e9);
    CHECK(hipFree(buf));
    return 0;
}
