// What can k_runs_extract's ACCESS PATTERN reach?  One workgroup streams one 90 KB bit-packed vector (2 h at 100 Hz) front to
// back, sweep by sweep, 65 536 vectors back to back (5.9 GB: far beyond the 256 MB Infinity Cache) -- with only a popcount
// per word, the next sweep's loads in flight, optionally one barrier per sweep (the block scan's), for workgroups of 256 /
// 512 / 1024 threads and 2 or 4 16-byte loads per thread and sweep; next to it the same bytes read as ONE linear stream
// (grid-stride, the pattern of k_vad_energy).  Prints TB/s.  If the per-vector pattern itself stops near 0.6 of 8 TB/s, the
// kernel's instruction count is not what bounds it (round 6 halved it and the time did not move).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o profiles/_bin/extract_pattern_ceiling profiles/extract_pattern_ceiling.hip
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

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int THREADS, int G, bool BARRIER, bool NT>
__global__ __launch_bounds__(THREADS) void k_vec(const unsigned* __restrict__ data, int words_per_vec, int stride_words, unsigned* __restrict__ out) {
    const unsigned* w = data + (size_t)blockIdx.x * stride_words;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(w), 0, words_per_vec * 4, 0x00020000);
    constexpr int SWEEP = THREADS * 4 * G;
    __shared__ unsigned s_t[2][16];
    v4u xn[G];
    auto request = [&](int base) {
#pragma unroll
        for (int g = 0; g < G; ++g) xn[g] = __builtin_amdgcn_raw_buffer_load_b128(rs, (base + (g * THREADS + (int)threadIdx.x) * 4) * 4, 0, NT ? 2 : 0);
    };
    request(0);
    unsigned acc = 0;
    int buf = 0;
    for (int base = 0; base < words_per_vec; base += SWEEP, buf ^= 1) {
        v4u x[G];
#pragma unroll
        for (int g = 0; g < G; ++g) x[g] = xn[g];
        if (base + SWEEP < words_per_vec) request(base + SWEEP);
        unsigned c = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) c += __popc(x[g].x) + __popc(x[g].y) + __popc(x[g].z) + __popc(x[g].w);
        if (BARRIER) {
            if ((threadIdx.x & 63) == 63) s_t[buf][threadIdx.x >> 6] = c;
            __syncthreads();
            c += s_t[buf][0];
        }
        acc += c;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int U>
__global__ __launch_bounds__(256) void k_linear(const v4u* __restrict__ p, long long nvec, unsigned* __restrict__ out) {
    const long long stride = (long long)gridDim.x * 256 * U;
    unsigned acc = 0;
    for (long long i = (long long)blockIdx.x * 256 * U + threadIdx.x; i < nvec; i += stride) {
        v4u wv[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const long long j = i + (long long)k * 256;
            wv[k] = j < nvec ? p[j] : v4u{0, 0, 0, 0};
        }
#pragma unroll
        for (int k = 0; k < U; ++k) acc += __popc(wv[k].x) + __popc(wv[k].y) + __popc(wv[k].z) + __popc(wv[k].w);
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

static float time_ms(void (*launch)(void*), void* ctx) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch(ctx);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) launch(ctx);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 3;
}

struct Ctx {
    const unsigned* data;
    unsigned* out;
    int n_vec, words, stride;
};

template <int THREADS, int G, bool BARRIER, bool NT>
static void launch_vec(void* c) {
    Ctx* x = (Ctx*)c;
    hipLaunchKernelGGL((k_vec<THREADS, G, BARRIER, NT>), dim3(x->n_vec), dim3(THREADS), 0, 0, x->data, x->words, x->stride, x->out);
}
template <int U>
static void launch_lin(void* c) {
    Ctx* x = (Ctx*)c;
    hipLaunchKernelGGL((k_linear<U>), dim3(256 * 8), dim3(256), 0, 0, (const v4u*)x->data, (long long)x->n_vec * x->stride / 4, x->out);
}

int main() {
    const int n_vec = 65536, words = 22500, stride = 22528;  // 720 000 samples, vectors 64-byte aligned
    unsigned *data, *out;
    const size_t bytes = (size_t)n_vec * stride * 4;
    CHECK(hipMalloc(&data, bytes));
    CHECK(hipMalloc(&out, n_vec * 4));
    CHECK(hipMemset(data, 0x5a, bytes));
    Ctx c{data, out, n_vec, words, stride};
    const double gb = (double)n_vec * words * 4 / 1e9;
#define ROW(NAME, FN) printf(" \"%s\": %.3f,\n", NAME, gb / time_ms(FN, &c))
    printf("{\"unit\": \"TB/s of vector words read (65 536 vectors of 90 000 bytes)\",\n");
    ROW("per_vector_256_threads_2_loads_barrier", (launch_vec<256, 2, true, false>));
    ROW("per_vector_256_threads_2_loads_no_barrier", (launch_vec<256, 2, false, false>));
    ROW("per_vector_256_threads_2_loads_barrier_nontemporal", (launch_vec<256, 2, true, true>));
    ROW("per_vector_256_threads_4_loads_barrier", (launch_vec<256, 4, true, false>));
    ROW("per_vector_512_threads_2_loads_barrier", (launch_vec<512, 2, true, false>));
    ROW("per_vector_512_threads_4_loads_barrier", (launch_vec<512, 4, true, false>));
    ROW("per_vector_1024_threads_2_loads_barrier", (launch_vec<1024, 2, true, false>));
    ROW("per_vector_1024_threads_4_loads_barrier_nontemporal", (launch_vec<1024, 4, true, true>));
    ROW("linear_stream_4_loads", (launch_lin<4>));
    printf(" \"linear_stream_8_loads\": %.3f}\n", gb / time_ms(launch_lin<8>, &c));
    return 0;
}
