// ffs_fft.h -- register-radix Stockham FFT building blocks for gfx950 (wave64).
//
// Every thread owns 16 complex fp32 values in VGPRs.  A transform of length L (16..4096) is
// computed by L/16 cooperating threads in at most three stages of radix <= 16:
//     L = 16 * R1 * R2,  R1 = min(16, L/16),  R2 = L / (16*R1)
// with one LDS exchange between consecutive stages.  Before and after the transform, thread u
// holds the elements  u + (L/16)*q  (q = 0..15)  in register slot q, so global loads/stores of
// consecutive threads touch consecutive elements and two transforms can be chained (as the
// mid pass does) without an exchange in between.
//
// Index math is the textbook Stockham autosort DIT step; it is mirrored line by line by the
// numpy model in oracle/fft_model.py (stockham_fft), which tests check against numpy.fft.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ffsa {

typedef float2 cf;
#define FFS_DEV __device__ __forceinline__

FFS_DEV cf mk(float x, float y) { return make_float2(x, y); }
FFS_DEV cf cadd(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
FFS_DEV cf csub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
FFS_DEV cf cmul(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
FFS_DEV cf cmul_negi(cf a) { return mk(a.y, -a.x); }  // a * (-i)

// ---- forward DFT butterflies on registers --------------------------------------------------
// run(t): in-place DFT of size R; afterwards output bin k sits in t[slot_of(k)].
template <int R>
struct Bfly;

template <>
struct Bfly<1> {
    static FFS_DEV void run(cf*) {}
    static constexpr int slot_of(int k) { return k; }
};

template <>
struct Bfly<2> {
    static FFS_DEV void run(cf* t) {
        cf a = t[0], b = t[1];
        t[0] = cadd(a, b);
        t[1] = csub(a, b);
    }
    static constexpr int slot_of(int k) { return k; }
};

FFS_DEV void dft4(cf& x0, cf& x1, cf& x2, cf& x3) {
    cf a0 = cadd(x0, x2), a1 = csub(x0, x2);
    cf a2 = cadd(x1, x3), a3 = cmul_negi(csub(x1, x3));
    x0 = cadd(a0, a2);
    x2 = csub(a0, a2);
    x1 = cadd(a1, a3);
    x3 = csub(a1, a3);
}

template <>
struct Bfly<4> {
    static FFS_DEV void run(cf* t) { dft4(t[0], t[1], t[2], t[3]); }
    static constexpr int slot_of(int k) { return k; }
};

#define FFS_SQRT_HALF 0.70710678118654752440f
#define FFS_COS_PI_8 0.92387953251128675613f
#define FFS_SIN_PI_8 0.38268343236508977173f

template <>
struct Bfly<8> {
    // n = 4*n1 + n2, k = k1 + 2*k2: size-2 DFTs over n1, twiddle W8^(n2*k1), size-4 DFTs over n2.
    static FFS_DEV void run(cf* t) {
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
            cf a = t[n2], b = t[n2 + 4];
            t[n2] = cadd(a, b);
            t[n2 + 4] = csub(a, b);
        }
        t[5] = cmul(t[5], mk(FFS_SQRT_HALF, -FFS_SQRT_HALF));
        t[6] = cmul_negi(t[6]);
        t[7] = cmul(t[7], mk(-FFS_SQRT_HALF, -FFS_SQRT_HALF));
        dft4(t[0], t[1], t[2], t[3]);
        dft4(t[4], t[5], t[6], t[7]);
    }
    // X[k1 + 2*k2] is left in slot 4*k1 + k2
    static constexpr int slot_of(int k) { return 4 * (k & 1) + (k >> 1); }
};

template <>
struct Bfly<16> {
    // n = 4*n1 + n2, k = k1 + 4*k2: DFT4 over n1, twiddle W16^(n2*k1), DFT4 over n2.
    static FFS_DEV void run(cf* t) {
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) dft4(t[n2], t[n2 + 4], t[n2 + 8], t[n2 + 12]);
        // slot n2 + 4*k1 now holds A[k1][n2]
        t[5] = cmul(t[5], mk(FFS_COS_PI_8, -FFS_SIN_PI_8));       // W16^1
        t[6] = cmul(t[6], mk(FFS_SQRT_HALF, -FFS_SQRT_HALF));     // W16^2
        t[7] = cmul(t[7], mk(FFS_SIN_PI_8, -FFS_COS_PI_8));       // W16^3
        t[9] = cmul(t[9], mk(FFS_SQRT_HALF, -FFS_SQRT_HALF));     // W16^2
        t[10] = cmul_negi(t[10]);                                  // W16^4
        t[11] = cmul(t[11], mk(-FFS_SQRT_HALF, -FFS_SQRT_HALF));  // W16^6
        t[13] = cmul(t[13], mk(FFS_SIN_PI_8, -FFS_COS_PI_8));     // W16^3
        t[14] = cmul(t[14], mk(-FFS_SQRT_HALF, -FFS_SQRT_HALF));  // W16^6
        t[15] = cmul(t[15], mk(-FFS_COS_PI_8, FFS_SIN_PI_8));     // W16^9
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) dft4(t[4 * k1], t[4 * k1 + 1], t[4 * k1 + 2], t[4 * k1 + 3]);
    }
    // X[k1 + 4*k2] is left in slot 4*k1 + k2
    static constexpr int slot_of(int k) { return 4 * (k & 3) + (k >> 2); }
};

// ---- transform shape ------------------------------------------------------------------------
template <int L>
struct Shape {
    static_assert(L >= 16 && L <= 4096 && (L & (L - 1)) == 0, "L must be a power of two in [16, 4096]");
    static constexpr int LT = L / 16;                      // threads per transform
    static constexpr int R1 = (LT >= 16) ? 16 : LT;        // second-stage radix (1 = absent)
    static constexpr int R2 = L / (16 * R1);               // third-stage radix (1 = absent)
    static constexpr int TW1 = 0;                          // offset of stage-1 table [R1][16]
    static constexpr int TW2 = R1 * 16;                    // offset of stage-2 table [R2][256]
    static constexpr int TW_TOTAL = R1 * 16 + (R2 > 1 ? R2 * 256 : 0);
};

// One Stockham stage on registers: NB = 16/R butterflies per thread; butterfly b works on
// register slots b + r*NB and has butterfly index j = u + (L/16)*b.  tw is laid out [r][j % NS].
template <int L, int R, int NS, bool TWIDDLE>
FFS_DEV void stage_compute(cf (&v)[16], int u, const cf* __restrict__ tw) {
    constexpr int NB = 16 / R;
    constexpr int LT = L / 16;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        cf t[R];
        const int jm = (u + LT * b) & (NS - 1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            t[r] = v[b + r * NB];
            if (TWIDDLE && r > 0) t[r] = cmul(t[r], tw[r * NS + jm]);
        }
        Bfly<R>::run(t);
#pragma unroll
        for (int r = 0; r < R; ++r) v[b + r * NB] = t[Bfly<R>::slot_of(r)];
    }
}

// Scatter stage outputs to their Stockham positions: element (b, r) -> (j/NS)*NS*R + j%NS + r*NS.
template <int L, int R, int NS, class Addr>
FFS_DEV void stage_scatter(const cf (&v)[16], cf* lds, int u, const Addr& addr) {
    constexpr int NB = 16 / R;
    constexpr int LT = L / 16;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = u + LT * b;
        const int base = (j / NS) * (NS * R) + (j & (NS - 1));
#pragma unroll
        for (int r = 0; r < R; ++r) lds[addr(base + r * NS)] = v[b + r * NB];
    }
}

template <int L, class Addr>
FFS_DEV void stage_gather(cf (&v)[16], const cf* lds, int u, const Addr& addr) {
    constexpr int LT = L / 16;
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = lds[addr(u + LT * q)];
}

// Forward DFT of length L over the LT threads that share `addr`'s LDS region.
// In: v[q] = x[u + LT*q].  Out: v[q] = X[u + LT*q].  All threads of the block must call it
// (it contains __syncthreads()).  tw = Shape<L> stage tables.
// Keep the (uniform) table pointer opaque at this program point: the twiddle loads that depend on
// it cannot be hoisted above it, which bounds how many table values are live at once.
FFS_DEV const cf* pin(const cf* p) {
    asm volatile("" : "+s"(p));
    return p;
}

template <int L, class Addr>
FFS_DEV void fft_regs(cf (&v)[16], cf* lds, int u, const Addr& addr, const cf* __restrict__ tw) {
    typedef Shape<L> S;
    stage_compute<L, 16, 1, false>(v, u, nullptr);
    if constexpr (S::R1 > 1) {
        __syncthreads();  // previous readers of this LDS region are done
        const cf* tw1 = pin(tw + S::TW1);  // stage-1 table loads overlap the exchange
        stage_scatter<L, 16, 1>(v, lds, u, addr);
        __syncthreads();
        stage_gather<L>(v, lds, u, addr);
        stage_compute<L, S::R1, 16, true>(v, u, tw1);
        if constexpr (S::R2 > 1) {
            __syncthreads();
            const cf* tw2 = pin(tw + S::TW2);
            stage_scatter<L, S::R1, 16>(v, lds, u, addr);
            __syncthreads();
            stage_gather<L>(v, lds, u, addr);
            stage_compute<L, S::R2, 256, true>(v, u, tw2);
        }
    }
}

// LDS addressing for a tile of C interleaved column transforms: element p of column c.
template <int C>
struct ColAddr {
    int c;
    FFS_DEV int operator()(int p) const { return p * C + c; }
};

// LDS addressing for one row transform per LT threads; one pad element per 32 keeps the
// stride-16 scatter of the first stage off a single bank.
struct RowAddr {
    int base;
    FFS_DEV int operator()(int p) const { return base + p + (p >> 5); }
};

}  // namespace ffsa
