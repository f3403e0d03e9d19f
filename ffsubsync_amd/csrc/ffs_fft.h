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

#include <utility>

namespace ffsa {

// A complex fp32 value is one 64-bit VGPR pair; arithmetic uses the packed-FP32 VALU ops of gfx950
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), whose op_sel / neg modifiers express the complex
// swizzles for free: a complex multiply is 2 instructions, a radix-4 butterfly 8 (16 and 4+.. scalar).
typedef float cf __attribute__((ext_vector_type(2)));
#define FFS_DEV __device__ __forceinline__
#define FFS_HD __host__ __device__ __forceinline__

FFS_HD cf mk(float x, float y) {
    cf r = {x, y};
    return r;
}
FFS_DEV cf cadd(cf a, cf b) { return a + b; }
FFS_DEV cf csub(cf a, cf b) { return a - b; }
// (a.x*b.x - a.y*b.y, a.x*b.y + a.y*b.x)
// one asm statement per complex multiply: the compiler's hazard recogniser treats every inline-asm VGPR result as a
// possible dst_sel forwarding hazard and puts an s_nop between two dependent statements; packed-FP32 ops write whole
// registers (no dst_sel), so inside one statement the pair can issue back to back
FFS_DEV cf cmul(cf a, cf b) {
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(r)
        : "v"(a), "v"(b));
    return r;
}
// same with a wave-uniform second factor (compile-time twiddle constants live in an SGPR pair)
FFS_DEV cf cmul_k(cf a, float kx, float ky) {
    const cf b = {kx, ky};
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(r)
        : "v"(a), "s"(b));
    return r;
}
// acc + a*b: two packed FMAs
FFS_DEV cf cmac(cf acc, cf a, cf b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "+v"(acc)
        : "v"(a), "v"(b));
    return acc;
}
// the same as a side-effecting statement: cannot be speculated, so `if (uniform) acc = cmac_v(...)` stays a branch
// instead of becoming compute-both-and-select (which doubles the live accumulators)
FFS_DEV cf cmac_v(cf acc, cf a, cf b) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
                 : "+v"(acc)
                 : "v"(a), "v"(b));
    return acc;
}
// acc + a*k with a wave-uniform k (SGPR pair): two packed FMAs
FFS_DEV cf cmac_k(cf acc, cf a, float kx, float ky) {
    const cf b = {kx, ky};
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "+v"(acc)
        : "v"(a), "s"(b));
    return acc;
}
// a + (-i)*t = (a.x + t.y, a.y - t.x)   and   a - (-i)*t = (a.x - t.y, a.y + t.x)
FFS_DEV cf add_negi(cf a, cf t) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(t));
    return r;
}
FFS_DEV cf sub_negi(cf a, cf t) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(t));
    return r;
}
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
    const cf a0 = cadd(x0, x2), a1 = csub(x0, x2);
    const cf a2 = cadd(x1, x3), t = csub(x1, x3);
    x0 = cadd(a0, a2);
    x2 = csub(a0, a2);
    x1 = add_negi(a1, t);
    x3 = sub_negi(a1, t);
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
        t[5] = cmul_k(t[5], FFS_SQRT_HALF, -FFS_SQRT_HALF);
        t[6] = cmul_negi(t[6]);
        t[7] = cmul_k(t[7], -FFS_SQRT_HALF, -FFS_SQRT_HALF);
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
        t[5] = cmul_k(t[5], FFS_COS_PI_8, -FFS_SIN_PI_8);       // W16^1
        t[6] = cmul_k(t[6], FFS_SQRT_HALF, -FFS_SQRT_HALF);     // W16^2
        t[7] = cmul_k(t[7], FFS_SIN_PI_8, -FFS_COS_PI_8);       // W16^3
        t[9] = cmul_k(t[9], FFS_SQRT_HALF, -FFS_SQRT_HALF);     // W16^2
        t[10] = cmul_negi(t[10]);                                  // W16^4
        t[11] = cmul_k(t[11], -FFS_SQRT_HALF, -FFS_SQRT_HALF);  // W16^6
        t[13] = cmul_k(t[13], FFS_SIN_PI_8, -FFS_COS_PI_8);     // W16^3
        t[14] = cmul_k(t[14], -FFS_SQRT_HALF, -FFS_SQRT_HALF);  // W16^6
        t[15] = cmul_k(t[15], -FFS_COS_PI_8, FFS_SIN_PI_8);     // W16^9
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
    // Stage tables (host: make_stage_tables).  A radix-16 stage stores only the six powers
    // w^1, w^2, w^3, w^4, w^8, w^12 per butterfly index ([6][NS]); smaller radices store w^r, [R][NS].
    static constexpr int ROWS1 = (R1 == 16) ? 6 : R1;
    static constexpr int ROWS2 = (R2 == 16) ? 6 : R2;
    static constexpr int TW1 = 0;
    static constexpr int TW2 = ROWS1 * 16;
};

// Per-thread twiddle registers of one twiddled stage.  For a fixed thread the butterfly indices
// j = u + LT*b never change, so the values are loaded once, up front, and reused by every
// transform the thread takes part in.
// SC: the caller's u is wave-uniform (column tiles of 64+ columns: a wave is one row phase), so the values are the
// same for the whole wave -- scalar loads into SGPRs, used as the scalar operand of the packed multiplies (no vector
// loads, no VGPRs; only for stages of radix < 16).
template <int L, int R, int NS, bool SC = false>
struct StageTw {
    static constexpr int NB = 16 / R;
    static constexpr int COUNT = (R == 16) ? 6 : (R > 1 ? (R - 1) * NB : 1);
    static_assert(!SC || R < 16, "scalar stage twiddles: radix < 16 only");
    cf w[COUNT];
    FFS_DEV void load(const cf* __restrict__ tab, int u) {
        constexpr int LT = L / 16;
        if constexpr (R == 16) {
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = tab[i * NS + (u & (NS - 1))];
        } else if constexpr (R > 1) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 1; r < R; ++r) w[b * (R - 1) + (r - 1)] = tab[r * NS + ((u + LT * b) & (NS - 1))];
        }
    }
};

template <int L, bool SC = false>
struct TwRegs {
    typedef Shape<L> S;
    StageTw<L, S::R1, 16, SC> s1;
    StageTw<L, S::R2, 256, SC> s2;
    FFS_DEV void load(const cf* __restrict__ tw, int u) {
        if constexpr (S::R1 > 1) s1.load(tw + S::TW1, u);
        if constexpr (S::R2 > 1) s2.load(tw + S::TW2, u);
    }
};

// Radix-16 butterfly with the stage twiddles w^r folded in as w^(4a) * w^b (r = 4a + b):
// inputs t[4a+b] *= w^(4a), DFT4 over a, outputs *= w^b and the constant W16^(b*k1), DFT4 over b.
// Six per-thread twiddles instead of fifteen (24 complex multiplies instead of 15).
FFS_DEV void bfly16_twiddled(cf* t, const cf* w /* w1 w2 w3 w4 w8 w12 */) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        t[4 + b] = cmul(t[4 + b], w[3]);
        t[8 + b] = cmul(t[8 + b], w[4]);
        t[12 + b] = cmul(t[12 + b], w[5]);
        dft4(t[b], t[4 + b], t[8 + b], t[12 + b]);
    }
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

// One Stockham stage on registers: NB = 16/R butterflies per thread; butterfly b works on
// register slots b + r*NB and has butterfly index j = u + (L/16)*b.
template <int L, int R, int NS, bool SC>
FFS_DEV void stage_compute(cf (&v)[16], const StageTw<L, R, NS, SC>& tw) {
    constexpr int NB = 16 / R;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        cf t[R];
#pragma unroll
        for (int r = 0; r < R; ++r) t[r] = v[b + r * NB];
        if constexpr (R == 16) {
            bfly16_twiddled(t, tw.w);
        } else {
#pragma unroll
            for (int r = 1; r < R; ++r) {
                const cf w = tw.w[b * (R - 1) + (r - 1)];
                t[r] = SC ? cmul_k(t[r], w.x, w.y) : cmul(t[r], w);
            }
            Bfly<R>::run(t);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) v[b + r * NB] = t[Bfly<R>::slot_of(r)];
    }
}

// First stage: one untwiddled radix-16 butterfly per thread.
FFS_DEV void stage_first(cf (&v)[16]) {
    cf t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = v[r];
    Bfly<16>::run(t);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = t[Bfly<16>::slot_of(r)];
}

// LDS addressing.  Every access of the transform touches positions of one of two shapes:
//   gather : p = u + LT*q                       (q compile-time)
//   scatter: p = (j/NS)*NS*R + j%NS + r*NS      (j = u + LT*b; b, r compile-time)
// An Addr type turns them into LDS element indices as  thread-dependent base + compile-time
// constant, so each ds_read/ds_write uses one base VGPR and an immediate offset.

// Tile of C interleaved column transforms (column c of the block): index = p*C + c.
template <int L, int C>
struct ColAddr {
    int uc;  // u*C + c
    int c;
    FFS_DEV ColAddr(int u, int c_) : uc(u * C + c_), c(c_) {}
    FFS_DEV void refresh() {}
    template <int Q>
    FFS_DEV int gather() const {
        return uc + (L / 16) * Q * C;
    }
    template <int R, int NS, int B, int RR>
    FFS_DEV int scatter(int u) const {
        constexpr int LT = L / 16;
        if constexpr (NS == 1) {
            // p = j*R + r with j = u + LT*B:  p*C + c = (uc - c)*R + c + (LT*B*R + r)*C
            return (uc - c) * R + c + (LT * B * R + RR) * C;
        } else {
            const int j = u + LT * B;
            return ((j / NS) * (NS * R) + (j & (NS - 1)) + RR * NS) * C + c;
        }
    }
};

// One row transform per LT threads, padded by one element per 16 (index p + p/16): conflict-free
// for the stride-16 scatter of the first stage (ds_write_b64 serves 16 lanes per cycle) and
// separable into base + constant because LT is a multiple of 16 for every row length >= 256.
template <int L>
struct RowAddr {
    static_assert(L >= 256, "row transforms need L >= 256");
    static constexpr int LT = L / 16;
    static constexpr int ROW_ELEMS = L + L / 16;
    int gbase;   // row_base + u + u/16
    int s0base;  // row_base + 17*u
    int row_base, u_lo, u_hi;
    FFS_DEV RowAddr(int row_base_, int u) : row_base(row_base_), u_lo(u & 15), u_hi(u >> 4) {
        gbase = row_base + u + (u >> 4);
        s0base = row_base + 17 * u;
    }
    FFS_DEV void refresh() {}
    FFS_DEV int at(int p) const { return row_base + p + (p >> 4); }
    // position L-1-(u + LT*Q) = (LT-1-u) + LT*(15-Q): the mirrored element of this thread's slot Q
    template <int Q>
    FFS_DEV int gather_mirror() const {
        const int um = LT - 1 - (16 * u_hi + u_lo);
        return row_base + um + (um >> 4) + LT * (15 - Q) + (LT / 16) * (15 - Q);
    }
    template <int Q>
    FFS_DEV int gather() const {
        return gbase + LT * Q + (LT / 16) * Q;
    }
    template <int R, int NS, int B, int RR>
    FFS_DEV int scatter(int) const {
        if constexpr (NS == 1) {
            static_assert(B == 0 && R == 16, "first stage is one radix-16 butterfly per thread");
            return s0base + RR;  // p = 16u + r  ->  p + p/16 = 17u + r
        } else if constexpr (NS == 16) {
            // j = u + LT*B, j/16 = u_hi + (LT/16)*B, j%16 = u_lo; p = (j/16)*16R + j%16 + 16r
            // p/16 = (j/16)*R + r
            return row_base + u_lo + u_hi * (17 * R) + (LT / 16) * B * (17 * R) + 17 * RR;
        } else {
            static_assert(NS == 256, "unexpected stage");
            // j = u + LT*B, p = (j/256)*256R + j%256 + 256r; LT is a multiple of 16 so p/16 splits
            const int j_hi = (u_hi + (LT / 16) * B) >> 4;              // j / 256
            const int j_mid = (u_hi + (LT / 16) * B) & 15;             // (j / 16) % 16
            return row_base + u_lo + j_mid * 17 + j_hi * (272 * R) + 272 * RR;
        }
    }
};

template <int L, int R, int NS, int B, class Addr, int... RR>
FFS_DEV void scatter_row(const cf (&v)[16], cf* lds, int u, const Addr& addr, std::integer_sequence<int, RR...>) {
    constexpr int NB = 16 / R;
    ((lds[addr.template scatter<R, NS, B, RR>(u)] = v[B + RR * NB]), ...);
}
template <int L, int R, int NS, class Addr, int... B>
FFS_DEV void stage_scatter_impl(const cf (&v)[16], cf* lds, int u, const Addr& addr, std::integer_sequence<int, B...>) {
    (scatter_row<L, R, NS, B>(v, lds, u, addr, std::make_integer_sequence<int, R>{}), ...);
}
// Scatter stage outputs to their Stockham positions: element (b, r) -> (j/NS)*NS*R + j%NS + r*NS.
template <int L, int R, int NS, class Addr>
FFS_DEV void stage_scatter(const cf (&v)[16], cf* lds, int u, const Addr& addr) {
    stage_scatter_impl<L, R, NS>(v, lds, u, addr, std::make_integer_sequence<int, 16 / R>{});
}

template <class Addr, int... Q>
FFS_DEV void stage_gather_impl(cf (&v)[16], const cf* lds, const Addr& addr, std::integer_sequence<int, Q...>) {
    ((v[Q] = lds[addr.template gather<Q>()]), ...);
}
template <int L, class Addr>
FFS_DEV void stage_gather(cf (&v)[16], const cf* lds, int, const Addr& addr) {
    stage_gather_impl(v, lds, addr, std::make_integer_sequence<int, 16>{});
}

// Workgroup barrier that only waits for this wave's LDS traffic.  __syncthreads() also drains the wave's
// outstanding GLOBAL loads (s_waitcnt vmcnt(0) in front of s_barrier), which defeats software prefetching:
// loads issued for the next row would be waited for at the first exchange of the current transform.  The
// exchanges of a transform only communicate through LDS, so lgkmcnt(0) is all they need.
FFS_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <bool LB>
FFS_DEV void block_sync() {
    if constexpr (LB)
        lds_barrier();
    else
        __syncthreads();
}

// Forward DFT of length L over the LT threads that share `addr`'s LDS region.
// In: v[q] = x[u + LT*q].  Out: v[q] = X[u + LT*q].  All threads of the block must call it
// (it contains __syncthreads()).  tw = this thread's preloaded stage twiddles.
// LB: barriers wait for LDS traffic only (lds_barrier), so global loads issued before the call stay in flight.
template <int L, class Addr, bool LB = false, bool SC = false>
FFS_DEV void fft_regs(cf (&v)[16], cf* lds, int u, Addr& addr, const TwRegs<L, SC>& tw) {
    typedef Shape<L> S;
    auto barrier = [] { block_sync<LB>(); };
    addr.refresh();
    stage_first(v);
    if constexpr (S::R1 > 1) {
        barrier();  // previous readers of this LDS region are done
        stage_scatter<L, 16, 1>(v, lds, u, addr);
        barrier();
        stage_gather<L>(v, lds, u, addr);
        stage_compute<L, S::R1, 16, SC>(v, tw.s1);
        if constexpr (S::R2 > 1) {
            barrier();
            stage_scatter<L, S::R1, 16>(v, lds, u, addr);
            barrier();
            stage_gather<L>(v, lds, u, addr);
            stage_compute<L, S::R2, 256, SC>(v, tw.s2);
        }
    }
}

// ---- column transforms of length 3 * 2^k ----------------------------------------------------
// A column transform of length L = 3*LI (LI a power of two) is three interleaved length-LI transforms
// (decimation in time) and one radix-3 combine:
//     F_g[k'] = sum_j x[3j + g] W_LI^(j k')                       g = 0..2, k' < LI
//     X[k' + LI*r] = sum_g W_L^(g k') W_3^(g r) F_g[k']           r = 0..2
// Thread u12 (= 3u + g) of a column holds x[u12 + (L/16)*q] = x[3(u + LTI*q) + g], i.e. exactly the
// inputs of sub-transform g in fft_regs' register layout -- the loads are the same as for a
// power-of-two column.  After the sub-transforms the three groups swap their (twiddled) F_g through
// LDS and thread (u, g) produces the outputs of r = g:  X[(u + LI*g) + LTI*q].
template <int L>
struct ColShape {
    static constexpr bool R3 = (L % 3 == 0);
    static constexpr int LI = R3 ? L / 3 : L;  // power-of-two transform length
    static constexpr int LT = L / 16;          // threads per column
    static constexpr int LTI = LI / 16;        // threads per power-of-two transform
    static constexpr int OSTEP = LTI;          // output index = out_base(u12) + OSTEP*q
    static FFS_DEV int out_base(int u12) { return R3 ? (u12 / 3) + LI * (u12 % 3) : u12; }
};

#define FFS_SQRT3_HALF 0.86602540378443864676f

// First half of a length-3*LI column transform: the three sub-transforms, twiddled, left in LDS as
// F'[g][k'][c] = W_L^(g k') F_g[k'] at lds[(g*LI + k')*C + c] (followed by a barrier).
template <int L, int C, bool LB = false>
FFS_DEV void col_fft3_front(cf (&v)[16], cf* lds, int u12, int c, const TwRegs<ColShape<L>::LI>& twr,
                            const cf* __restrict__ tw3) {
    constexpr int LI = ColShape<L>::LI, LTI = ColShape<L>::LTI;
    const int u = u12 / 3, g = u12 % 3;
    ColAddr<LI, C> addr(u, c);
    fft_regs<LI, ColAddr<LI, C>, LB>(v, lds + g * (LI * C), u, addr, twr);
    // W_L^(g k') from the block's LDS copy of the table (a global load issued up front would pin
    // sixteen register pairs across the whole sub-transform)
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = cmul(v[q], tw3[g * (u + LTI * q)]);
    block_sync<LB>();
#pragma unroll
    for (int q = 0; q < 16; ++q) lds[(g * LI + u + LTI * q) * C + c] = v[q];
    block_sync<LB>();
}

// X_r = a + W_3^r b + W_3^(2r) cc  with W_3 = -1/2 - i*sqrt(3)/2:  a + alpha*(b + cc) + beta*(-i)*(b - cc),
// (alpha, beta) = (1, 0), (-1/2, sqrt3/2), (-1/2, -sqrt3/2) for r = 0, 1, 2
FFS_DEV cf radix3_out(cf a, cf b, cf cc, float alpha, float beta) {
    const cf t = cadd(b, cc), d = csub(b, cc);
    return mk(a.x + alpha * t.x + beta * d.y, a.y + alpha * t.y - beta * d.x);
}

// Forward DFT of one column of a C-column tile.  In: v[q] = x[u12 + LT*q].  Out: v[q] =
// X[out_base(u12) + OSTEP*q].  tw3 = W_L^k (k < L) in LDS, outside the L*C elements at `lds`; used only
// for L = 3*LI.
template <int L, int C, bool LB = false, bool SC = false>
FFS_DEV void col_fft(cf (&v)[16], cf* lds, int u12, int c, const TwRegs<ColShape<L>::LI, SC>& twr,
                     const cf* __restrict__ tw3) {
    typedef ColShape<L> CS;
    if constexpr (!CS::R3) {
        ColAddr<L, C> addr(u12, c);
        fft_regs<L, ColAddr<L, C>, LB, SC>(v, lds, u12, addr, twr);
    } else {
        static_assert(!SC, "scalar stage twiddles: power-of-two columns only");
        constexpr int LI = CS::LI, LTI = CS::LTI;
        const int u = u12 / 3, g = u12 % 3;
        col_fft3_front<L, C, LB>(v, lds, u12, c, twr, tw3);
        const float alpha = (g == 0) ? 1.0f : -0.5f;
        const float beta = (g == 0) ? 0.0f : (g == 1 ? FFS_SQRT3_HALF : -FFS_SQRT3_HALF);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = (u + LTI * q) * C + c;
            v[q] = radix3_out(lds[k], lds[LI * C + k], lds[2 * LI * C + k], alpha, beta);
        }
    }
}

// ---- column transforms of length 3*LI with all three sub-transforms in ONE thread ---------------------------
// col_fft's radix-3 path spreads the three interleaved sub-transforms over three threads and combines them through
// LDS: 112 LDS operations per 16 values (the sub-transform's exchange, the twiddle table, 16 writes, 48 reads) against
// 32 for a power-of-two column -- the windowless plans' pass A and last pass were LDS-bound at ~2 TB/s.  Here thread
// u (of LTI = LI/16 per column) holds x[3(u + LTI*q) + g] for all g: 48 values.  The three sub-transforms run one
// after the other through the same LI*C-element LDS tile (exactly a power-of-two column of length LI), then
//     X[k' + LI*r] = F_0[k'] + W_3^r W_L^k' F_1[k'] + W_3^2r W_L^2k' F_2[k'],      k' = u + LTI*q
// is register arithmetic.  W_L^k' = W_L^u * W_48^q (L = 48*LTI): one per-thread value and compile-time constants.
// In:  v[g][q] = x[3(u + LTI*q) + g].   Out: v[r][q] = X[u + LTI*q + LI*r].
static constexpr float kW48[16][2] = {{1.000000000e+00f, 0.000000000e+00f}, {9.914448614e-01f, -1.305261922e-01f}, {9.659258263e-01f, -2.588190451e-01f}, {9.238795325e-01f, -3.826834324e-01f}, {8.660254038e-01f, -5.000000000e-01f}, {7.933533403e-01f, -6.087614290e-01f}, {7.071067812e-01f, -7.071067812e-01f}, {6.087614290e-01f, -7.933533403e-01f}, {5.000000000e-01f, -8.660254038e-01f}, {3.826834324e-01f, -9.238795325e-01f}, {2.588190451e-01f, -9.659258263e-01f}, {1.305261922e-01f, -9.914448614e-01f}, {0.000000000e+00f, -1.000000000e+00f}, {-1.305261922e-01f, -9.914448614e-01f}, {-2.588190451e-01f, -9.659258263e-01f}, {-3.826834324e-01f, -9.238795325e-01f}};
static constexpr float kW24[16][2] = {{1.000000000e+00f, 0.000000000e+00f}, {9.659258263e-01f, -2.588190451e-01f}, {8.660254038e-01f, -5.000000000e-01f}, {7.071067812e-01f, -7.071067812e-01f}, {5.000000000e-01f, -8.660254038e-01f}, {2.588190451e-01f, -9.659258263e-01f}, {0.000000000e+00f, -1.000000000e+00f}, {-2.588190451e-01f, -9.659258263e-01f}, {-5.000000000e-01f, -8.660254038e-01f}, {-7.071067812e-01f, -7.071067812e-01f}, {-8.660254038e-01f, -5.000000000e-01f}, {-9.659258263e-01f, -2.588190451e-01f}, {-1.000000000e+00f, 0.000000000e+00f}, {-9.659258263e-01f, 2.588190451e-01f}, {-8.660254038e-01f, 5.000000000e-01f}, {-7.071067812e-01f, 7.071067812e-01f}};

static constexpr float kW32[16][2] = {{1.000000000e+00f, 0.000000000e+00f}, {9.807852804e-01f, -1.950903220e-01f}, {9.238795325e-01f, -3.826834324e-01f}, {8.314696123e-01f, -5.555702330e-01f}, {7.071067812e-01f, -7.071067812e-01f}, {5.555702330e-01f, -8.314696123e-01f}, {3.826834324e-01f, -9.238795325e-01f}, {1.950903220e-01f, -9.807852804e-01f}, {0.000000000e+00f, -1.000000000e+00f}, {-1.950903220e-01f, -9.807852804e-01f}, {-3.826834324e-01f, -9.238795325e-01f}, {-5.555702330e-01f, -8.314696123e-01f}, {-7.071067812e-01f, -7.071067812e-01f}, {-8.314696123e-01f, -5.555702330e-01f}, {-9.238795325e-01f, -3.826834324e-01f}, {-9.807852804e-01f, -1.950903220e-01f}};

// NS = 3 (L = 3*LI, above) or NS = 2 (L = 2*LI: X[k'] = F_0 + W_L^k' F_1, X[k' + LI] = F_0 - W_L^k' F_1 with
// W_L^k' = W_L^u * W_32^q -- a 512-row column as two 256-row sub-transforms, one LDS exchange each, instead of the
// three-stage 16*16*2 column with two).
template <int NS, int LI, int C, bool LB = false>
FFS_DEV void colnr_fft(cf (&v)[NS][16], cf* lds, int u, int c, const TwRegs<LI>& twr, cf wu, cf wu2) {
    static_assert(NS == 2 || NS == 3, "two or three sub-transforms per thread");
    ColAddr<LI, C> addr(u, c);
#pragma unroll
    for (int g = 0; g < NS; ++g) fft_regs<LI, ColAddr<LI, C>, LB>(v[g], lds, u, addr, twr);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        if constexpr (NS == 2) {
            const cf t1 = cmul(cmul_k(v[1][q], kW32[q][0], kW32[q][1]), wu);
            const cf f0 = v[0][q];
            v[0][q] = cadd(f0, t1);
            v[1][q] = csub(f0, t1);
        } else {
            const cf t1 = cmul(cmul_k(v[1][q], kW48[q][0], kW48[q][1]), wu);
            const cf t2 = cmul(cmul_k(v[2][q], kW24[q][0], kW24[q][1]), wu2);
            const cf sum = cadd(t1, t2), dif = csub(t1, t2);
            const cf m = mk(v[0][q].x - 0.5f * sum.x, v[0][q].y - 0.5f * sum.y);
            const cf e = mk(FFS_SQRT3_HALF * dif.x, FFS_SQRT3_HALF * dif.y);
            v[0][q] = cadd(v[0][q], sum);
            v[1][q] = add_negi(m, e);  // W_3 = -1/2 - i sqrt(3)/2:  m - i*e
            v[2][q] = sub_negi(m, e);  //                            m + i*e
        }
    }
}

}  // namespace ffsa
