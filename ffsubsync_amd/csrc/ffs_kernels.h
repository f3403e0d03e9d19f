// ffs_kernels.h -- device kernels of the alignment hot path (gfx950).
//
// Cross-correlation of a reference activity vector with packed pairs of candidate vectors:
//
//   z = a' + i*b'  (two candidates share one complex transform; x' = 2x-1, aligners.py:55-57)
//   out = FFT( FFT(z) * conj(FFT(r')) ) / N      =>  Re out[m] = sum_i a'[i] r'[i+m],
//                                                     Im out[m] = sum_i b'[i] r'[i+m]
//
// which is the reference's `convolve` (aligners.py:70-74) re-indexed by m = d mod N with
// d = N-1-S-k.  Both FFTs are forward transforms of length N = N1*N2 done in two HBM passes
// each (four-step), and the inner two passes fuse into one kernel:
//
//   k_pass_a : columns n2: load/±1-map/zero-pad, length-N1 FFT over n1, twiddle W_N^(n2*k1)
//              -> tile layout T[x/C][k1][x%C]                                  (write 8N bytes)
//   k_mid    : rows k1: length-N2 FFT over n2 (= spectrum row), * conj(R)/N held in registers,
//              length-N2 FFT over k2, twiddle W_N^(k1*m1) -> same tile layout, in place
//                                                                  (read 8N + write 8N bytes)
//   k_pass_c : columns m1: length-N1 FFT over k1 -> out[m1 + N2*m2]; lag-window mask and
//              per-block argmax nominees; the correlation itself is never written (read 8N)
//
// followed by tiny kernels that gather the nominees of each candidate, re-evaluate them exactly
// (integer counts / fp64) and take the max over candidates (aligners.py:154-167).
#pragma once
#include "ffs_fft.h"

namespace ffsa {

constexpr int KBLK = 6;   // nominees kept per pass-C block
constexpr int KNOM = 16;  // nominees kept per candidate
constexpr int RSEG = 16;  // blocks sharing each exact re-evaluation

constexpr int HALF_REF = 1;   // reference slot holds rows 0..N1/2 only
constexpr int HALF_LAST = 2;  // so does the last candidate slot (single real candidate)
constexpr int PAIR_ROWS = 8;  // block-segmented mid pass: mirror-row pairs on one XCD
// Section experiments (timing only, WRONG RESULTS): compiled in by `make lab` (-DFFS_LAB -> libffsalign_lab.so) and
// selected there through FFS_MID_DEBUG / FFS_PASS_A_DEBUG; the product library contains none of it -- FFS_LABF() is a
// constant false and the branches fold away.
constexpr int DBG_NO_STORE = 32768;                 // mid pass keeps its results (read-only kernel)
constexpr int DBG_NO_FFT = 256, DBG_HOT_MEM = 512;  // mid pass without row transforms / every pair on pair 0's buffers
// pass A: no stores / no input loads / unit twiddles instead of the table loads / every block stores into tile 0 of
// slot 0 (writes stay in L2) / every block reads transform 0's vectors (input reads become L2 hits)
constexpr int DBG_PA_NO_STORE = 1024, DBG_PA_NO_INPUT = 2048, DBG_PA_NO_TW = 4096, DBG_PA_HOT_STORE = 8192;
constexpr int DBG_PA_HOT_INPUT = 16384;
#ifdef FFS_LAB
#define FFS_LABF(flags, bit) (((flags) & (bit)) != 0)
#else
#define FFS_LABF(flags, bit) (false)
#endif
constexpr int STORE_8B = 4;   // pass A: plain 8-byte stores for 64-column tiles

struct XformDesc {  // one packed transform (slot 0 of a pair is the reference: b = a, len_b = 0)
    const void* a;
    const void* b;
    int32_t len_a, len_b;
    float a0, a1, b0, b1;  // u8: mapped sample values x' for byte == 0 / != 0
    // Samples [lead, len) of the transform input are a[lead .. len); positions below `lead` are zero
    // padding and `a` may point in front of the vector there (block-segmented mode: the stretch of the
    // reference that block 0 meets starts |d_lo| samples before the vector).  Never dereferenced.
    int32_t lead_a, lead_b;
    // bit-packed inputs (DT == 2) only: sample n of the transform input is bit off + n of the 32-bit
    // little-endian word array at a / b (bit i of the array = (word[i >> 5] >> (i & 31)) & 1); off may be
    // negative in front of `lead`.  Byte / float inputs move the pointer instead and leave these 0.
    int32_t off_a, off_b;
};

struct CandDesc {  // one candidate (one FFTAligner solve)
    const void* s;     // candidate samples
    const void* r;     // its reference
    int32_t S, R;
    int32_t d_lo, d_hi;  // inclusive lag window (already intersected with [-S, Nref-1-S])
    int32_t n_ref;       // the reference's transform length for (R, S)
    int32_t flags;       // bit 0 preset by the host: no lag for the transform pipeline (see CAND_*)
    float margin;        // fp32 tie margin for nominee collection
    int32_t d_zero;      // CAND_HAS_ZERO: largest lag of the reference's window with an empty overlap
    double s0, s1, r0, r1;  // mapped two-level values (fp64, as the reference computes them)
};

// CandDesc.flags.  The transforms only ever evaluate lags with a non-empty overlap, d in (-S, R): every
// other lag of the reference's window has c(d) = 0 exactly (its `convolve` entries there are fp64
// rounding noise around 0), so they are represented by ONE virtual nominee (score 0, lag d_zero = the
// largest such lag = the first such k, np.argmax's pick among equals) that k_finalize_cands compares
// with the best real lag.  This is what lets a windowless solve use a transform of length >= R+S-1
// instead of the reference's 2^ceil(log2(R+S)).
constexpr int CAND_NO_LAGS = 1;    // nothing for the transform / nominee kernels to do
constexpr int CAND_HAS_ZERO = 16;  // d_zero is valid
// Element types of the two vectors (FFS_DTYPE_* codes 0..3), preset by the host: bits 8..10 the candidate's, bits
// 12..14 the reference's.  Read only by the mixed-type (DT == 4) instantiations of the exact re-evaluation kernels --
// e.g. a four-level float64 reference from the weighted fused VAD (speech_transformers.py:290-293) against bit-packed
// subtitle rasters.
constexpr int CAND_DTS_SHIFT = 8, CAND_DTR_SHIFT = 12;

struct BlockNom {
    float bmax;
    int32_t cnt;
    float val[KBLK];
    int32_t d[KBLK];
};

struct NomList {
    int32_t count;
    int32_t flags;
    float gmax;
    int32_t pad;
    int32_t d[KNOM];
    float val[KNOM];
};

struct RescoreAcc {
    unsigned int n11, n1x, nx1, pad;  // two-level inputs: exact integer counts (atomics are exact)
    double part[RSEG];                // float inputs: one fp64 partial per segment, summed in order
};

struct CandResult {
    double score;
    long long offset;
    float score_f32;
    int32_t flags;
};
struct PairResult {
    double score;
    long long offset;
    int32_t best_cand;
    int32_t flags;
};

// Overflow pool: every in-window lag within the tie margin of a candidate whose nominee lists
// overflowed (flat-topped correlation peaks) is appended here and re-evaluated exactly.
struct PoolHeader {
    unsigned int count;     // entries appended (may exceed capacity)
    unsigned int capacity;
    unsigned int pad0, pad1;
};
struct PoolEntry {
    int32_t ci;
    int32_t d;
    float val;
    int32_t pad;
    double score;
};
constexpr int POOL_D_BIAS = 1 << 30;
struct PoolBest {  // per candidate
    unsigned long long key;  // order-preserving image of the best exact score (0 = none)
    int32_t d;               // largest lag attaining it, stored as d + POOL_D_BIAS (0 = none)
    float val;
    unsigned int count;      // entries this candidate asked for (each candidate has its own quota)
    unsigned int overflow;   // != 0: some of its lags did not fit -> the candidate stays FFS_FLAG_AMBIGUOUS
};

FFS_DEV unsigned long long score_key(double x) {
    const long long b = __double_as_longlong(x);
    return b >= 0 ? ((unsigned long long)b | 0x8000000000000000ull) : ~(unsigned long long)b;
}
FFS_DEV double key_score(unsigned long long k) {
    const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// Tie margin actually applied around a maximum vmax: the candidate's data-independent margin, or
// 24 ulp of the maximum itself if that is larger.  The fp32 pipeline's error grows with the DC
// content of the signals (measured up to ~6.4 eps * |c| at activity densities of 2 % / 98 %, where
// |c| reaches 7e5), so a margin proportional to sqrt(S*R) alone would be too tight there.
FFS_DEV float eff_margin(float cand_margin, float vmax) {
    return fmaxf(cand_margin, 24.0f * 5.9604645e-08f * fabsf(vmax));
}

FFS_DEV bool better(float v1, int d1, float v2, int d2) { return v1 > v2 || (v1 == v2 && d1 > d2); }

// Row walks of the mid / last passes: ONE wave-uniform 64-bit row pointer that advances by a wave-uniform stride plus
// one 32-bit byte offset per lane, i.e. `global_load_dwordx2 v, v_off, s[ptr:ptr+1]` and two scalar adds per row.
// Left to itself the compiler re-associates the chain into p + q*stride and evaluates it per lane (two v_mad_u64_u32,
// two moves and a 64-bit vector add per load: 55 of the ~410 VALU instructions of every mid-pass item, plus 30
// transient address VGPRs).  gstep() makes the advanced pointer opaque, so the chain stays s_add_u32/s_addc_u32; the
// pointers carry the global address space explicitly because an opaque generic pointer would turn the access into a
// flat_load (which also counts against lgkmcnt, i.e. against the LDS-only barriers).
typedef const __attribute__((address_space(1))) char* gcptr;
typedef __attribute__((address_space(1))) char* gptr;
FFS_DEV cf gload(gcptr p, unsigned off) { return *(const __attribute__((address_space(1))) cf*)(p + off); }
FFS_DEV void gstore(gptr p, unsigned off, cf v) { *(__attribute__((address_space(1))) cf*)(p + off) = v; }
// Streaming variants (`nt`): the intermediates are written once and read once, a whole launch (13 GB) later, so they
// need not displace anything in L2 / the Infinity Cache.  FFS_NT is a compile-time mask (A/B builds: -DFFS_NT=0):
// 1 = first-pass stores, 2 = mid-pass loads, 4 = mid-pass stores, 8 = last-pass loads.  All four on, measured against
// none (profiles/nt_ab.sh, r03_ab_experiments.json::run3l): first pass 4.83 -> 4.78 us/pair (windowless 11.2 -> 10.5),
// mid 7.63 -> 7.53, pruned last pass 1.21 -> 1.16; seven-ratio solves/s +1.4 % (+2 % on the windowless and
// reference-length plans, +2.5-3 % single-ratio).
#ifndef FFS_NT
#define FFS_NT 15
#endif
template <int BIT>
FFS_DEV cf gload_s(gcptr p, unsigned off) {
    if constexpr ((FFS_NT & BIT) != 0)
        return __builtin_nontemporal_load((const __attribute__((address_space(1))) cf*)(p + off));
    else
        return gload(p, off);
}
template <int BIT>
FFS_DEV void gstore_s(gptr p, unsigned off, cf v) {
    if constexpr ((FFS_NT & BIT) != 0)
        __builtin_nontemporal_store(v, (__attribute__((address_space(1))) cf*)(p + off));
    else
        gstore(p, off, v);
}
// first-pass stores: generic byte pointer + 32-bit offset (-DFFS_NT=0 keeps the plain stores it was tuned with)
#if FFS_NT & 1
#define FFS_PA_STORE(base, off, val) gstore_s<1>((gptr)(base), (off), (val))
#else
#define FFS_PA_STORE(base, off, val) (*reinterpret_cast<cf*>((base) + (off)) = (val))
#endif
FFS_DEV void gstep(gcptr& p, size_t stride) {
    p += stride;
    asm volatile("" : "+s"(p));
}
// The sixteen step twiddles W_N^(k1*LT*q) of a row are wave-uniform when a block owns one row: fetched once into
// scalar registers (s_load + SGPR operands of the packed multiplies) instead of inside the store loop, where every
// store to the work buffer would force a reload (the stores go through opaque pointers: no restrict information).
struct RowStepTw {
    float x[16], y[16];
    FFS_DEV void load(const cf* __restrict__ ts, int k1) {
#pragma unroll
        for (int q = 1; q < 16; ++q) {  // k1 wave-uniform: scalar loads; the "s" operand of cmul_k keeps them in SGPRs
            const cf t = ts[k1 * 16 + q];
            x[q] = t.x;
            y[q] = t.y;
        }
    }
    FFS_DEV cf times(cf wb, int q) const { return q == 0 ? wb : cmul_k(wb, x[q], y[q]); }
};
FFS_DEV void gstep(gptr& p, size_t stride) {
    p += stride;
    asm volatile("" : "+s"(p));
}

// Branch-free sample fetch: the load is always issued (index clamped to element 0, so the sixteen
// loads of a thread are in flight together) and the zero padding is applied by a select.
template <int DT>
FFS_DEV float load_mapped(const void* p, int len, int n, float v0, float v1, int lead = 0) {
    const bool in = n < len && n >= lead;
    const int idx = in ? n : lead;
    float val;
    if (DT == 0) {
        val = (reinterpret_cast<const unsigned char*>(p)[idx] != 0) ? v1 : v0;
    } else if (DT == 3) {
        val = (float)(2.0 * reinterpret_cast<const double*>(p)[idx] - 1.0);
    } else {
        val = 2.0f * reinterpret_cast<const float*>(p)[idx] - 1.0f;
    }
    return in ? val : 0.0f;
}

// Two-level bytes -> mapped sample values, zero beyond the end of the vector.  Element q of the thread
// is sample n_base + LT*N2*q.  The zero-padding test is skipped for the leading QF values of q when the
// whole block (columns < col_end, all LT row phases) is inside the vector there -- a block-uniform
// condition, so the three variants are selected by scalar branches.
// MASK: b[q] is an all-ones / all-zeros word (from a signed bit-field extract) and the value is picked with
// one v_bfi_b32; otherwise b[q] is any byte value and the pick is a compare + select.
template <bool MASK>
FFS_DEV float pick_level(unsigned b, float v0, float v1) {
    if (MASK) return __uint_as_float((b & __float_as_uint(v1)) | (~b & __float_as_uint(v0)));
    return b ? v1 : v0;
}
template <int LT, int QF, bool MASK>
FFS_DEV void map_bytes_from(const unsigned (&b)[16], float v0, float v1, int len, int n_base, int N2, float (&out)[16]) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float x = pick_level<MASK>(b[q], v0, v1);
        out[q] = (q < QF || n_base + LT * N2 * q < len) ? x : 0.0f;
    }
}
template <int LT, bool MASK = false>
FFS_DEV void map_bytes(const unsigned (&b)[16], float v0, float v1, int len, int n_base, int N2, int col_end,
                       int lead, float (&out)[16]) {
    if (lead > 0) {  // rare (one reference block per pair in block-segmented mode): test both ends
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int n = n_base + LT * N2 * q;
            out[q] = (n >= lead && n < len) ? pick_level<MASK>(b[q], v0, v1) : 0.0f;
        }
        return;
    }
    const int rows_full = (len >= col_end) ? (len - col_end) / N2 + 1 : 0;  // rows r with r*N2 + col_end - 1 < len
    const int q_full = rows_full / LT;                                      // q with every row u + LT*q inside
    if (q_full >= 12)
        map_bytes_from<LT, 12, MASK>(b, v0, v1, len, n_base, N2, out);
    else if (q_full >= 6)
        map_bytes_from<LT, 6, MASK>(b, v0, v1, len, n_base, N2, out);
    else
        map_bytes_from<LT, 0, MASK>(b, v0, v1, len, n_base, N2, out);
}

// value of `x` in the neighbour lane (lane ^ 1): one DPP move, quad_perm [1,0,3,2]
FFS_DEV float from_neighbour(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
}

// Tile layout of the intermediate arrays: element (x, k1) of a transform (x = n2 or m1, k1 the
// column-transform index) lives at ((x / CL)*N1 + k1)*CL + x % CL with CL = 2^log2CL >= C columns, so
// a row k1 is a sequence of CL*8-byte chunks (256 B for CL = 32) and a block's C-column tile is a
// C*8-byte slice of each chunk.  Returns the offset of (tile*C + c, k1 = 0); rows advance by CL.
template <int L, int C>
FFS_DEV size_t tile_base(int tile, int c, int log2CL) {
    const int x = tile * C + c;
    return (((size_t)(x >> log2CL) * L) << log2CL) + (size_t)(x & ((1 << log2CL) - 1));
}

// Prefetch block of pass A, bit-packed inputs.  Pass A is a write stream (26 MB per pair) with 0.7 MB of input reads
// sprinkled over it, and those few reads cost a quarter of its time: section experiments (FFS_PASS_A_DEBUG) give 6.15
// us/pair as is, 4.37 without the input loads and 4.55 when every block reads the same, L2-resident vectors -- the
// reads are slow, and slow the writes down, only when they go to HBM one line at a time in the middle of the write
// traffic.  So the lines of the transform a few grid rows further down are fetched in ONE burst per XCD: the last
// eight workgroups of a grid row (gridDim.x = nt + 8 = 0 mod 8, so workgroup x runs on XCD x % 8, which also runs
// the tiles (x % 8) * nt/8 ...) each touch the lines under their XCD's `ncols` columns, all L rows of both vectors;
// the consumers find them in their L2 a few microseconds later (6.1 -> 5.0 us/pair at twelve rows ahead).
FFS_DEV void prefetch_bit_inputs(const XformDesc* __restrict__ descs, int y, int n_desc, int xcd, int L, int N2, int ncols, int nthreads) {
    if (y >= n_desc) return;
    const XformDesc dn = descs[y];
    const int colA = xcd * ncols;
    unsigned acc = 0;
    for (int t = threadIdx.x; t < 2 * L; t += nthreads) {
        const int h = t / L, row = t % L;
        const auto* src = (const __attribute__((address_space(1))) unsigned*)(h ? dn.b : dn.a);
        const int off = h ? dn.off_b : dn.off_a, len = h ? dn.len_b : dn.len_a, lead = h ? dn.lead_b : dn.lead_a;
        if (len <= lead) continue;
        // bits [b0, b1] of this row under the XCD's columns, clipped to the vector (ncols = 512 bits: at most two lines)
        int b0 = off + row * N2 + colA, b1 = b0 + ncols - 1;
        b0 = b0 > off + lead ? b0 : off + lead;
        b1 = b1 < off + len - 1 ? b1 : off + len - 1;
        if (b0 > b1) continue;
        acc |= src[b0 >> 5] | src[b1 >> 5];
    }
    asm volatile("" ::"v"(acc));  // keeps the loads alive
}

// --------------------------------------------------------------------------------------------
// pass A.  grid = (N2/C column tiles [+ input prefetch blocks: N2/128 for byte inputs, 8 (one per XCD) for bit-packed
// ones], n_transforms); block = (L/16)*C threads; thread (c = tid % C, u = tid / C).
// Paired transforms (bit-packed inputs, power-of-two columns): a group's reference and its single last candidate are
// both real vectors with half slots, so ONE column transform of z = ref + i*last serves both -- per column
// ref^[k1] = (Z[k1] + conj Z[L-k1])/2 and last^[k1] = (Z[k1] - conj Z[L-k1])/(2i), the mirror rows fetched through the
// column tile in LDS.  PM = 0: one grid row per transform, no pairing.  PM = 1: grid rows = transform GROUPS, every
// block a paired one (groups of two transforms).  PM = 2: xf_per_pair - 1 grid rows per group -- row 0 the paired
// transform, row j the plain transform j -- in one launch.
template <int L, int C, int DT, int PM = 0>
__global__ __launch_bounds__((L / 16) * C) void k_pass_a(const XformDesc* __restrict__ descs, cf* __restrict__ work,
                                                         int N2, long long N, const cf* __restrict__ tw,
                                                         const cf* __restrict__ tb, const cf* __restrict__ ts,
                                                         const cf* __restrict__ tw3, int log2CL, int xf_per_pair,
                                                         int slots_per_pair, int nt, int pf_ahead,
                                                         unsigned* __restrict__ pf_sink, int half_flags, int row_sel) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    typedef ColShape<L> CS;
    constexpr int LT = L / 16;
    // grid row -> (transform group, transform within the group, paired or not).  row_sel (PM == 0 only; 0 = every
    // transform): first | count << 16 -- the launch covers transforms [first, first + count) of every group, e.g. only
    // the reference slots, or only the candidate slots, when the two come in different element types
    constexpr bool CAN_PAIR = PM != 0;
    const int sel_first = PM == 0 ? (row_sel & 0xffff) : 0, sel_count = PM == 0 ? (row_sel >> 16) : 0;
    const int rows_per_group = PM == 2 ? xf_per_pair - 1 : (PM == 1 ? 1 : (sel_count ? sel_count : xf_per_pair));
    auto desc_of = [&](int y, bool* is_paired) {
        const int g = y / rows_per_group, j = y % rows_per_group;
        *is_paired = CAN_PAIR && j == 0;
        return g * xf_per_pair + sel_first + j;
    };
    if (DT == 0 && blockIdx.x >= (unsigned)nt) {
        // Prefetch block.  The input bytes are the only HBM reads of this kernel, and a read that misses
        // while every CU streams writes takes microseconds (measured: 9.0 -> 6.9 us/pair with cache-resident
        // inputs).  gridDim.x - nt extra blocks per grid row touch the 128-byte input lines of the
        // transform pf_ahead rows further down, one line group (128 columns, all L rows, both halves)
        // each; dispatch order puts them a few microseconds ahead of the blocks that need the lines, on
        // the same XCD (= same L2): group index mirrors the tile mapping below.
        const int y = blockIdx.y + pf_ahead;
        if (y >= (int)gridDim.y) return;
        bool pp0;
        const XformDesc dn = descs[desc_of(y, &pp0)];
        const int i = blockIdx.x - nt, ng = gridDim.x - nt;
        const int grp = (i % 8) * (ng / 8) + i / 8;
        unsigned acc = 0;
        for (int row = threadIdx.x; row < L; row += LT * C) {
            const int n0 = row * N2 + grp * 128;
            unsigned wa = 0, wb = 0;
            if (n0 + 4 <= dn.len_a && n0 >= dn.lead_a) __builtin_memcpy(&wa, reinterpret_cast<const unsigned char*>(dn.a) + n0, 4);
            if (n0 + 4 <= dn.len_b && n0 >= dn.lead_b) __builtin_memcpy(&wb, reinterpret_cast<const unsigned char*>(dn.b) + n0, 4);
            acc |= wa | wb;
        }
        if (acc == 0xdeadbeefu && pf_sink) *pf_sink = acc;  // keeps the loads alive (pf_sink is scratch)
        return;
    }
    if (DT == 2 && blockIdx.x >= (unsigned)nt) {
        const int y = (int)blockIdx.y + pf_ahead;
        if (y >= (int)gridDim.y) return;
        bool pp;
        const int di = desc_of(y, &pp);
        const int n_desc = ((int)gridDim.y / rows_per_group) * xf_per_pair;
        prefetch_bit_inputs(descs, di, n_desc, (int)blockIdx.x - nt, L, N2, (nt / 8) * C, LT * C);
        if (pp) prefetch_bit_inputs(descs, di + xf_per_pair - 1, n_desc, (int)blockIdx.x - nt, L, N2, (nt / 8) * C, LT * C);
        return;
    }
    bool paired;
    const int desc_index = desc_of((int)blockIdx.y, &paired);
    const int group = (int)blockIdx.y / rows_per_group;
    const int c = threadIdx.x % C;
    const int u = threadIdx.x / C;
    // XCD-aware tile mapping: workgroup b runs on XCD b % 8 (each XCD has its own L2), and the input
    // bytes of 128/C neighbouring tiles share one 128-byte line, so give every XCD a contiguous run of
    // tiles instead of every eighth one -- otherwise each input line is fetched by up to 8 L2s.
    const int tile = (nt % 8 == 0) ? (int)(blockIdx.x % 8) * (nt / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int n2 = tile * C + c;
    XformDesc d = descs[FFS_LABF(half_flags, DBG_PA_HOT_INPUT) ? 0 : desc_index];
    if constexpr (CAN_PAIR) {
        static_assert(DT == 2 && !CS::R3, "paired first pass: bit-packed inputs, power-of-two columns");
        if (paired) {  // block-uniform
            const XformDesc dl = descs[desc_index + xf_per_pair - 1];
            d.b = dl.a, d.len_b = dl.len_a, d.b0 = dl.a0, d.b1 = dl.a1, d.lead_b = dl.lead_a, d.off_b = dl.off_a;
        }
    }
    // every table value this thread needs is requested up front, together with the inputs
    // (tiles of 64+ columns: u is the wave's row phase, the stage twiddles are wave-uniform -> scalar registers)
    constexpr bool TWS = (C % 64 == 0) && !CS::R3 && (LT > 1) && (LT < 16);
    TwRegs<CS::LI, TWS> twr;
    twr.load(tw, TWS ? __builtin_amdgcn_readfirstlane(u) : (CS::R3 ? u / 3 : u));
    cf* s_tw3 = lds + L * C;  // block copy of W_L^k for the radix-3 combine (behind the column tile)
    if constexpr (CS::R3) {
        for (int i = threadIdx.x; i < L; i += LT * C) s_tw3[i] = tw3[i];
        __syncthreads();
    }
    // W_N^(n2*k1), k1 = ob + OSTEP*q (ob = out_base(u)):  tb[u][n2] * g^q with g = W_N^(n2*OSTEP) (the
    // host lays the tables out per column shape).  Only g, g^2, g^4, g^8 are
    // fetched (ts rows 1, 2, 4, 8); the other powers are built with at most three multiplications
    // each, which trades eleven table loads per thread for eleven packed complex multiplies.
    cf wq[16];
    if constexpr (!CS::R3) {
        if FFS_LABF(half_flags, DBG_PA_NO_TW) {
            wq[0] = wq[1] = wq[2] = wq[4] = wq[8] = mk(1.0f, 0.0f);
        } else {
            wq[0] = tb[u * N2 + n2];
            wq[1] = ts[1 * N2 + n2];
            wq[2] = ts[2 * N2 + n2];
            wq[4] = ts[4 * N2 + n2];
            wq[8] = ts[8 * N2 + n2];
        }
    }
    cf v[16];
    if constexpr (DT == 0 && (C == 16 || C == 32 || C == 64)) {
        // Byte inputs: the 64 sixteen-byte pieces a wave needs per candidate (64/C rows x C/16 pieces
        // for each of the 16 q) are fetched by ONE dwordx4 load per lane and redistributed through
        // LDS, instead of sixteen scalar byte loads per lane.
        constexpr int PPR = C / 16;   // pieces per row chunk
        constexpr int UPW = 64 / C;   // rows (u values) per wave
        const int lane = threadIdx.x & 63;
        const int wave = threadIdx.x >> 6;
        unsigned char* stage = smem + wave * 2048;
        const int pr = lane & 3;  // piece within this q: row pr / PPR, sixteen-byte part pr % PPR
        const int lrow = (wave * UPW + pr / PPR) + LT * (lane >> 2);
        const int n0 = lrow * N2 + tile * C + (pr % PPR) * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(h ? d.b : d.a);
            const int len = h ? d.len_b : d.len_a;
            const int lead = h ? d.lead_b : d.lead_a;
            uint4 w = make_uint4(0u, 0u, 0u, 0u);
            if (n0 >= lead && n0 + 16 <= len) {
                __builtin_memcpy(&w, src + n0, 16);
            } else if (n0 < len && n0 + 16 > lead) {  // a piece that straddles either end of the data
                unsigned char tmp[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) tmp[k] = (n0 + k < len && n0 + k >= lead) ? src[n0 + k] : (unsigned char)0;
                __builtin_memcpy(&w, tmp, 16);
            }
            *reinterpret_cast<uint4*>(stage + h * 1024 + lane * 16) = w;
        }
        __syncthreads();
        const int piece = (u % UPW) * PPR + c / 16;
        // all 32 byte reads are issued back to back; the empty asm keeps the compiler from sinking each
        // one into its own "n < len" branch (one exposed LDS round trip per element otherwise)
        unsigned ba[16], bb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            ba[q] = stage[(q * 4 + piece) * 16 + (c & 15)];
            bb[q] = stage[1024 + (q * 4 + piece) * 16 + (c & 15)];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(ba[q]), "+v"(bb[q]));
        float xa[16], xb[16];
        map_bytes<LT>(ba, d.a0, d.a1, d.len_a, u * N2 + n2, N2, tile * C + C, d.lead_a, xa);
        map_bytes<LT>(bb, d.b0, d.b1, d.len_b, u * N2 + n2, N2, tile * C + C, d.lead_b, xb);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = mk(xa[q], xb[q]);
    } else if constexpr (DT == 2) {
      {
        // Bit-packed inputs: the tile's L rows x C columns are L windows of C bits per vector, at arbitrary
        // bit offsets.  The block first builds the ALIGNED windows in LDS -- one thread per 32 window bits:
        // two dword loads and a funnel shift (v_alignbit) -- so that bit c of a row's window is column c;
        // afterwards a thread's 2 x 16 samples cost one LDS read (compile-time offset), one bit-field
        // extract and one select each.  An eighth of the byte path's input lines, no prefetch blocks.
        constexpr int CW = (C + 31) / 32;  // dwords per window
        unsigned* stage = reinterpret_cast<unsigned*>(smem);  // [2][L][CW]; the FFT's first LDS write is behind a barrier
        const int col0 = tile * C;
        for (int i = threadIdx.x; i < 2 * L * CW; i += LT * C) {
            const int h = i / (L * CW), row = (i / CW) % L, j = i % CW;
            // (descriptor pointers are generic: say "global" so the loads are global_load, not flat_load)
            const auto* src = (const __attribute__((address_space(1))) unsigned*)(h ? d.b : d.a);
            const int off = h ? d.off_b : d.off_a, len = h ? d.len_b : d.len_a, lead = h ? d.lead_b : d.lead_a;
            const int bit0 = off + row * N2 + col0 + 32 * j;  // first bit of this dword of the window
            const int w = bit0 >> 5;                          // arithmetic shift: floor
            // only dwords that hold a valid sample (bits off+lead .. off+len-1) are touched
            const int w_lo = (off + lead) >> 5, w_hi = (off + len - 1) >> 5;
            unsigned d0 = 0, d1 = 0;
            if (len > lead && !FFS_LABF(half_flags, DBG_PA_NO_INPUT)) {
                if (w >= w_lo && w <= w_hi) d0 = src[w];
                if (w + 1 >= w_lo && w + 1 <= w_hi) d1 = src[w + 1];
            }
            stage[i] = __builtin_amdgcn_alignbit(d1, d0, (unsigned)bit0 & 31u);
        }
        __syncthreads();
        const unsigned* wa = stage + u * CW + (c >> 5);
        const unsigned* wb = wa + L * CW;
        const unsigned sh = c & 31;
        unsigned ba[16], bb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            ba[q] = (unsigned)__builtin_amdgcn_sbfe((int)wa[LT * q * CW], sh, 1u);  // 0 or ~0
            bb[q] = (unsigned)__builtin_amdgcn_sbfe((int)wb[LT * q * CW], sh, 1u);
        }
        float xa[16], xb[16];
        map_bytes<LT, true>(ba, d.a0, d.a1, d.len_a, u * N2 + n2, N2, tile * C + C, d.lead_a, xa);
        map_bytes<LT, true>(bb, d.b0, d.b1, d.len_b, u * N2 + n2, N2, tile * C + C, d.lead_b, xb);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = mk(xa[q], xb[q]);
      }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int n = (u + LT * q) * N2 + n2;
            v[q].x = load_mapped<DT>(d.a, d.len_a, n, d.a0, d.a1, d.lead_a);
            v[q].y = load_mapped<DT>(d.b, d.len_b, n, d.b0, d.b1, d.lead_b);  // absent candidate: len_b == 0
        }
    }
    // transform blockIdx.y of the launch is transform (y % xf_per_pair) of pair (y / xf_per_pair); a pair
    // owns slots_per_pair consecutive length-N buffers
    const int xi = desc_index - group * xf_per_pair;
    cf* out = work + ((size_t)group * slots_per_pair + xi) * N;
    if FFS_LABF(half_flags, DBG_PA_HOT_STORE) out = work;
    // HALF_REF: the reference transform (slot 0) is of a real signal, so its rows k1 > L/2 mirror the
    // rows L - k1 (X[N-k] = conj X[k]); k_mid rebuilds them and they are not stored at all.
    // HALF_LAST: with an odd candidate count the last packed transform carries ONE real candidate; its
    // product with the reference spectrum stays Hermitian, so neither its rows k1 > L/2 nor their
    // results are ever needed (the last pass rebuilds them by conjugation, see k_pass_c*).
    const int k1_end = (((half_flags & HALF_REF) && xi == 0) || ((half_flags & HALF_LAST) && xi == xf_per_pair - 1))
                           ? L / 2 + 1 : L;
    if constexpr (CS::R3) {
        // Radix-3 columns: the last step (the combine) reads its inputs from LDS anyway, so the outputs
        // are re-dealt for the store: thread (cp, rg) produces the rows k1 = kg + 2*LTI*j + LI*r (j < 8;
        // rg = r*2*LTI + kg) of the column pair (2cp, 2cp+1) -- eight 16-byte stores with no lane
        // exchange, 16-byte LDS reads, and r is wave-uniform.
        constexpr int LI = CS::LI, KG = 2 * CS::LTI;
        static_assert(C % 2 == 0 && (C / 2) * KG % 64 == 0, "r must be wave-uniform");
        col_fft3_front<L, C>(v, lds, u, c, twr, s_tw3);
        const int cp = threadIdx.x % (C / 2), rg = threadIdx.x / (C / 2);
        const int r = __builtin_amdgcn_readfirstlane(rg / KG), kg = rg % KG;
        const float alpha = (r == 0) ? 1.0f : -0.5f;
        const float beta = (r == 0) ? 0.0f : (r == 1 ? FFS_SQRT3_HALF : -FFS_SQRT3_HALF);
        // W_N^(n2*k1) for both columns: h_j = tb[rg][n2] * g^j, g = W_N^(n2*KG) (ts rows 1, 2, 4)
        const int n2p = tile * C + 2 * cp;
        const float4 t0 = *reinterpret_cast<const float4*>(&tb[(size_t)rg * N2 + n2p]);
        const float4 t1 = *reinterpret_cast<const float4*>(&ts[(size_t)1 * N2 + n2p]);
        const float4 t2 = *reinterpret_cast<const float4*>(&ts[(size_t)2 * N2 + n2p]);
        const float4 t4 = *reinterpret_cast<const float4*>(&ts[(size_t)4 * N2 + n2p]);
        cf h[2][8];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const cf g1 = e ? mk(t1.z, t1.w) : mk(t1.x, t1.y), g2 = e ? mk(t2.z, t2.w) : mk(t2.x, t2.y);
            const cf g4 = e ? mk(t4.z, t4.w) : mk(t4.x, t4.y);
            h[e][0] = e ? mk(t0.z, t0.w) : mk(t0.x, t0.y);
            h[e][1] = cmul(h[e][0], g1);
            h[e][2] = cmul(h[e][0], g2);
            h[e][3] = cmul(h[e][1], g2);
#pragma unroll
            for (int j = 0; j < 4; ++j) h[e][4 + j] = cmul(h[e][j], g4);
        }
        const float4* l4 = reinterpret_cast<const float4*>(lds);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ((kg + KG * j) * C + 2 * cp) / 2;  // float4 index of F'[0][k'][2cp]
            const float4 a = l4[k], b = l4[k + LI * C / 2], cc = l4[k + LI * C];
            const cf x0 = cmul(radix3_out(mk(a.x, a.y), mk(b.x, b.y), mk(cc.x, cc.y), alpha, beta), h[0][j]);
            const cf x1 = cmul(radix3_out(mk(a.z, a.w), mk(b.z, b.w), mk(cc.z, cc.w), alpha, beta), h[1][j]);
            const int k1 = kg + KG * j + LI * r;
            if (k1 < k1_end)
                *reinterpret_cast<float4*>(&out[tile_base<L, C>(tile, 2 * cp, log2CL) + ((size_t)k1 << log2CL)]) =
                    make_float4(x0.x, x0.y, x1.x, x1.y);
        }
        return;
    }
    col_fft<L, C, false, TWS>(v, lds, u, c, twr, s_tw3);
    // v[q] = Y[k1 = u + LT*q][n2];  twiddle W_N^(n2*k1) = h_q = wq[0] * g^q, built as h_(q-b) * g^b
    const int ob = CS::out_base(u);
    wq[3] = cmul(wq[0], wq[1]);   // h_1
    wq[5] = cmul(wq[0], wq[2]);   // h_2
    wq[6] = cmul(wq[3], wq[2]);   // h_3
    wq[7] = cmul(wq[0], wq[4]);   // h_4
    wq[9] = cmul(wq[3], wq[4]);   // h_5
    wq[10] = cmul(wq[5], wq[4]);  // h_6
    wq[11] = cmul(wq[6], wq[4]);  // h_7
    if (CAN_PAIR && paired) {
        // v[q] = Z[k1 = u + LT*q] of z = ref + i*last (no twiddle yet: the mirror row has its own).  The raw column goes
        // through the tile in LDS, every thread picks up Z[L - k1] for its rows k1 <= L/2 and writes the two separated
        // spectra, times W_N^(n2*k1), to the reference slot and the last slot (half slots: rows 0..L/2).
        const cf h[9] = {wq[0], wq[3], wq[5], wq[6], wq[7], wq[9], wq[10], wq[11], cmul(wq[0], wq[8])};
        __syncthreads();  // the transform's last exchange has been read
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[(u + LT * q) * C + c] = v[q];
        __syncthreads();
        cf* out_l = out + (size_t)(xf_per_pair - 1) * N;
        const size_t o0 = tile_base<L, C>(tile, c, log2CL);
#pragma unroll
        for (int q = 0; q <= 8; ++q) {
            const int k1 = u + LT * q;
            if (k1 > L / 2) continue;
            const cf m = lds[((L - k1) & (L - 1)) * C + c];
            const cf a = mk(0.5f * (v[q].x + m.x), 0.5f * (v[q].y - m.y));
            const cf b = mk(0.5f * (v[q].y + m.y), 0.5f * (m.x - v[q].x));
            const unsigned ob8 = (unsigned)((o0 + ((size_t)k1 << log2CL)) * sizeof(cf));
            FFS_PA_STORE(reinterpret_cast<char*>(out), ob8, cmul(a, h[q]));
            FFS_PA_STORE(reinterpret_cast<char*>(out_l), ob8, cmul(b, h[q]));
        }
        return;
    }
    {
        const cf h[8] = {wq[0], wq[3], wq[5], wq[6], wq[7], wq[9], wq[10], wq[11]};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v[q] = cmul(v[q], h[q]);
            v[q + 8] = cmul(v[q + 8], cmul(h[q], wq[8]));
        }
    }
    if (C >= 64 && (half_flags & STORE_8B)) {
        // 64-column tiles: an 8-byte store per lane already writes one whole 512-byte row chunk per wave
        // instruction, so the lane-pair exchange below (64 VALU instructions of ~330 per wave; the kernel is
        // ~45 % VALU-busy) buys nothing here: measured 6.54 -> 6.28 us/pair
        // a slot is at most 2^24 elements: 32-bit element offsets from the (block-uniform) slot pointer,
        // one add per store; the row guard only exists for the two half slots (block-uniform choice)
        // (byte offsets in 32-bit arithmetic -- 2^27 at most -- so that the stores can use the scalar base +
        // 32-bit lane offset addressing mode instead of 64-bit address pairs)
        const unsigned o0 = ((unsigned)tile_base<L, C>(FFS_LABF(half_flags, DBG_PA_HOT_STORE) ? 0 : tile, c, log2CL) +
                             ((unsigned)ob << log2CL)) * (unsigned)sizeof(cf);
        const unsigned ostep = ((unsigned)CS::OSTEP << log2CL) * (unsigned)sizeof(cf);
        char* outb = reinterpret_cast<char*>(out);
        if FFS_LABF(half_flags, DBG_PA_NO_STORE) {
            // keep the values alive without memory traffic
#pragma unroll
            for (int q = 0; q < 16; ++q) asm volatile("" ::"v"(v[q]));
        } else if (k1_end == L) {
#pragma unroll
            for (int q = 0; q < 16; ++q) FFS_PA_STORE(outb, o0 + ostep * q, v[q]);
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (ob + CS::OSTEP * q < k1_end) FFS_PA_STORE(outb, o0 + ostep * q, v[q]);
        }
    } else if constexpr (C >= 2) {
        // Pair up neighbouring columns so every lane issues 8 x 16-byte stores instead of 16 x 8-byte:
        // the even-c lane writes rows q = 0,2,.. of columns (c, c+1), the odd-c lane rows q = 1,3,..
        // (a wave store then covers 8 full 128-byte rows).
        // Even lanes end up with (own v[2j], neighbour's v[2j]), odd lanes with (neighbour's v[2j+1], own
        // v[2j+1]); the exchange is a DPP move feeding a select (no LDS permute).
        const bool odd = c & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const cf e = v[2 * j], o = v[2 * j + 1];
            // (every lane executes all four moves: a DPP read of a lane that is masked off returns nothing)
            const float nox = from_neighbour(o.x), noy = from_neighbour(o.y);
            const float nex = from_neighbour(e.x), ney = from_neighbour(e.y);
            float4 pk;
            pk.x = odd ? nox : e.x;
            pk.y = odd ? noy : e.y;
            pk.z = odd ? o.x : nex;
            pk.w = odd ? o.y : ney;
            const int k1 = ob + CS::OSTEP * (2 * j + (odd ? 1 : 0));
            if (k1 < k1_end)
                *reinterpret_cast<float4*>(&out[tile_base<L, C>(tile, c & ~1, log2CL) + ((size_t)k1 << log2CL)]) = pk;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (ob + CS::OSTEP * q < k1_end) out[tile_base<L, C>(tile, c, log2CL) + ((size_t)(ob + CS::OSTEP * q) << log2CL)] = v[q];
    }
}

template <int L, int... Q>
FFS_DEV void mirror_store(const cf (&v)[16], cf* lds, const RowAddr<L>& addr, std::integer_sequence<int, Q...>) {
    ((lds[addr.template gather<Q>()] = v[Q]), ...);
}
template <int L, int... Q>
FFS_DEV void mirror_load(cf (&v)[16], const cf* lds, const RowAddr<L>& addr, std::integer_sequence<int, Q...>) {
    ((v[Q] = lds[addr.template gather_mirror<Q>()]), ...);
}

// --------------------------------------------------------------------------------------------
// mid pass.  grid = (N1/ROWS, n_pairs); block = 256 threads; ROWS = 256/(L/16) rows per block.
// Slot 0 of each pair is the reference transform; slots 1..n_slots-1 are transformed in place.
// SEP: L/16 >= C, so element u + LT*q of a row sits at  off0 + q*(LT*N1)  (one 32-bit lane offset
// plus a wave-uniform stride) instead of needing sixteen independent 64-bit addresses.
template <int L, bool SEP, bool PF = false>
__global__ __launch_bounds__(256, 3) void k_mid(cf* __restrict__ work, int N1, int log2C, long long N, int n_slots,
                                                float inv_n, const cf* __restrict__ tw, const cf* __restrict__ tb,
                                                const cf* __restrict__ ts, int half_flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int LT = L / 16;
    constexpr int ROWS = 256 / LT;
    const int ref_half = half_flags & HALF_REF;
    constexpr int ROW_STRIDE = RowAddr<L>::ROW_ELEMS;
    const int row = threadIdx.x / LT;
    const int u = threadIdx.x % LT;
    const int k1 = blockIdx.x * ROWS + row;
    RowAddr<L> addr(row * ROW_STRIDE, u);
    const int C = 1 << log2C;
    cf* base = work + (size_t)blockIdx.y * n_slots * N;

    // element x = u + LT*q of row k1 lives at ((x/C)*N1 + k1)*C + x%C
    const unsigned off0 = (unsigned)(((u >> log2C) * N1 + k1) * C + (u & (C - 1)));
    const size_t qstride = (size_t)LT * N1;
    auto at = [&](cf* buf, int q) -> cf& {
        const int x = u + LT * q;
        return buf[((x >> log2C) * N1 + k1) * C + (x & (C - 1))];
    };
    // SEP: a scalar row pointer that advances by the (opaque) stride + one 32-bit lane offset per access -- no
    // per-element vector address arithmetic and no hoisted table of sixteen 64-bit offsets (32 VGPRs)
    auto load_row = [&](cf(&x)[16], const cf* buf, unsigned off_elems) {
        if constexpr (SEP) {
            gcptr p = (gcptr)buf;
            size_t stride = qstride * sizeof(cf);
            unsigned off = off_elems * (unsigned)sizeof(cf);
            asm volatile("" : "+s"(stride), "+v"(off));
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                x[q] = gload_s<2>(p, off);
                gstep(p, stride);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = at(const_cast<cf*>(buf), q);
        }
    };

    // HALF_LAST (one row per block): rows above N1/2 of the single-candidate slot are never needed
    const int s_end = (L == 4096 && SEP && (half_flags & HALF_LAST) && k1 > N1 / 2) ? n_slots - 1 : n_slots;
    if (s_end <= 1) return;  // block-uniform: nothing to do for this row (single-candidate solves)
    TwRegs<L> twr;
    twr.load(tw, u);
    cf rr[16];
    // ref_half (one row per block only): pass A stored the reference rows 0..N1/2; for k1 > N1/2,
    // conj(R[k1][k2]) = R[N1-k1][N2-1-k2] (Hermitian spectrum of a real signal in the four-step index
    // k = k1 + N1*k2), i.e. the transformed row N1-k1 read backwards -- one more exchange through LDS.
    const bool mirrored = (L == 4096) && SEP && ref_half && k1 > N1 / 2;
    if (mirrored) {
        const unsigned offr = (unsigned)(((u >> log2C) * N1 + (N1 - k1)) * C + (u & (C - 1)));
        load_row(rr, base, offr);
    } else {
        load_row(rr, base, off0);
    }
    // PF: the first candidate row is requested before the reference row is transformed, every further one before the
    // two transforms of its predecessor (staging buffer xl, LDS-only barriers: see k_mid_seg_one)
    cf xl[PF ? 16 : 1];
    if constexpr (PF) {
        load_row(xl, base + (size_t)1 * N, off0);
        __builtin_amdgcn_sched_barrier(0);
    }
    fft_regs<L, RowAddr<L>, PF>(rr, lds, u, addr, twr);
    if constexpr (L == 4096) {
        if (mirrored) {
            block_sync<PF>();
            mirror_store(rr, lds, addr, std::make_integer_sequence<int, 16>{});
            block_sync<PF>();
            mirror_load(rr, lds, addr, std::make_integer_sequence<int, 16>{});
        }
    }
    const float sgn = mirrored ? inv_n : -inv_n;
#pragma unroll
    for (int q = 0; q < 16; ++q) rr[q] = mk(rr[q].x * inv_n, rr[q].y * sgn);  // conj(R)/N

    const cf wb = tb[(size_t)k1 * LT + u];  // W_N^(k1*u)
    constexpr bool ONE_ROW = (L == 4096) && SEP;  // one row per block: k1 is wave-uniform
    RowStepTw stw;
    if constexpr (ONE_ROW) stw.load(ts, __builtin_amdgcn_readfirstlane(k1));
    for (int s = 1; s < s_end; ++s) {
        cf* buf = base + (size_t)s * N;
        cf v[16];
        if constexpr (PF) {
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = xl[q];
            if (s + 1 < s_end) load_row(xl, base + (size_t)(s + 1) * N, off0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            load_row(v, buf, off0);
        }
        fft_regs<L, RowAddr<L>, PF>(v, lds, u, addr, twr);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = cmul(v[q], rr[q]);
        fft_regs<L, RowAddr<L>, PF>(v, lds, u, addr, twr);
        if constexpr (SEP) {
            gptr p = (gptr)buf;
            size_t stride = qstride * sizeof(cf);
            unsigned off = off0 * (unsigned)sizeof(cf);
            cf wbl = wb;  // opaque: the sixteen products wb*ts[q] are not hoisted out of the slot loop
            asm volatile("" : "+s"(stride), "+v"(off), "+v"(wbl));
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf w = ONE_ROW ? stw.times(wbl, q) : ((q == 0) ? wbl : cmul(wbl, ts[k1 * 16 + q]));  // W_N^(k1*(u + LT*q))
                gstore_s<4>(p, off, cmul(v[q], w));
                gstep(p, stride);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const cf w = (q == 0) ? wb : cmul(wb, ts[k1 * 16 + q]);  // W_N^(k1*(u + LT*q))
                at(buf, q) = cmul(v[q], w);
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// mid pass, block-segmented mode (N2 = 4096: one row per block).  A pair is n_blocks groups of n_slots
// length-N buffers (N = the block transform length M): group k holds the pass-A output of block k of
// the reference (slot 0, rows 0..N1/2 when ref_half) and of the candidate transforms (slots 1..).
// For every candidate slot the spectrum products of the blocks are ADDED, then one row transform goes
// back:  acc = sum_k FFT(S_k row) * conj(FFT(R_k row)) / N.  The result replaces group 0's slot (its
// own input row was consumed two blocks earlier), so mid writes and the last pass reads 1/n_blocks of
// what the unsegmented pipeline moves.
// PAIR_ROWS: rows k1 and N1-k1 read the same stored reference rows (one of them mirrored).  Workgroup b
// runs on XCD b % 8, so the two rows of a pair get block indices 8 apart: same XCD (same L2), dispatched
// back to back -- the second read of every reference row can then be an L2 hit instead of HBM traffic.
// Pair j = (j, N1-j) for 0 < j < N1/2, pair 0 = the two self-mirrored rows (0, N1/2).
//
// k_mid_seg_pipe (solves with more than four packed candidate slots): two candidate slots per sweep over the
// blocks -- each reference row is loaded and transformed for two slots -- with two accumulator rows;
// software-pipelined: the sixteen loads of the NEXT row (reference row of the next block, or the next slot's row) are issued
// before the transform of the current one, so the HBM latency of every row hides behind a row transform
// instead of stalling the block (two blocks per CU = two waves per SIMD leave little else to switch to).
// Rows alternate between two register buffers; the item order of a sweep is R_0 A_0 B_0 R_1 A_1 B_1 ...
template <int L>
__global__ __launch_bounds__(256, 2) void k_mid_seg_pipe(cf* __restrict__ work, int N1, int log2C, long long N, int n_slots,
                                                         int n_blocks, float inv_n, const cf* __restrict__ tw,
                                                         const cf* __restrict__ tb, const cf* __restrict__ ts,
                                                         int half_flags) {
    static_assert(L == 4096, "one row per 256-thread block");
    const int ref_half = half_flags & HALF_REF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int LT = L / 16;
    cf* s_rr = lds + RowAddr<L>::ROW_ELEMS;  // [16][LT]: conj(R_k)/N of the current block, thread-private columns
    const int u = threadIdx.x;
    int k1 = blockIdx.x;
    if ((half_flags & PAIR_ROWS) && N1 % 16 == 0) {  // mirror-row pairs on one XCD, see above
        const int b = blockIdx.x, j = 8 * (b / 16) + (b % 8), second = (b / 8) & 1;
        k1 = j == 0 ? (second ? N1 / 2 : 0) : (second ? N1 - j : j);
    }
    RowAddr<L> addr(0, u);
    const int C = 1 << log2C;
    constexpr bool no_fft = false;
    cf* base = work + (size_t)blockIdx.y * n_blocks * n_slots * N;
    const int s_end = ((half_flags & HALF_LAST) && k1 > N1 / 2) ? n_slots - 1 : n_slots;
    if (s_end <= 1) return;
    const bool mirrored = ref_half && k1 > N1 / 2;
    const unsigned off0 = (unsigned)(((u >> log2C) * N1 + k1) * C + (u & (C - 1)));
    const unsigned offr = mirrored ? (unsigned)(((u >> log2C) * N1 + (N1 - k1)) * C + (u & (C - 1))) : off0;
    const size_t qstride = (size_t)LT * N1;
    const unsigned off0b = off0 * (unsigned)sizeof(cf), offrb = offr * (unsigned)sizeof(cf);  // < 8*N: fits 32 bits
    const cf scale = mk(inv_n, mirrored ? inv_n : -inv_n);
    TwRegs<L> twr;
    twr.load(tw, u);
    const cf wb = tb[(size_t)k1 * LT + u];  // W_N^(k1*u)
    RowStepTw stw;
    stw.load(ts, k1);
    for (int s = 1; s < s_end; s += 2) {
        const bool two = s + 1 < s_end;
        cf acc_a[16], acc_b[16], x0[16], x1[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc_a[q] = acc_b[q] = mk(0.f, 0.f);
        auto issue = [&](cf(&x)[16], int k, int slot) {  // slot 0 = the reference row
            // uniform 64-bit row pointer + 32-bit byte offset per lane: the load takes its base from an SGPR pair
            gcptr src = (gcptr)(base + ((size_t)k * n_slots + slot) * N);
            unsigned off = slot ? off0b : offrb;
            size_t stride = qstride * sizeof(cf);
            // opaque: otherwise off + q*stride is hoisted as sixteen loop-invariant 64-bit VGPR offsets (32 registers
            // and a 64-bit add per load); this way the row pointers are scalar adds and the lane offset one VGPR
            asm volatile("" : "+s"(stride), "+v"(off));
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                x[q] = gload_s<2>(src, off);
                gstep(src, stride);
            }
            __builtin_amdgcn_sched_barrier(0);  // the loads go out HERE, ahead of the transform that follows
        };
        auto consume_ref = [&](cf(&x)[16]) {
            if (!no_fft) fft_regs<L, RowAddr<L>, true>(x, lds, u, addr, twr);
            if (mirrored && !no_fft) {  // conj(R[k1][k2]) = R[N1-k1][N2-1-k2]: the mirror row, read backwards
                lds_barrier();
                mirror_store(x, lds, addr, std::make_integer_sequence<int, 16>{});
                lds_barrier();
                mirror_load(x, lds, addr, std::make_integer_sequence<int, 16>{});
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) s_rr[q * LT + u] = x[q] * scale;  // conj(R_k)/N
        };
        auto consume_acc = [&](cf(&x)[16], cf(&acc)[16]) {
            if (!no_fft) fft_regs<L, RowAddr<L>, true>(x, lds, u, addr, twr);
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = cmac(acc[q], x[q], s_rr[q * LT + u]);
        };
        // one block: on entry `a` holds R_k (loads in flight).  two slots: R_k in a, A_k in b, B_k in a, and
        // R_(k+1) goes to b -- the buffers swap roles for the next block; one slot: R_(k+1) returns to a.
        auto block2 = [&](cf(&a)[16], cf(&b)[16], int k) {
            issue(b, k, s);
            consume_ref(a);
            issue(a, k, s + 1);
            consume_acc(b, acc_a);
            if (k + 1 < n_blocks) issue(b, k + 1, 0);
            consume_acc(a, acc_b);
        };
        auto block1 = [&](cf(&a)[16], cf(&b)[16], int k) {
            issue(b, k, s);
            consume_ref(a);
            if (k + 1 < n_blocks) issue(a, k + 1, 0);
            consume_acc(b, acc_a);
        };
        issue(x0, 0, 0);
        if (two) {
            for (int k = 0; k < n_blocks; k += 2) {
                block2(x0, x1, k);
                if (k + 1 < n_blocks) block2(x1, x0, k + 1);
            }
        } else {
            for (int k = 0; k < n_blocks; ++k) block1(x0, x1, k);
        }
        if (!no_fft) fft_regs<L, RowAddr<L>, true>(acc_a, lds, u, addr, twr);
        auto put = [&](const cf(&acc)[16], int slot) {
            gptr dst = (gptr)(base + (size_t)slot * N);
            size_t stride = qstride * sizeof(cf);
            unsigned off = off0b;
            asm volatile("" : "+s"(stride), "+v"(off));
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                gstore_s<4>(dst, off, cmul(acc[q], stw.times(wb, q)));  // W_N^(k1*(u + LT*q))
                gstep(dst, stride);
            }
        };
        put(acc_a, s);
        if (two) {
            if (!no_fft) fft_regs<L, RowAddr<L>, true>(acc_b, lds, u, addr, twr);
            put(acc_b, s + 1);
        }
    }
}

// --------------------------------------------------------------------------------------------
// Block-segmented mid pass, ONE sweep: up to four accumulator rows, so every reference row of a solve with up to
// eight candidates (four packed slots) is loaded and transformed once -- 19 instead of 22 row transforms per row for
// seven candidates, and the reference rows are read once instead of twice.  The accumulators alone are 128 VGPRs; it
// fits in 256 (two blocks per CU) because nothing else stays live across items: conj(R_k)/N is parked in LDS, row
// pointers are scalar, lane offsets opaque to the optimiser (no hoisted address tables).  PF: the loads of the next
// item are issued into a staging buffer before the current item is transformed (copied to the work buffer when it is
// its turn: sixteen moves per item buy static buffer roles, i.e. one copy of the transform code per item kind).
// NA = accumulator rows: 4 in general; solves with ONE packed slot (one or two candidates: every FFTAligner.fit) use the
// NA = 1 instantiation, which keeps conj(R_k)/N in registers, needs the row buffer only and fits three blocks on a CU
// (the transform core runs 9 % faster there: profiles/fft_core_rate.hip).
template <int L, bool PF, int NA = 4>
__global__ __launch_bounds__(256, NA == 1 ? 3 : 2) void k_mid_seg_one(cf* __restrict__ work, int N1, int log2C, long long N,
                                                                      int n_slots, int n_blocks, float inv_n,
                                                                      const cf* __restrict__ tw, const cf* __restrict__ tb,
                                                                      const cf* __restrict__ ts, int half_flags) {
    static_assert(L == 4096, "one row per 256-thread block");
    const int ref_half = half_flags & HALF_REF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int LT = L / 16;
    // conj(R_k)/N of the current block: parked in LDS ([16][LT], thread-private columns) next to four accumulator rows,
    // in registers next to one (then the block needs the row buffer only and three blocks share a CU)
    constexpr bool RR_REGS = NA == 1;
    cf* s_rr = lds + RowAddr<L>::ROW_ELEMS;
    cf rr[RR_REGS ? 16 : 1];
    const int u = threadIdx.x;
    int k1 = blockIdx.x;
    if ((half_flags & PAIR_ROWS) && N1 % 16 == 0) {  // mirror-row pairs on one XCD, see above
        const int b = blockIdx.x, j = 8 * (b / 16) + (b % 8), second = (b / 8) & 1;
        k1 = j == 0 ? (second ? N1 / 2 : 0) : (second ? N1 - j : j);
    }
    RowAddr<L> addr(0, u);
    const int C = 1 << log2C;
    // section experiments (lab build only, profiles/mid_sections.py): DBG_HOT_MEM makes every pair use pair 0's buffers
    // (all traffic becomes L2 hits: compute + LDS + issue only), DBG_NO_FFT drops the row transforms (memory only)
    const bool no_fft = FFS_LABF(half_flags, DBG_NO_FFT);
    cf* base = work + (size_t)(FFS_LABF(half_flags, DBG_HOT_MEM) ? 0 : blockIdx.y) * n_blocks * n_slots * N;
    const int s_end = ((half_flags & HALF_LAST) && k1 > N1 / 2) ? n_slots - 1 : n_slots;
    const int na = s_end - 1;  // candidate slots of this row: 1..4 (the host falls back to k_mid_seg_pipe beyond)
    if (na <= 0) return;
    const bool mirrored = ref_half && k1 > N1 / 2;
    const unsigned off0b = (unsigned)(((u >> log2C) * N1 + k1) * C + (u & (C - 1))) * (unsigned)sizeof(cf);
    const unsigned offrb =
        mirrored ? (unsigned)(((u >> log2C) * N1 + (N1 - k1)) * C + (u & (C - 1))) * (unsigned)sizeof(cf) : off0b;
    const cf scale = mk(inv_n, mirrored ? inv_n : -inv_n);
    TwRegs<L> twr;
    twr.load(tw, u);
    cf acc[NA][16], x[16], xl[PF ? 16 : 1];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[a][q] = mk(0.f, 0.f);
    // request row (block k, slot) into `dst`: a scalar row pointer that advances by the (opaque) stride + one 32-bit
    // lane offset -- no per-load vector address arithmetic, no table of hoisted 64-bit offsets
    auto request = [&](cf* dst, int k, int slot) {
        gcptr p = (gcptr)(base + ((size_t)k * n_slots + slot) * N);
        size_t stride = (size_t)LT * N1 * sizeof(cf);
        unsigned off = slot ? off0b : offrb;
        asm volatile("" : "+s"(stride), "+v"(off));
        if (slot) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                dst[q] = gload_s<2>(p, off);
                gstep(p, stride);
            }
        } else {
            // reference rows are read twice (as themselves and as the mirror of row N1 - k1): cached loads -- the same
            // pace as streaming ones (7.55 vs 7.51 us/pair) at 33.9 instead of 34.8 MB of HBM traffic per pair
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                dst[q] = gload(p, off);
                gstep(p, stride);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // the loads go out HERE, ahead of the transform that follows
    };
    // bring item (k, slot) into x; with PF it was requested one item earlier into xl, and (nk, nslot) is requested now
    auto fetch = [&](int k, int slot, int nk, int nslot) {
        if constexpr (PF) {
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = xl[q];
            if (nk < n_blocks) request(xl, nk, nslot);
        } else {
            request(x, k, slot);
        }
    };
    if constexpr (PF) request(xl, 0, 0);
    for (int k = 0; k < n_blocks; ++k) {
        fetch(k, 0, k, 1);
        if (!no_fft) fft_regs<L, RowAddr<L>, true>(x, lds, u, addr, twr);
        if (mirrored && !no_fft) {  // conj(R[k1][k2]) = R[N1-k1][N2-1-k2]: the mirror row, read backwards
            lds_barrier();
            mirror_store(x, lds, addr, std::make_integer_sequence<int, 16>{});
            lds_barrier();
            mirror_load(x, lds, addr, std::make_integer_sequence<int, 16>{});
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {  // conj(R_k)/N
            if constexpr (RR_REGS)
                rr[q] = x[q] * scale;
            else
                s_rr[q * LT + u] = x[q] * scale;
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            if (a >= na) break;
            const bool last = a + 1 == na;
            fetch(k, 1 + a, last ? k + 1 : k, last ? 0 : 2 + a);
            if (!no_fft) fft_regs<L, RowAddr<L>, true>(x, lds, u, addr, twr);
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][q] = cmac(acc[a][q], x[q], RR_REGS ? rr[q] : s_rr[q * LT + u]);
        }
    }
    cf wbl = tb[(size_t)k1 * LT + u];  // W_N^(k1*u)
    RowStepTw stw;
    stw.load(ts, k1);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        if (a >= na) break;
        if (!no_fft) fft_regs<L, RowAddr<L>, true>(acc[a], lds, u, addr, twr);
        if FFS_LABF(half_flags, DBG_NO_STORE) {
#pragma unroll
            for (int q = 0; q < 16; ++q) asm volatile("" ::"v"(acc[a][q]));
            continue;
        }
        gptr dst = (gptr)(base + (size_t)(1 + a) * N);
        size_t stride = (size_t)LT * N1 * sizeof(cf);
        unsigned off = off0b;
        asm volatile("" : "+s"(stride), "+v"(off), "+v"(wbl));  // opaque: the sixteen products wb*ts[q] are not hoisted
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            gstore_s<4>(dst, off, cmul(acc[a][q], stw.times(wbl, q)));  // W_N^(k1*(u + LT*q))
            gstep(dst, stride);
        }
    }
}

// Slot map: a pair owns n_slots = 1 + n_packed consecutive length-N buffers -- slot 0 = the reference transform,
// slot 1 + k = packed candidate transform k (updated in place by the mid pass).
FFS_DEV int slot_stride(int n_slots) { return n_slots; }
FFS_DEV int cand_slot(int, int kp, int) { return 1 + kp; }

struct WinParams {
    int lo[2], hi[2];
    float marg[2];
    int seg, shift;  // block-segmented mode: output index m is lag m + shift (no wrap-around)
    // the window in output-index space: m passes iff (unsigned)(m - ra[h][i]) <= rw[h][i] for i = 0 or 1
    int ra[2][2];
    unsigned rw[2][2];
    // the same window as ONE range test for callers whose m is always a valid output index (0 <= m < N): the
    // window is either one range of m or -- lags on both sides of zero -- everything but one gap in the middle:
    // m passes iff ((unsigned)(m - g0[h]) <= gw[h]) != ginv[h]
    int g0[2];
    unsigned gw[2];
    int ginv[2];
};

// lag of output index m: circular (lags 0..d_hi at m = d, negative ones at m = d + N) or, in
// block-segmented mode, the plain shift
FFS_DEV int lag_of(const WinParams& wp, int h, int m, int nN) {
    if (wp.seg) return m + wp.shift;
    return (m <= wp.hi[h]) ? m : m - nN;
}

// Is output index m (>= 0; -1 = "no value") inside half h's lag window?  Two unsigned range tests.
FFS_DEV bool in_window(const WinParams& wp, int h, int m) {
    return (unsigned)(m - wp.ra[h][0]) <= wp.rw[h][0] || (unsigned)(m - wp.ra[h][1]) <= wp.rw[h][1];
}
// ALLM: every m the caller passes is a valid output index -- one range test (two instructions per value)
template <bool ALLM>
FFS_DEV bool in_window_t(const WinParams& wp, int h, int m) {
    if constexpr (ALLM)
        return ((unsigned)(m - wp.g0[h]) <= wp.gw[h]) != (wp.ginv[h] != 0);
    else
        return in_window(wp, h, m);
}

FFS_DEV WinParams load_window(const CandDesc* __restrict__ cands, int cand0, int kp, int n_cand, int nN, int seg_shift = 0,
                              int seg = 0) {
    WinParams w;
    w.seg = seg;
    w.shift = seg_shift;
    constexpr int NEVER = 0x40000000;  // start of an empty range (width 0): no index ever equals it
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const bool present = (2 * kp + h) < n_cand;
        const CandDesc& cd = cands[cand0 + (present ? 2 * kp + h : 0)];
        w.lo[h] = cd.d_lo;
        w.hi[h] = (present && !(cd.flags & CAND_NO_LAGS)) ? cd.d_hi : cd.d_lo - 1;  // absent/empty: nothing passes
        w.marg[h] = cd.margin;
        const int lo = w.lo[h], hi = w.hi[h];
        w.ra[h][0] = w.ra[h][1] = NEVER;
        w.rw[h][0] = w.rw[h][1] = 0u;
        w.g0[h] = NEVER;
        w.gw[h] = 0u;
        w.ginv[h] = 0;
        if (hi >= lo) {
            if (seg) {  // m = d - shift
                w.ra[h][0] = lo - seg_shift;
                w.rw[h][0] = (unsigned)(hi - lo);
                w.g0[h] = lo - seg_shift;
                w.gw[h] = (unsigned)(hi - lo);
            } else {
                if (lo >= 0) {  // one range at m = d
                    w.g0[h] = lo;
                    w.gw[h] = (unsigned)(hi - lo);
                } else if (hi < 0) {  // one range at m = d + N
                    w.g0[h] = nN + lo;
                    w.gw[h] = (unsigned)(hi - lo);
                } else {  // m in [0, hi] or [N + lo, N - 1]: everything but the gap hi + 1 .. N + lo - 1
                    w.ginv[h] = 1;
                    if (nN + lo - 1 >= hi + 1) {
                        w.g0[h] = hi + 1;
                        w.gw[h] = (unsigned)(nN + lo - hi - 2);
                    }
                }
                if (hi >= 0) {  // lags 0..hi sit at m = d
                    const int a = lo > 0 ? lo : 0;
                    w.ra[h][0] = a;
                    w.rw[h][0] = (unsigned)(hi - a);
                }
                if (lo < 0) {  // negative lags at m = d + N
                    const int b = hi < -1 ? hi : -1;
                    w.ra[h][1] = nN + lo;
                    w.rw[h][1] = (unsigned)(b - lo);
                }
            }
        }
    }
    return w;
}

// Per-block maximum + near-tie nominees of the NV values each thread holds: v[q].x belongs to the
// real-part candidate, v[q].y to the imaginary-part one, at output index m_of(q) (< 0: no value).
// The window test of every value is done once (a bit per value); lags are only worked out for the few
// values that reach the tie margin.
// dm_b: added to m_of(q) for the imaginary half (0 normally; k_pass_c3's paired columns: the imaginary half is the
// SAME candidate's column C columns further on).
template <int NV, int NW, class MOf>
FFS_DEV void block_nominees(const cf* v, MOf m_of, const WinParams& wp, int nN, unsigned char* smem, int tid,
                            BlockNom* __restrict__ out_a, BlockNom* __restrict__ out_b, int dm_b = 0) {
    // callers with a whole column per thread (NV >= 16) only ever pass valid output indices
    constexpr bool ALLM = NV >= 16;
    float bv[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int m = m_of(q);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool ok = in_window_t<ALLM>(wp, h, h ? m + dm_b : m);
            const float val = h ? v[q].y : v[q].x;
            bv[h] = fmaxf(bv[h], ok ? val : -INFINITY);
        }
    }
    const float tmax[2] = {bv[0], bv[1]};  // this thread's own in-window maxima
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) bv[h] = fmaxf(bv[h], __shfl_xor(bv[h], sft, 64));
    }
    __syncthreads();  // callers are done with their LDS data; reuse it as scratch
    float* s_val = reinterpret_cast<float*>(smem);         // [NW][2]
    float* s_bmax = reinterpret_cast<float*>(smem + 512);  // [2]
    int* s_cnt = reinterpret_cast<int*>(smem + 528);       // [2]
    float* s_lval = reinterpret_cast<float*>(smem + 544);  // [2][KBLK]
    int* s_ld = reinterpret_cast<int*>(smem + 544 + 64);   // [2][KBLK]
    const int wave = tid / 64, lane = tid % 64;
    if (lane == 0) {
        s_val[wave * 2 + 0] = bv[0];
        s_val[wave * 2 + 1] = bv[1];
    }
    __syncthreads();
    if (tid < 2) {
        float fv = -INFINITY;
        for (int w = 0; w < NW; ++w) fv = fmaxf(fv, s_val[w * 2 + tid]);
        s_bmax[tid] = fv;
        s_cnt[tid] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float thr = s_bmax[h] - eff_margin(wp.marg[h], s_bmax[h]);
        // A thread holds a nominee iff its own maximum reaches the threshold, so nearly every wave skips the NV
        // per-value tests (each one a divergent branch around an LDS atomic) in one step; the window test is simply
        // repeated for the few threads that get here (keeping NV validity bits alive across the barriers cost more:
        // the compiler held them as 2*NV lane masks in scalar registers and spilled those).
        if (!(tmax[h] >= thr)) continue;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const float val = h ? v[q].y : v[q].x;
            int m = m_of(q) + (h ? dm_b : 0);
            asm volatile("" : "+v"(m));  // opaque: a fresh test here, not the first loop's results kept alive
            if (val >= thr && in_window_t<ALLM>(wp, h, m)) {
                const int slot = atomicAdd(&s_cnt[h], 1);
                if (slot < KBLK) {
                    s_lval[h * KBLK + slot] = val;
                    s_ld[h * KBLK + slot] = lag_of(wp, h, m, nN);
                }
            }
        }
    }
    __syncthreads();
    if (tid < 2) {
        BlockNom& o = tid ? *out_b : *out_a;
        o.bmax = s_bmax[tid];
        o.cnt = s_cnt[tid];
        for (int i = 0; i < KBLK; ++i) {
            o.val[i] = s_lval[tid * KBLK + i];
            o.d[i] = s_ld[tid * KBLK + i];
        }
    }
}

// Exhaustive variant for flagged candidates: append every in-window value within the margin of the
// candidate's global fp32 maximum to the pool.
// Every flagged candidate may append `quota` lags (the pool's capacity divided by the number of flagged
// candidates of the call's sub-batches), so one degenerate pair -- a silent reference, a wide window --
// cannot push its batch neighbours out of the pool: a candidate's answer never depends on the others.
template <int NV, class MOf>
FFS_DEV void block_collect_all(const cf* v, MOf m_of, const WinParams& wp, int nN, const int (&ci)[2],
                               const bool (&want)[2], const float (&thr)[2], PoolHeader* __restrict__ pool,
                               PoolEntry* __restrict__ entries, PoolBest* __restrict__ best, unsigned quota) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int m = m_of(q);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!want[h]) continue;
            const int d = lag_of(wp, h, m, nN);
            const bool ok = (m >= 0) && (d >= wp.lo[h]) && (d <= wp.hi[h]);
            const float val = h ? v[q].y : v[q].x;
            if (ok && val >= thr[h]) {
                const unsigned mine = atomicAdd(&best[ci[h]].count, 1u);
                const unsigned slot = mine < quota ? atomicAdd(&pool->count, 1u) : 0xffffffffu;
                if (slot < pool->capacity) {
                    PoolEntry e;
                    e.ci = ci[h];
                    e.d = d;
                    e.val = val;
                    e.pad = 0;
                    e.score = 0.0;
                    entries[slot] = e;
                } else {
                    best[ci[h]].overflow = 1u;
                }
            }
        }
    }
}

// Flagged candidates (nominee overflow, flag 2) of a sub-batch, appended by k_nominees: xlist[0] = count,
// xlist[1 + e] = local candidate index.  The exhaustive instantiations of the last pass walk this list
// (grid.y = a few rows, entry e handled by row e % gridDim.y) instead of launching one block per
// transform and tile that would nearly always exit at once.
//
// Does the listed half of transform kp need the exhaustive pass?
FFS_DEV bool exhaustive_wanted(const NomList* __restrict__ noms, const CandDesc* __restrict__ cands, int cand0, int kp,
                               int n_cand, int only_half, int (&ci)[2], bool (&want)[2], float (&thr)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const bool present = (2 * kp + h) < n_cand;
        ci[h] = cand0 + (present ? 2 * kp + h : 0);
        want[h] = present && h == only_half && (noms[ci[h]].flags & 2) != 0;
        thr[h] = want[h] ? noms[ci[h]].gmax - eff_margin(cands[ci[h]].margin, noms[ci[h]].gmax) : INFINITY;
    }
    return want[0] || want[1];
}

// --------------------------------------------------------------------------------------------
// pass C.  grid = (N2/C, n_candidate_transforms); same thread mapping as pass A.
// grid.y enumerates the candidate transforms of the pairs in flight: ly = lp*n_packed + k uses work
// slot lp*n_slots + 1 + k and candidates first_cand + lp*n_cand + {2k, 2k+1}.
// MODE 0: nominees; 1: write the full correlation (diagnostics); 2: exhaustive collect for flagged
// candidates (blocks of unflagged transforms exit immediately).
template <int L, int C, int MODE>
__global__ __launch_bounds__((L / 16) * C) void k_pass_c(const cf* __restrict__ work, int N2, long long N,
                                                         const cf* __restrict__ tw, const CandDesc* __restrict__ cands,
                                                         int first_cand, int n_cand, int n_packed, int n_slots,
                                                         BlockNom* __restrict__ bnom, float* __restrict__ out_a,
                                                         float* __restrict__ out_b, const NomList* __restrict__ noms,
                                                         PoolHeader* __restrict__ pool, PoolEntry* __restrict__ entries,
                                                         int log2CL, const cf* __restrict__ tw3,
                                                         const int* __restrict__ xlist, PoolBest* __restrict__ pbest,
                                                         int pool_shares, int half_last) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int LT = L / 16;
    constexpr int NT = LT * C;
    constexpr int NW = NT / 64;
    constexpr bool WRITE = (MODE == 1);
    const int tid = threadIdx.x;
    const int c = tid % C;
    const int u = tid / C;
    const int tile = blockIdx.x;
    for (int e = (MODE == 2) ? (int)blockIdx.y : 0;; e += gridDim.y) {
    int ly = blockIdx.y, only_half = -1;
    if (MODE == 2) {
        if (e >= xlist[0]) return;
        const int lc = xlist[1 + e];
        ly = (lc / n_cand) * n_packed + (lc % n_cand) / 2;
        only_half = (lc % n_cand) & 1;
        __syncthreads();  // the previous entry's LDS traffic is over
    }
    const int lp = ly / n_packed, kp = ly % n_packed;
    int xci[2];
    bool xwant[2];
    float xthr[2];
    if (MODE == 2) {
        if (!exhaustive_wanted(noms, cands, first_cand + lp * n_cand, kp, n_cand, only_half, xci, xwant, xthr)) continue;
    }
    const cf* in = work + (size_t)(lp * slot_stride(n_slots) + cand_slot(n_slots, kp, n_packed)) * N;
    typedef ColShape<L> CS;
    TwRegs<CS::LI> twr;
    twr.load(tw, CS::R3 ? u / 3 : u);
    cf v[16];
    if (half_last && kp == n_packed - 1) {
        // HALF_LAST: the single-candidate slot holds rows 0..L/2 only; its spectrum product is Hermitian, and
        // after the row pass that symmetry reads  Y[L-k1][m1] = conj(Y[k1][m1])  (same column!), so the
        // missing rows are the conjugates of stored ones
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = u + LT * q;
            const cf y = in[tile_base<L, C>(tile, c, log2CL) + ((size_t)(row <= L / 2 ? row : L - row) << log2CL)];
            v[q] = row <= L / 2 ? y : mk(y.x, -y.y);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = in[tile_base<L, C>(tile, c, log2CL) + ((size_t)(u + LT * q) << log2CL)];
    }
    cf* s_tw3 = lds + L * C;
    if constexpr (CS::R3) {
        for (int i = tid; i < L; i += NT) s_tw3[i] = tw3[i];
        __syncthreads();
    }
    col_fft<L, C>(v, lds, u, c, twr, s_tw3);
    // v[q] = out[m], m = m1 + N2*m2, m1 = tile*C + c, m2 = ob + OSTEP*q
    const int m1 = tile * C + c;
    const int ob = CS::out_base(u);
    if (WRITE) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const size_t m = (size_t)m1 + (size_t)N2 * (ob + CS::OSTEP * q);
            if (out_a) out_a[m] = v[q].x;
            if (out_b) out_b[m] = v[q].y;
        }
        return;
    }
    const WinParams wp = load_window(cands, first_cand + lp * n_cand, kp, n_cand, (int)N);
    if (MODE == 2) {
        block_collect_all<16>(
            v, [&](int q) { return m1 + N2 * (ob + CS::OSTEP * q); }, wp, (int)N, xci, xwant, xthr, pool, entries, pbest,
            pool->capacity / (unsigned)(pool_shares * (xlist[0] > 0 ? xlist[0] : 1)));
        continue;
    }
    block_nominees<16, NW>(
        v, [&](int q) { return m1 + N2 * (ob + CS::OSTEP * q); }, wp, (int)N, smem, tid,
        &bnom[((size_t)ly * 2 + 0) * gridDim.x + tile], &bnom[((size_t)ly * 2 + 1) * gridDim.x + tile]);
    return;
    }
}

// --------------------------------------------------------------------------------------------
#ifndef FFS_A3P_WAVES
#define FFS_A3P_WAVES 3
#endif
#ifndef FFS_C3P_CHUNK
#define FFS_C3P_CHUNK 16
#endif
#ifndef FFS_C3P_WAVES
#define FFS_C3P_WAVES 2  // paired-column instantiation with three sub-transforms: 205 VGPRs (37 spilled at three waves)
#endif
#ifndef FFS_C3_WAVES
#define FFS_C3_WAVES 3
#endif
// pass C for columns of length L = NS*LI (NS = 3: 192/384/768 rows; NS = 2: 512 rows), all sub-transforms of a column in
// one thread (colnr_fft): nominees only (MODE 0 of
// k_pass_c; the diagnostic and exhaustive modes stay with k_pass_c, the work layout is the same).
// grid = (N2/C, n_candidate_transforms); block = (LI/16)*C = 256 threads.
// PAIRED = the instantiation for a HALF_LAST slot (ONE real candidate; launched on its own: grid = (N2/C/2, n_pairs)):
// every column of that slot's product spectrum is Hermitian (rows above L/2 are the conjugates of stored rows, see
// k_pass_c), i.e. every output column is REAL -- so two columns share one complex transform, z = A + i*B -> Re = column
// A's correlation values, Im = column B's.  A block covers the tile pair (2t, 2t+1) (one whole 512-byte row chunk per
// row when C = 32): half the column transforms for this slot.  The other slots run the plain instantiation
// (grid.y = n_pairs * slots_here, slots_here = n_packed minus the half-last slot).
template <int NS, int LI, int C, bool PAIRED>
__global__ __launch_bounds__(256, (PAIRED && NS == 3) ? FFS_C3P_WAVES : FFS_C3_WAVES) void k_pass_c3(const cf* __restrict__ work, int N2, long long N,
                                                 const cf* __restrict__ tw, const CandDesc* __restrict__ cands,
                                                 int first_cand, int n_cand, int n_packed, int n_slots,
                                                 BlockNom* __restrict__ bnom, int log2CL, const cf* __restrict__ tw3,
                                                 int slots_here) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int L = NS * LI, LTI = LI / 16, NT = LTI * C, NW = NT / 64;
    static_assert(NT == 256, "256 threads per block");
    const int tid = threadIdx.x;
    const int c = tid % C, u = tid / C;
    const int tiles = N2 / C;  // block records per (transform, half)
    const int tile = PAIRED ? 2 * (int)blockIdx.x : (int)blockIdx.x;
    const int lp = PAIRED ? (int)blockIdx.y : (int)blockIdx.y / slots_here;
    const int kp = PAIRED ? n_packed - 1 : (int)blockIdx.y % slots_here;
    const int ly = lp * n_packed + kp;
    const cf* in = work + (size_t)(lp * slot_stride(n_slots) + cand_slot(n_slots, kp, n_packed)) * N;
    TwRegs<LI> twr;
    twr.load(tw, u);
    const cf wu = tw3[u], wu2 = tw3[2 * u];  // W_L^u, W_L^2u
    cf v[NS][16];
    if constexpr (PAIRED) {
        const unsigned oa = (unsigned)tile_base<L, C>(tile, c, log2CL) * (unsigned)sizeof(cf);
        const unsigned ob = (unsigned)tile_base<L, C>(tile + 1, c, log2CL) * (unsigned)sizeof(cf);
        const unsigned row_bytes = (unsigned)sizeof(cf) << log2CL;
        const gcptr pin = (gcptr)in;  // scalar base + one 32-bit byte offset per load
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            constexpr int CH = FFS_C3P_CHUNK;  // rows of A and of B requested together
#pragma unroll
            for (int q0 = 0; q0 < 16; q0 += CH) {
                cf a[CH], b[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int row = NS * (u + LTI * (q0 + j)) + g;
                    const unsigned ro = (unsigned)(row <= L / 2 ? row : L - row) * row_bytes;
                    a[j] = gload(pin, oa + ro);  // (plain loads: streaming ones measured 1.9 -> 2.0 us/pair here)
                    b[j] = gload(pin, ob + ro);
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int row = NS * (u + LTI * (q0 + j)) + g;
                    const float sg = row <= L / 2 ? 1.0f : -1.0f;  // conjugate the mirrored rows
                    v[g][q0 + j] = mk(a[j].x - sg * b[j].y, sg * a[j].y + b[j].x);  // A' + i*B'
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        // scalar row pointer stepping by the (opaque) stride of 3*LTI rows + one 32-bit lane offset per sub-transform
        size_t stride = ((size_t)(NS * LTI) << log2CL) * sizeof(cf);
        unsigned off = ((unsigned)tile_base<L, C>(tile, c, log2CL) + ((unsigned)(NS * u) << log2CL)) * (unsigned)sizeof(cf);
        const unsigned row_bytes = (unsigned)sizeof(cf) << log2CL;
        asm volatile("" : "+s"(stride), "+v"(off));
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            gcptr p = (gcptr)in;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                v[g][q] = gload_s<8>(p, off + g * row_bytes);
                gstep(p, stride);
            }
        }
    }
    colnr_fft<NS, LI, C>(v, lds, u, c, twr, wu, wu2);
    // v[r][q] = out[m], m = m1 + N2*m2, m1 = tile*C + c, m2 = u + LTI*q + LI*r
    const int m1 = tile * C + c;
    WinParams wp = load_window(cands, first_cand + lp * n_cand, kp, n_cand, (int)N);
    if constexpr (PAIRED) {  // the imaginary half carries the same candidate (columns m1 + C): same window, same margin
        wp.lo[1] = wp.lo[0], wp.hi[1] = wp.hi[0], wp.marg[1] = wp.marg[0];
        wp.g0[1] = wp.g0[0], wp.gw[1] = wp.gw[0], wp.ginv[1] = wp.ginv[0];
#pragma unroll
        for (int i = 0; i < 2; ++i) wp.ra[1][i] = wp.ra[0][i], wp.rw[1][i] = wp.rw[0][i];
    }
    // PAIRED: both halves are block records of part 0 of this transform (tiles `tile` and `tile + 1`)
    block_nominees<16 * NS, NW>(
        &v[0][0], [&](int i) { return m1 + N2 * (u + LTI * (i % 16) + LI * (i / 16)); }, wp, (int)N, smem, tid,
        &bnom[((size_t)ly * 2 + 0) * tiles + tile],
        PAIRED ? &bnom[((size_t)ly * 2 + 0) * tiles + tile + 1] : &bnom[((size_t)ly * 2 + 1) * tiles + tile], PAIRED ? C : 0);
}

// --------------------------------------------------------------------------------------------
// pass A for columns of length L = NS*LI, bit-packed inputs, all sub-transforms of a column in one thread (colnr_fft).
// grid = (N2/C, n_transforms); block = (LI/16)*C = 256 threads; same outputs as k_pass_a<L, ., 2>.
// tbR[u][n2] = W_N^(n2*u) (u < LTI), tsR[i][n2] = W_N^(n2*LTI*2^i) (i < 4), thR[r-1][n2] = W_N^(n2*LI*r) (0 < r < NS).
// Paired transforms (round 3; when the reference slot AND the last candidate slot are half slots; PM as in k_pass_a:
// 0 = none, 1 = every grid row a paired group of two transforms, 2 = xf_per_pair - 1 grid rows per group, row 0 the
// paired one): both hold ONE real vector, so their columns share one complex column transform -- z = ref + i*cand,
// Z = R + i*S with R, S Hermitian, hence  R[k] = (Z[k] + conj(Z[L-k]))/2,  S[k] = (Z[k] - conj(Z[L-k]))/(2i)  for the
// stored rows k <= L/2.  The mirror rows Z[L-k] sit in other threads of the same column: they go through the LDS tile
// once (rows >= L/2 only, one LI-row block at a time).  One column transform + one exchange instead of two column
// transforms.
template <int NS, int LI, int C, int PM>
__global__ __launch_bounds__(256, (PM != 0 && NS == 3) ? FFS_A3P_WAVES : FFS_C3_WAVES) void k_pass_a3(const XformDesc* __restrict__ descs, cf* __restrict__ work, int N2,
                                                 long long N, const cf* __restrict__ tw, const cf* __restrict__ tbR,
                                                 const cf* __restrict__ tsR, const cf* __restrict__ thR,
                                                 const cf* __restrict__ tw3, int log2CL, int xf_per_pair,
                                                 int slots_per_pair, int nt, int half_flags, int row_sel) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    constexpr int L = NS * LI, LTI = LI / 16, NT = LTI * C, CW = (C + 31) / 32;
    static_assert(NT == 256, "256 threads per block");
    // grid row -> (transform group, transform within the group, paired or not); row_sel as in k_pass_a
    const int sel_first = PM == 0 ? (row_sel & 0xffff) : 0, sel_count = PM == 0 ? (row_sel >> 16) : 0;
    const int rows_per_group = PM == 2 ? xf_per_pair - 1 : (PM == 1 ? 1 : (sel_count ? sel_count : xf_per_pair));
    auto desc_of = [&](int y, bool* is_paired) {
        const int g = y / rows_per_group, j = y % rows_per_group;
        *is_paired = PM != 0 && j == 0;
        return g * xf_per_pair + sel_first + j;
    };
    if (blockIdx.x >= (unsigned)nt) {  // input prefetch block (see prefetch_bit_inputs); distance in bits 16..23 of half_flags
        const int y = (int)blockIdx.y + ((half_flags >> 16) & 255);
        if (y >= (int)gridDim.y) return;
        bool pp;
        const int di = desc_of(y, &pp);
        const int n_desc = ((int)gridDim.y / rows_per_group) * xf_per_pair;
        prefetch_bit_inputs(descs, di, n_desc, (int)blockIdx.x - nt, L, N2, (nt / 8) * C, NT);
        if (pp) prefetch_bit_inputs(descs, di + xf_per_pair - 1, n_desc, (int)blockIdx.x - nt, L, N2, (nt / 8) * C, NT);
        return;
    }
    bool paired;
    const int desc_index = desc_of((int)blockIdx.y, &paired);
    const int group = (int)blockIdx.y / rows_per_group;
    const int c = threadIdx.x % C, u = threadIdx.x / C;
    const int tile = (nt % 8 == 0) ? (int)(blockIdx.x % 8) * (nt / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;  // see k_pass_a
    const int n2 = tile * C + c;
    XformDesc d = descs[desc_index];
    if (PM != 0 && paired) {  // block-uniform: the imaginary half = the single candidate of the group's last transform
        const XformDesc dl = descs[desc_index + xf_per_pair - 1];
        d.b = dl.a, d.off_b = dl.off_a, d.len_b = dl.len_a, d.lead_b = dl.lead_a, d.b0 = dl.a0, d.b1 = dl.a1;
    }
    TwRegs<LI> twr;
    twr.load(tw, u);
    const cf wu = tw3[u], wu2 = tw3[2 * u];
    const cf b0 = tbR[(size_t)u * N2 + n2];
    const cf g1 = tsR[n2], g2 = tsR[(size_t)N2 + n2], g4 = tsR[(size_t)2 * N2 + n2], g8 = tsR[(size_t)3 * N2 + n2];
    const cf h1 = thR[n2], h2 = NS == 3 ? thR[(size_t)N2 + n2] : h1;
    // aligned C-bit windows of all L rows of both vectors in LDS (as in k_pass_a's bit path)
    unsigned* stage = reinterpret_cast<unsigned*>(smem);  // [2][L][CW]
    const int col0 = tile * C;
    for (int i = threadIdx.x; i < 2 * L * CW; i += NT) {
        const int h = i / (L * CW), row = (i / CW) % L, j = i % CW;
        const auto* src = (const __attribute__((address_space(1))) unsigned*)(h ? d.b : d.a);
        const int off = h ? d.off_b : d.off_a, len = h ? d.len_b : d.len_a, lead = h ? d.lead_b : d.lead_a;
        const int bit0 = off + row * N2 + col0 + 32 * j;
        const int w = bit0 >> 5;
        const int w_lo = (off + lead) >> 5, w_hi = (off + len - 1) >> 5;
        unsigned d0 = 0, d1 = 0;
        if (len > lead) {
            if (w >= w_lo && w <= w_hi) d0 = src[w];
            if (w + 1 >= w_lo && w + 1 <= w_hi) d1 = src[w + 1];
        }
        stage[i] = __builtin_amdgcn_alignbit(d1, d0, (unsigned)bit0 & 31u);
    }
    __syncthreads();
    cf v[NS][16];
    const unsigned sh = c & 31;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned* win = stage + h * (L * CW) + (NS * u) * CW + (c >> 5);
        const float v0 = h ? d.b0 : d.a0, v1 = h ? d.b1 : d.a1;
        const int len = h ? d.len_b : d.len_a, lead = h ? d.lead_b : d.lead_a;
        // rows [0, rows_full) lie inside the vector for every column of the tile, rows >= rows_any outside for all
        // of them: per q (rows NS*LTI*q .. NS*LTI*(q+1)-1 over the block) the class is block-uniform
        const int col_end = col0 + C;
        const int rows_full = (lead == 0 && len >= col_end) ? (len - col_end) / N2 + 1 : 0;
        const int rows_any = (len > col0) ? (len - col0 + N2 - 1) / N2 : 0;
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            // the sixteen window words of sub-transform g are requested back to back and pinned: left alone, the
            // compiler sinks every read into the block-uniform branches below and waits for each one separately
            // (one exposed LDS round trip per sample: 64-96 per thread)
            unsigned w[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) w[q] = win[(NS * LTI * q + g) * CW];
#pragma unroll
            for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(w[q]));
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r0 = NS * LTI * q;
                float x = 0.0f;
                if (r0 < rows_any) {
                    const unsigned b = (unsigned)__builtin_amdgcn_sbfe((int)w[q], sh, 1u);  // 0 or ~0
                    x = pick_level<true>(b, v0, v1);
                    if (r0 + NS * LTI > rows_full) {
                        const int n = (NS * (u + LTI * q) + g) * N2 + n2;
                        x = (n >= lead && n < len) ? x : 0.0f;
                    }
                }
                if (h)
                    v[g][q].y = x;
                else
                    v[g][q].x = x;
            }
        }
    }
    colnr_fft<NS, LI, C>(v, lds, u, c, twr, wu, wu2);
    // v[r][q] = Y[k1 = u + LTI*q + LI*r][n2]; twiddle W_N^(n2*k1) = b0 * g^q * h_r
    cf wq[16];
    wq[0] = b0;
    wq[1] = cmul(b0, g1);
    wq[2] = cmul(b0, g2);
    wq[3] = cmul(wq[1], g2);
#pragma unroll
    for (int q = 0; q < 4; ++q) wq[4 + q] = cmul(wq[q], g4);
#pragma unroll
    for (int q = 0; q < 8; ++q) wq[8 + q] = cmul(wq[q], g8);
    const unsigned o0 = ((unsigned)tile_base<L, C>(tile, c, log2CL) + ((unsigned)u << log2CL)) * (unsigned)sizeof(cf);
    const unsigned ostep = ((unsigned)LTI << log2CL) * (unsigned)sizeof(cf);
    auto row_off = [&](int r, int q) { return o0 + ostep * q + (((unsigned)(LI * r)) << log2CL) * (unsigned)sizeof(cf); };
    auto tw_of = [&](int r, int q) { return r == 0 ? wq[q] : cmul(wq[q], r == 1 ? h1 : h2); };
    if (PM != 0 && paired) {
        char* out_r = reinterpret_cast<char*>(work + ((size_t)group * slots_per_pair) * N);
        char* out_s = reinterpret_cast<char*>(work + ((size_t)group * slots_per_pair + slots_per_pair - 1) * N);
        // row k1 <= L/2 from Z[k1] = z and its mirror Z[L-k1] = m
        auto emit = [&](int r, int q, cf z, cf m) {
            const cf w = tw_of(r, q);
            const cf a = mk(0.5f * (z.x + m.x), 0.5f * (z.y - m.y));   // (z + conj(m)) / 2
            const cf b = mk(0.5f * (z.y + m.y), -0.5f * (z.x - m.x));  // (z - conj(m)) / (2i)
            FFS_PA_STORE(out_r, row_off(r, q), cmul(a, w));
            FFS_PA_STORE(out_s, row_off(r, q), cmul(b, w));
        };
        if (u == 0) emit(0, 0, v[0][0], v[0][0]);  // row 0 mirrors onto itself
#pragma unroll
        for (int rr = NS - 1; rr >= 1; --rr) {
            // rows [LI*rr, LI*(rr+1)) through the tile; they are the mirrors of rows (L - LI*(rr+1), L - LI*rr]
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; ++q) lds[(u + LTI * q) * C + c] = v[rr][q];
            __syncthreads();
            constexpr int dummy = 0;
            (void)dummy;
            const int lo = L - LI * (rr + 1), hi = L - LI * rr;  // compile-time after unrolling
#pragma unroll
            for (int r = 0; r < NS; ++r)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int k_min = LTI * q + LI * r, k_max = k_min + LTI - 1;
                    if (k_max <= lo || k_min > hi || k_min > L / 2) continue;  // no lane of this (r, q) is in range
                    const int k1 = u + k_min;
                    if (k1 > lo && k1 <= hi && k1 <= L / 2) emit(r, q, v[r][q], lds[(L - k1 - LI * rr) * C + c]);
                }
        }
    } else {
        const int xi = desc_index - group * xf_per_pair;
        cf* out = work + ((size_t)group * slots_per_pair + xi) * N;
        const int k1_end = (((half_flags & HALF_REF) && xi == 0) || ((half_flags & HALF_LAST) && xi == xf_per_pair - 1))
                               ? L / 2 + 1 : L;
        char* outb = reinterpret_cast<char*>(out);
#pragma unroll
        for (int r = 0; r < NS; ++r) {
            if (LI * r >= k1_end) break;  // block-uniform: the whole third of the rows is beyond the stored half
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k1 = u + LTI * q + LI * r;
                if (k1 < k1_end) FFS_PA_STORE(outb, row_off(r, q), cmul(v[r][q], tw_of(r, q)));
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// pass C, pruned to the output bins the lag window can reach.  The last pass produces
// out[m1 + N2*m2]; a window of +-max_offset lags only ever touches a few values of m2 (four of 512
// for +-6000 lags at N = 2^21), so instead of the full length-N1 column transform each needed bin is
// evaluated directly, out[m2] = sum_k1 Y[k1] * W_N1^(k1*m2): one streaming read of the tile, a
// handful of multiply-adds per element, no LDS exchange of the data.  Used when the union of bins
// over the call's candidates has at most MAXBINS entries; otherwise k_pass_c runs.
constexpr int MAXBINS = 8;
// W_16^e = exp(-2*pi*i*e/16), (re, im) pairs
static __constant__ float c_w16[32] = {
    1.0f, -0.0f, FFS_COS_PI_8, -FFS_SIN_PI_8, FFS_SQRT_HALF, -FFS_SQRT_HALF, FFS_SIN_PI_8, -FFS_COS_PI_8,
    0.0f, -1.0f, -FFS_SIN_PI_8, -FFS_COS_PI_8, -FFS_SQRT_HALF, -FFS_SQRT_HALF, -FFS_COS_PI_8, -FFS_SIN_PI_8,
    -1.0f, 0.0f, -FFS_COS_PI_8, FFS_SIN_PI_8, -FFS_SQRT_HALF, FFS_SQRT_HALF, -FFS_SIN_PI_8, FFS_COS_PI_8,
    0.0f, 1.0f, FFS_SIN_PI_8, FFS_COS_PI_8, FFS_SQRT_HALF, FFS_SQRT_HALF, FFS_COS_PI_8, FFS_SIN_PI_8};
struct BinList {
    int n;
    int b[MAXBINS];  // signed bin offsets (m2 or m2 - N1)
};

template <int L, int C, bool EXH>
__global__ __launch_bounds__((L / 16) * C) void k_pass_c_pruned(const cf* __restrict__ work, int N2, long long N,
                                                                const cf* __restrict__ twn1,
                                                                const CandDesc* __restrict__ cands, int first_cand,
                                                                int n_cand, int n_packed, int n_slots,
                                                                BlockNom* __restrict__ bnom, BinList bins,
                                                                const NomList* __restrict__ noms,
                                                                PoolHeader* __restrict__ pool,
                                                                PoolEntry* __restrict__ entries, int log2CL,
                                                                const int* __restrict__ xlist, int seg, int seg_shift,
                                                                PoolBest* __restrict__ pbest, int pool_shares,
                                                                int half_last) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LT = L / 16;
    constexpr int NT = LT * C;
    constexpr int NW = NT / 64;
    constexpr int NG = LT;                         // partial sums per (bin, column): one per thread row u
    cf* s_tw = reinterpret_cast<cf*>(smem + 1024);                     // [L]   W_L^k
    cf* s_part = reinterpret_cast<cf*>(smem + 1024 + L * sizeof(cf));  // [MAXBINS][NG][C]
    const int tid = threadIdx.x;
    const int c = tid % C;
    const int u = tid / C;
    const int tile = blockIdx.x;
    for (int e = EXH ? (int)blockIdx.y : 0;; e += gridDim.y) {
    int ly = blockIdx.y, only_half = -1;
    if (EXH) {
        if (e >= xlist[0]) return;
        const int lc = xlist[1 + e];
        ly = (lc / n_cand) * n_packed + (lc % n_cand) / 2;
        only_half = (lc % n_cand) & 1;
        __syncthreads();  // the previous entry's LDS traffic is over
    }
    const int lp = ly / n_packed, kp = ly % n_packed;
    int xci[2];
    bool xwant[2];
    float xthr[2];
    if (EXH) {
        if (!exhaustive_wanted(noms, cands, first_cand + lp * n_cand, kp, n_cand, only_half, xci, xwant, xthr)) continue;
    }
#if FFS_NT & 8
#define FFS_PC_LOAD(base, ...) gload_s<8>((gcptr)(base), (unsigned)((__VA_ARGS__) * sizeof(cf)))
#else
#define FFS_PC_LOAD(base, ...) ((base)[(__VA_ARGS__)])
#endif
    const cf* in = work + (size_t)(lp * slot_stride(n_slots) + cand_slot(n_slots, kp, n_packed)) * N;
    cf v[16];
    if (half_last && kp == n_packed - 1) {
        // HALF_LAST: rows 0..L/2 only (see k_pass_c): sum_k1 Y[k1] W^(k1 m2) over all rows equals
        // Y[0] + (-1)^m2 Y[L/2] + 2 Re sum_{0<k1<L/2} Y[k1] W^(k1 m2) -- only the real part is used
        // (the slot's imaginary candidate does not exist), so weighting the stored rows is all it takes
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = u + LT * q;
            const float wgt = (row == 0 || row == L / 2) ? 1.0f : 2.0f;
            v[q] = mk(0.f, 0.f);
            if (row <= L / 2) {
                const cf y = FFS_PC_LOAD(in, tile_base<L, C>(tile, c, log2CL) + ((size_t)row << log2CL));
                v[q] = mk(wgt * y.x, wgt * y.y);
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q)
            v[q] = FFS_PC_LOAD(in, tile_base<L, C>(tile, c, log2CL) + ((size_t)(u + LT * q) << log2CL));
    }
    for (int i = tid; i < L; i += NT) s_tw[i] = twn1[i];
    __syncthreads();
    // W_L^(b*(u + LT*q)) = W_L^(b*u) * W_16^(b*q)  (L = 16*LT): the second factor is the same for the
    // whole block, so the sum over q runs on scalar constants (two packed FMAs per term) and the
    // per-thread factor is applied once per bin.
    for (int i = 0; i < bins.n; ++i) {
        const unsigned b = (unsigned)(bins.b[i] + L) % L;
        cf a = mk(0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const unsigned e = (b * q) & 15u;
            a = cmac_k(a, v[q], c_w16[2 * e], c_w16[2 * e + 1]);
        }
        s_part[(i * NG + u) * C + c] = cmul(a, s_tw[((unsigned)u * b) % L]);
    }
    __syncthreads();
    // thread t finishes (bin, column) pairs t, t + NT, ...  (more than one only when LT < MAXBINS)
    constexpr int NVF = (MAXBINS + LT - 1) / LT;
    cf val[NVF];
    int mm[NVF];
#pragma unroll
    for (int j = 0; j < NVF; ++j) {
        val[j] = mk(0.f, 0.f);
        mm[j] = -1;
        const int idx = tid + j * NT;
        if (idx < bins.n * C) {
            const int i = idx / C, cc = idx % C;
            for (int g = 0; g < NG; ++g) {
                const cf p = s_part[(i * NG + g) * C + cc];
                val[j].x += p.x;
                val[j].y += p.y;
            }
            const int m2 = (bins.b[i] + L) % L;
            mm[j] = tile * C + cc + N2 * m2;
        }
    }
    const WinParams wp = load_window(cands, first_cand + lp * n_cand, kp, n_cand, (int)N, seg_shift, seg);
    if (EXH) {
        block_collect_all<NVF>(
            val, [&](int j) { return mm[j]; }, wp, (int)N, xci, xwant, xthr, pool, entries, pbest,
            pool->capacity / (unsigned)(pool_shares * (xlist[0] > 0 ? xlist[0] : 1)));
        continue;
    }
    block_nominees<NVF, NW>(
        val, [&](int j) { return mm[j]; }, wp, (int)N, smem, tid, &bnom[((size_t)ly * 2 + 0) * gridDim.x + tile],
        &bnom[((size_t)ly * 2 + 1) * gridDim.x + tile]);
    return;
    }
}

// --------------------------------------------------------------------------------------------
// nominee gather: one wave per candidate. cand index ci = first_cand + blockIdx.x maps to
// block-nominee row (local transform, half) = (lx, h) given by cand_xf[ci - first_cand].
__global__ __launch_bounds__(64) void k_nominees(const BlockNom* __restrict__ bnom, int tiles, int n_cand,
                                                 int n_packed, const CandDesc* __restrict__ cands,
                                                 NomList* __restrict__ noms, int first_cand,
                                                 int* __restrict__ xlist) {
    const int ci = first_cand + blockIdx.x;
    const int lp = blockIdx.x / n_cand, jc = blockIdx.x % n_cand;
    const int row = (lp * n_packed + jc / 2) * 2 + (jc & 1);
    const CandDesc& cd = cands[ci];
    NomList& nl = noms[ci];
    const int lane = threadIdx.x;
    if (cd.flags & CAND_NO_LAGS) {
        if (lane == 0) {
            nl.count = 0;
            nl.flags = 1;
            nl.gmax = -INFINITY;
        }
        return;
    }
    const BlockNom* rows = bnom + (size_t)row * tiles;
    float g = -INFINITY;
    for (int t = lane; t < tiles; t += 64) g = fmaxf(g, rows[t].bmax);
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) g = fmaxf(g, __shfl_xor(g, sft, 64));
    __shared__ int s_count;
    __shared__ int s_flags;
    if (lane == 0) {
        s_count = 0;
        s_flags = 0;
    }
    __syncthreads();
    const float thr = g - eff_margin(cd.margin, g);
    for (int t = lane; t < tiles; t += 64) {
        const BlockNom& b = rows[t];
        if (b.bmax >= thr) {
            if (b.cnt > KBLK) atomicOr(&s_flags, 2);
            const int n = b.cnt < KBLK ? b.cnt : KBLK;
            for (int i = 0; i < n; ++i) {
                if (b.val[i] >= thr) {
                    const int slot = atomicAdd(&s_count, 1);
                    if (slot < KNOM) {
                        nl.d[slot] = b.d[i];
                        nl.val[slot] = b.val[i];
                    } else {
                        atomicOr(&s_flags, 2);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (lane == 0) {
        nl.count = s_count < KNOM ? s_count : KNOM;
        nl.flags = s_flags;
        nl.gmax = g;
        if (s_flags & 2) xlist[1 + atomicAdd(&xlist[0], 1)] = blockIdx.x;  // exhaustive sweep wanted
    }
}

// --------------------------------------------------------------------------------------------
// exact re-evaluation of c(d) = sum_{i in overlap} s'[i] * r'[i+d] for every nominee of a candidate.
// grid = (RSEG, n_cands): block x takes the x-th contiguous segment of the overlap and streams it
// 16 bytes per lane per step (unaligned dwordx4 loads; the reference side is shifted by d).
// Two-level inputs: three integer counts (n11, n1x, nx1) via byte-flag popcounts; float inputs:
// fp64 dot product.  Partial sums are added to acc[ci*KNOM + nominee] with atomics.
FFS_DEV unsigned nz_flags(unsigned w) {  // bit 7 of every byte that is non-zero
    return (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;
}

// Bit-packed vectors (DT == 2): bit i of a vector = (word[i >> 5] >> (i & 31)) & 1.
FFS_DEV unsigned get_bit(const void* p, int i) { return (reinterpret_cast<const unsigned*>(p)[i >> 5] >> (i & 31)) & 1u; }

// Exact counts over i in [a, b) of bit-packed s and r:  n11 += #(s[i] & r[i+d]), n1x += #s[i], nx1 += #r[i+d].
// Thread tid of nt takes every nt-th group of four s-words (128 samples); the r side is funnel-shifted into
// place (v_alignbit).  Requires 0 <= a, b <= S and 0 <= a + d, b + d <= R.
FFS_DEV void bit_counts(const void* sp, const void* rp, int R, int d, int a, int b, int tid, int nt, unsigned& n11,
                        unsigned& n1x, unsigned& nx1) {
    if (a >= b) return;
    const unsigned* __restrict__ s = reinterpret_cast<const unsigned*>(sp);
    const unsigned* __restrict__ r = reinterpret_cast<const unsigned*>(rp);
    const int w_first = a >> 5, w_last = (b - 1) >> 5, wr_max = (R - 1) >> 5;
    for (int w0 = w_first + 4 * tid; w0 <= w_last; w0 += 4 * nt) {
        const int bit0 = 32 * w0 + d;  // r bit under bit 0 of s-word w0
        const int jr = bit0 >> 5;      // arithmetic shift: floor
        const unsigned sh = (unsigned)bit0 & 31u;
        unsigned rw[5];
        if (jr >= 0 && jr + 4 <= wr_max && w0 > w_first && w0 + 3 < w_last) {
            // interior: four whole s-words (one 16-byte load; vectors start on 64-byte boundaries) against five r-words
            // (a 4-byte aligned 16-byte load + one dword), no edge masks
            uint4 sv, rv4;
            __builtin_memcpy(&sv, s + w0, 16);
            __builtin_memcpy(&rv4, r + jr, 16);
            const unsigned r4 = r[jr + 4];
            const unsigned sw4[4] = {sv.x, sv.y, sv.z, sv.w}, rr[5] = {rv4.x, rv4.y, rv4.z, rv4.w, r4};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned rv = __builtin_amdgcn_alignbit(rr[k + 1], rr[k], sh);
                n11 += __popc(sw4[k] & rv);
                n1x += __popc(sw4[k]);
                nx1 += __popc(rv);
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) rw[k] = (jr + k >= 0 && jr + k <= wr_max) ? r[jr + k] : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int w = w0 + k;
            if (w <= w_last) {
                unsigned m = 0xffffffffu;
                if (w == w_first) m &= 0xffffffffu << (a & 31);
                if (w == w_last) m &= 0xffffffffu >> (31 - ((b - 1) & 31));
                const unsigned sw = s[w] & m;
                const unsigned rv = __builtin_amdgcn_alignbit(rw[k + 1], rw[k], sh) & m;
                n11 += __popc(sw & rv);
                n1x += __popc(sw);
                nx1 += __popc(rv);
            }
        }
    }
}

// Float inputs: the mapped sample x' = 2x - 1 in fp64 (DT 1: fp32 samples, DT 3: fp64 samples, exactly the
// reference's arithmetic, aligners.py:55-57).
template <int DT>
FFS_DEV double mapped_sample(const void* p, int i) {
    if (DT == 3) return 2.0 * reinterpret_cast<const double*>(p)[i] - 1.0;
    return 2.0 * (double)reinterpret_cast<const float*>(p)[i] - 1.0;
}

// Mixed element types (DT == 4): the mapped sample x' of either vector in fp64, whatever its storage -- a two-level
// byte / bit picks the fp64 level value (v0, v1 = 2*lo-1, 2*hi-1 as the reference computes them), a float sample is
// mapped as aligners.py:55-57 does.  `dt` is wave-uniform (one branch per call site, no divergence).
FFS_DEV double sample_any(const void* p, int dt, int i, double v0, double v1) {
    if (dt == 2) return get_bit(p, i) ? v1 : v0;
    if (dt == 0) return reinterpret_cast<const unsigned char*>(p)[i] ? v1 : v0;
    if (dt == 3) return 2.0 * reinterpret_cast<const double*>(p)[i] - 1.0;
    return 2.0 * (double)reinterpret_cast<const float*>(p)[i] - 1.0;
}
// sum over i in [a, b) of s'[i] * r'[i + d] for this thread's share (every nt-th sample), any pairing of types
FFS_DEV double mixed_dot(const CandDesc& cd, int d, int a, int b, int tid, int nt) {
    const int dts = (cd.flags >> CAND_DTS_SHIFT) & 7, dtr = (cd.flags >> CAND_DTR_SHIFT) & 7;
    double sum = 0.0;
    if (dts == 2 && dtr == 3) {
        // the common pairing: bit-packed candidate, float64 reference.  sum s'[i] r'[i+d] = s0 * sum r' + (s1 - s0) *
        // sum over set bits of r': two fp64 accumulators, one 8-byte load per sample, the bit word shared by 32 lanes
        // (two samples per 16-byte load, four loads in flight per lane: the loop is latency-bound otherwise)
        const unsigned* __restrict__ sw = reinterpret_cast<const unsigned*>(cd.s);
        const double* __restrict__ r = reinterpret_cast<const double*>(cd.r) + d;
        double all = 0.0, set = 0.0, all1 = 0.0, set1 = 0.0;
        int i = a + 2 * tid;
#pragma unroll 4
        for (; i + 1 < b; i += 2 * nt) {
            double2 x;
            __builtin_memcpy(&x, r + i, 16);
            const double x0 = 2.0 * x.x - 1.0, x1 = 2.0 * x.y - 1.0;
            all += x0;
            all1 += x1;
            set += ((sw[i >> 5] >> (i & 31)) & 1u) ? x0 : 0.0;
            set1 += ((sw[(i + 1) >> 5] >> ((i + 1) & 31)) & 1u) ? x1 : 0.0;
        }
        if (i < b) {  // the odd sample at the end of the range
            const double x0 = 2.0 * r[i] - 1.0;
            all += x0;
            set += ((sw[i >> 5] >> (i & 31)) & 1u) ? x0 : 0.0;
        }
        all += all1;
        set += set1;
        return cd.s0 * all + (cd.s1 - cd.s0) * set;
    }
    for (int i = a + tid; i < b; i += nt) sum += sample_any(cd.s, dts, i, cd.s0, cd.s1) * sample_any(cd.r, dtr, i + d, cd.r0, cd.r1);
    return sum;
}

template <int DT>
__global__ __launch_bounds__(256) void k_rescore(const CandDesc* __restrict__ cands, const NomList* __restrict__ noms,
                                                 RescoreAcc* __restrict__ acc, int first_cand) {
    const int ci = first_cand + blockIdx.y;
    const NomList& nl = noms[ci];
    const int count = nl.count;
    if (count <= 0) return;
    const CandDesc& cd = cands[ci];
    constexpr int VEC = (DT == 0) ? 16 : (DT == 2 ? 128 : (DT == 3 ? 2 : 4));  // (DT 4, mixed types: 4)  // elements per 16-byte load
    for (int ni = 0; ni < count; ++ni) {
        const int d = nl.d[ni];
        const int i0 = d < 0 ? -d : 0;
        const int i1 = (cd.R - d) < cd.S ? (cd.R - d) : cd.S;
        if (i1 <= i0) continue;
        int seg = (i1 - i0 + (int)gridDim.x - 1) / (int)gridDim.x;
        seg = (seg + VEC - 1) / VEC * VEC;
        const int a = i0 + (int)blockIdx.x * seg;
        const int b = (a + seg) < i1 ? (a + seg) : i1;
        if (a >= b) continue;
        RescoreAcc& out = acc[(size_t)ci * KNOM + ni];
        if (DT == 2) {
            unsigned int n11 = 0, n1x = 0, nx1 = 0;
            bit_counts(cd.s, cd.r, cd.R, d, a, b, (int)threadIdx.x, 256, n11, n1x, nx1);
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) {
                n11 += __shfl_xor(n11, sft, 64);
                n1x += __shfl_xor(n1x, sft, 64);
                nx1 += __shfl_xor(nx1, sft, 64);
            }
            if ((threadIdx.x & 63) == 0 && (n1x | nx1)) {
                atomicAdd(&out.n11, n11);
                atomicAdd(&out.n1x, n1x);
                atomicAdd(&out.nx1, nx1);
            }
        } else if (DT == 0) {
            const unsigned char* s = reinterpret_cast<const unsigned char*>(cd.s);
            const unsigned char* r = reinterpret_cast<const unsigned char*>(cd.r) + d;
            unsigned int n11 = 0, n1x = 0, nx1 = 0;
            for (int i = a + VEC * (int)threadIdx.x; i < b; i += VEC * 256) {
                if (i + VEC <= b) {
                    uint4 sv, rv;
                    __builtin_memcpy(&sv, s + i, 16);
                    __builtin_memcpy(&rv, r + i, 16);
                    const unsigned sw[4] = {sv.x, sv.y, sv.z, sv.w}, rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned fs = nz_flags(sw[k]), fr = nz_flags(rw[k]);
                        n11 += __popc(fs & fr);
                        n1x += __popc(fs);
                        nx1 += __popc(fr);
                    }
                } else {
                    for (int k = i; k < b; ++k) {
                        const unsigned sb = s[k] != 0, rb = r[k] != 0;
                        n11 += sb & rb;
                        n1x += sb;
                        nx1 += rb;
                    }
                }
            }
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) {
                n11 += __shfl_xor(n11, sft, 64);
                n1x += __shfl_xor(n1x, sft, 64);
                nx1 += __shfl_xor(nx1, sft, 64);
            }
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(&out.n11, n11);
                atomicAdd(&out.n1x, n1x);
                atomicAdd(&out.nx1, nx1);
            }
        } else if (DT == 3 || DT == 4) {
            double sum = 0.0;
            if (DT == 4)
                sum = mixed_dot(cd, d, a, b, (int)threadIdx.x, 256);
            else
                for (int i = a + (int)threadIdx.x; i < b; i += 256) sum += mapped_sample<3>(cd.s, i) * mapped_sample<3>(cd.r, i + d);
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) sum += __shfl_xor(sum, sft, 64);
            __shared__ double s_part3[4];
            __syncthreads();
            if ((threadIdx.x & 63) == 0) s_part3[threadIdx.x >> 6] = sum;
            __syncthreads();
            if (threadIdx.x == 0) out.part[blockIdx.x] = ((s_part3[0] + s_part3[1]) + s_part3[2]) + s_part3[3];
        } else {
            const float* s = reinterpret_cast<const float*>(cd.s);
            const float* r = reinterpret_cast<const float*>(cd.r) + d;
            double sum = 0.0;
            for (int i = a + VEC * (int)threadIdx.x; i < b; i += VEC * 256) {
                if (i + VEC <= b) {
                    float4 sv, rv;
                    __builtin_memcpy(&sv, s + i, 16);
                    __builtin_memcpy(&rv, r + i, 16);
                    sum += (2.0 * (double)sv.x - 1.0) * (2.0 * (double)rv.x - 1.0);
                    sum += (2.0 * (double)sv.y - 1.0) * (2.0 * (double)rv.y - 1.0);
                    sum += (2.0 * (double)sv.z - 1.0) * (2.0 * (double)rv.z - 1.0);
                    sum += (2.0 * (double)sv.w - 1.0) * (2.0 * (double)rv.w - 1.0);
                } else {
                    for (int k = i; k < b; ++k) sum += (2.0 * (double)s[k] - 1.0) * (2.0 * (double)r[k] - 1.0);
                }
            }
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) sum += __shfl_xor(sum, sft, 64);
            // fixed-order combine of the four waves: the result does not depend on scheduling
            __shared__ double s_part[4];
            __syncthreads();
            if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
            __syncthreads();
            if (threadIdx.x == 0) out.part[blockIdx.x] = ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3];
        }
    }
}

// Exact correlation of the mapped two-level vectors at lag d from the three counts (n11 = samples where both are at
// their upper level, n1x / nx1 = the candidate's / the reference's upper-level samples inside the overlap).  ONE fixed
// sequence of fp64 operations (explicit fused multiply-adds): the transform path's nominees and the run-boundary path's
// every lag are scored by exactly this arithmetic, so the two paths agree bit for bit.
FFS_DEV double two_level_score(const CandDesc& cd, int n11, int n1x, int nx1, int d) {
    const int i0 = d < 0 ? -d : 0;
    const int i1 = (cd.R - d) < cd.S ? (cd.R - d) : cd.S;
    const int ov = i1 > i0 ? (i1 - i0) : 0;
    const int n10 = n1x - n11, n01 = nx1 - n11;
    const int n00 = ov - n11 - n10 - n01;
    double r = (double)n00 * (cd.s0 * cd.r0);
    r = __builtin_fma((double)n01, cd.s0 * cd.r1, r);
    r = __builtin_fma((double)n10, cd.s1 * cd.r0, r);
    return __builtin_fma((double)n11, cd.s1 * cd.r1, r);
}

FFS_DEV double exact_score(const CandDesc& cd, const RescoreAcc& a, int d, int dt) {
    if (dt == 1 || dt == 3 || dt == 4) {
        double sum = 0.0;
        for (int i = 0; i < RSEG; ++i) sum += a.part[i];
        return sum;
    }
    return two_level_score(cd, (int)a.n11, (int)a.n1x, (int)a.nx1, d);
}

// Pool re-evaluation: one block per entry (grid-stride), the whole overlap in one block.
template <int DT>
__global__ __launch_bounds__(256) void k_pool_rescore(const CandDesc* __restrict__ cands, const PoolHeader* __restrict__ pool,
                                                      PoolEntry* __restrict__ entries, PoolBest* __restrict__ best) {
    const unsigned n = pool->count < pool->capacity ? pool->count : pool->capacity;
    __shared__ unsigned int s_cnt[3][4];
    __shared__ double s_sum[4];
    for (unsigned e = blockIdx.x; e < n; e += gridDim.x) {
        const int ci = entries[e].ci, d = entries[e].d;
        const CandDesc& cd = cands[ci];
        const int i0 = d < 0 ? -d : 0;
        const int i1 = (cd.R - d) < cd.S ? (cd.R - d) : cd.S;
        double score = 0.0;
        if (DT != 1 && DT != 3 && DT != 4) {
            const unsigned char* s = reinterpret_cast<const unsigned char*>(cd.s);
            const unsigned char* r = reinterpret_cast<const unsigned char*>(cd.r) + (DT == 0 ? d : 0);
            unsigned int n11 = 0, n1x = 0, nx1 = 0;
            if (DT == 2) bit_counts(cd.s, cd.r, cd.R, d, i0, i1, (int)threadIdx.x, 256, n11, n1x, nx1);
            for (int i = i0 + 16 * (int)threadIdx.x; DT == 0 && i < i1; i += 16 * 256) {
                if (i + 16 <= i1) {
                    uint4 sv, rv;
                    __builtin_memcpy(&sv, s + i, 16);
                    __builtin_memcpy(&rv, r + i, 16);
                    const unsigned sw[4] = {sv.x, sv.y, sv.z, sv.w}, rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned fs = nz_flags(sw[k]), fr = nz_flags(rw[k]);
                        n11 += __popc(fs & fr);
                        n1x += __popc(fs);
                        nx1 += __popc(fr);
                    }
                } else {
                    for (int k = i; k < i1; ++k) {
                        const unsigned sb = s[k] != 0, rb = r[k] != 0;
                        n11 += sb & rb;
                        n1x += sb;
                        nx1 += rb;
                    }
                }
            }
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) {
                n11 += __shfl_xor(n11, sft, 64);
                n1x += __shfl_xor(n1x, sft, 64);
                nx1 += __shfl_xor(nx1, sft, 64);
            }
            __syncthreads();
            if ((threadIdx.x & 63) == 0) {
                s_cnt[0][threadIdx.x >> 6] = n11;
                s_cnt[1][threadIdx.x >> 6] = n1x;
                s_cnt[2][threadIdx.x >> 6] = nx1;
            }
            __syncthreads();
            RescoreAcc a;
            a.n11 = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
            a.n1x = s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3];
            a.nx1 = s_cnt[2][0] + s_cnt[2][1] + s_cnt[2][2] + s_cnt[2][3];
            score = exact_score(cd, a, d, 0);
        } else {
            double sum = 0.0;
            if (DT == 4)
                sum = mixed_dot(cd, d, i0, i1, (int)threadIdx.x, 256);
            else
                for (int i = i0 + (int)threadIdx.x; i < i1; i += 256)
                    sum += mapped_sample<DT>(cd.s, i) * mapped_sample<DT>(cd.r, i + d);
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) sum += __shfl_xor(sum, sft, 64);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = sum;
            __syncthreads();
            score = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
        }
        if (threadIdx.x == 0) {
            entries[e].score = score;
            atomicMax(&best[ci].key, score_key(score));
        }
    }
}

// second phase: among the entries attaining a candidate's best exact score keep the largest lag
__global__ void k_pool_pick(const PoolHeader* __restrict__ pool, const PoolEntry* __restrict__ entries,
                            PoolBest* __restrict__ best) {
    const unsigned n = pool->count < pool->capacity ? pool->count : pool->capacity;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const PoolEntry& en = entries[e];
        if (score_key(en.score) == best[en.ci].key) atomicMax(&best[en.ci].d, en.d + POOL_D_BIAS);
    }
}

// The window's lags with an empty overlap (exact value 0, see CAND_HAS_ZERO) against the best real lag:
// larger score wins, equal scores go to the larger lag (np.argmax's first k).
FFS_DEV void apply_zero_rule(const CandDesc& cd, CandResult& r) {
    if (!(cd.flags & CAND_HAS_ZERO)) return;
    if ((r.flags & 1) || 0.0 > r.score || (0.0 == r.score && (long long)cd.d_zero > r.offset)) {
        r.score = 0.0;
        r.offset = cd.d_zero;
        r.score_f32 = 0.0f;
        r.flags &= ~(1 | 2);
    }
}

// one thread per candidate: best nominee by exact score (ties -> largest d = first k)
__global__ void k_finalize_cands(const CandDesc* __restrict__ cands, const NomList* __restrict__ noms,
                                 const RescoreAcc* __restrict__ acc, CandResult* __restrict__ out, int n, int dt,
                                 const PoolHeader* __restrict__ pool, const PoolBest* __restrict__ pbest) {
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= n) return;
    const CandDesc& cd = cands[ci];
    const NomList& nl = noms[ci];
    CandResult r;
    if ((cd.flags & CAND_NO_LAGS) || nl.count == 0) {
        // every lag masked: np.argmax of all -inf is k=0 (aligners.py:45-48)
        r.score = -INFINITY;
        r.offset = (long long)cd.n_ref - 1 - cd.S;
        r.score_f32 = -INFINITY;
        r.flags = 1;
    } else {
        double bs = -INFINITY;
        int bd = INT32_MIN;
        float bf = 0.f;
        for (int i = 0; i < nl.count; ++i) {
            const double sc = exact_score(cd, acc[(size_t)ci * KNOM + i], nl.d[i], dt);
            if (sc > bs || (sc == bs && nl.d[i] > bd)) {
                bs = sc;
                bd = nl.d[i];
                bf = nl.val[i];
            }
        }
        r.score = bs;
        r.offset = bd;
        r.score_f32 = bf;
        r.flags = nl.flags & 2;
        // nominee lists overflowed: the exhaustive pool holds every lag within the margin, unless the
        // pool itself overflowed (then the best-of-list answer above stands, flagged ambiguous)
        if ((nl.flags & 2) && !pbest[ci].overflow && pbest[ci].key != 0) {
            r.score = key_score(pbest[ci].key);
            r.offset = pbest[ci].d - POOL_D_BIAS;
            r.flags = 0;
        }
    }
    apply_zero_rule(cd, r);
    out[ci] = r;
}

// one thread per pair: MaxScoreAligner.transform (aligners.py:154-167)
__global__ void k_finalize_pairs(CandResult* __restrict__ cres, PairResult* __restrict__ out, int n_pairs, int n_cand,
                                 long long filter_max) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    PairResult r;
    r.score = 0;
    r.offset = 0;
    r.best_cand = -1;
    r.flags = 0;
    for (int j = 0; j < n_cand; ++j) {
        CandResult& c = cres[(size_t)p * n_cand + j];
        const long long ao = c.offset < 0 ? -c.offset : c.offset;
        if (filter_max >= 0 && ao > filter_max) {
            c.flags |= 4;
            continue;
        }
        if (r.best_cand < 0 || c.score > r.score) {
            r.score = c.score;
            r.offset = c.offset;
            r.best_cand = j;
            r.flags = c.flags;
        }
    }
    out[p] = r;
}

// --------------------------------------------------------------------------------------------
// direct exact correlation for short inputs: one block per candidate, every lag of the window
// evaluated exactly (integer counts / fp64 in index order), argmax with ties -> largest d.
template <int DT>
__global__ __launch_bounds__(256) void k_direct(const CandDesc* __restrict__ cands, CandResult* __restrict__ out) {
    const int ci = blockIdx.x;
    const CandDesc cd = cands[ci];
    __shared__ double s_sc[256];
    __shared__ int s_d[256];
    double bs = -INFINITY;
    int bd = INT32_MIN;
    if (!(cd.flags & CAND_NO_LAGS)) {
        for (int d = cd.d_lo + (int)threadIdx.x; d <= cd.d_hi; d += 256) {
            const int i0 = d < 0 ? -d : 0;
            const int i1 = (cd.R - d) < cd.S ? (cd.R - d) : cd.S;
            double sc;
            if (DT == 4) {
                sc = mixed_dot(cd, d, i0, i1, 0, 1);
            } else if (DT != 1 && DT != 3) {
                const unsigned char* s = reinterpret_cast<const unsigned char*>(cd.s);
                const unsigned char* r = reinterpret_cast<const unsigned char*>(cd.r);
                RescoreAcc a;
                a.n11 = a.n1x = a.nx1 = 0;
                for (int i = i0; i < i1; ++i) {
                    const unsigned int sb = (DT == 2) ? get_bit(s, i) : (unsigned)(s[i] != 0);
                    const unsigned int rb = (DT == 2) ? get_bit(r, i + d) : (unsigned)(r[i + d] != 0);
                    a.n11 += sb & rb;
                    a.n1x += sb;
                    a.nx1 += rb;
                }
                sc = exact_score(cd, a, d, 0);
            } else {
                sc = 0.0;
                for (int i = i0; i < i1; ++i) sc += mapped_sample<DT>(cd.s, i) * mapped_sample<DT>(cd.r, i + d);
            }
            if (sc > bs || (sc == bs && d > bd)) {
                bs = sc;
                bd = d;
            }
        }
    }
    s_sc[threadIdx.x] = bs;
    s_d[threadIdx.x] = bd;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) {
            const double os = s_sc[threadIdx.x + st];
            const int od = s_d[threadIdx.x + st];
            if (os > s_sc[threadIdx.x] || (os == s_sc[threadIdx.x] && od > s_d[threadIdx.x])) {
                s_sc[threadIdx.x] = os;
                s_d[threadIdx.x] = od;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        CandResult r;
        if (s_d[0] == INT32_MIN) {
            r.score = -INFINITY;
            r.offset = (long long)cd.n_ref - 1 - cd.S;
            r.score_f32 = -INFINITY;
            r.flags = 1 | 8;
        } else {
            r.score = s_sc[0];
            r.offset = s_d[0];
            r.score_f32 = (float)s_sc[0];
            r.flags = 8;
        }
        apply_zero_rule(cd, r);
        out[ci] = r;
    }
}

// --------------------------------------------------------------------------------------------
// VAD frame-energy sweep (speech_transformers.py:133-150: one Python call per 10 ms frame).
// speech  <=>  sum(x^2) >= thr_lin * n   (== 10*log10(mean x^2) >= thr_db, evaluated exactly in integers / fp64),
// n = samples in the frame (the last frame may be short).
//
// A wave owns VAD_FPT = 8 consecutive frames -- one BYTE of the bit-packed label vector -- and walks them four at a
// time: lane l < frame_len/8 loads its 16-byte vector of each of the four frames (four independent nontemporal loads in
// flight per lane; the sweep reads every byte once), squares with v_dot2_i32_i16 (two samples per instruction; a pair
// of squares is at most 2^31 and is taken as an unsigned value) and adds in 64 bits.  The wave sum is split into a
// 20-bit and a 14-bit half so that both reduce in 32-bit DPP adds (quad / half-mirror / mirror: every lane of a row
// ends up with the row total) and the four row totals are combined on the scalar unit.  Round 2's kernel (one frame
// per wave iteration, v_mad_u64_u32 chains: ~40 VALU instructions per 16 bytes) was VALU-bound at 4.6-5.3 TB/s; a
// read-only sweep reaches 6.2-6.9 (profiles/read_ceiling.hip).
// Outputs (either may be null): labels[f] = 1.0f / non_speech (fp32), bits[f >> 3] bit (f & 7) = speech.
#define VAD_FPT 8
FFS_DEV unsigned row_total(unsigned v) {  // sum over the 16 lanes of a DPP row, in every lane of the row
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror
    return v;
}
FFS_DEV unsigned long long wave_total(unsigned long long part) {  // part < 2^34 per lane; wave-uniform result
    const unsigned lo = row_total((unsigned)part & 0xFFFFFu), hi = row_total((unsigned)(part >> 20));
    unsigned long long t = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        t += (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)lo, 16 * r) +
             ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, 16 * r) << 20);
    return t;
}
typedef int vad_v4i __attribute__((ext_vector_type(4)));
// lo^2 + hi^2 of the two int16 halves of w (at most 2^31: the bit pattern is the unsigned sum).  Inline asm: the
// __builtin_amdgcn_sdot2 + bit_cast form of this loop is folded wrongly by this compiler (all four words become word 0).
FFS_DEV unsigned squares2(int w) {
    int d = 0;
    asm("v_dot2c_i32_i16 %0, %1, %1" : "+v"(d) : "v"(w));
    return (unsigned)d;
}
FFS_DEV unsigned long long squares8(vad_v4i w) {  // exact sum of the eight squared int16 samples of a 16-byte vector
    unsigned long long a = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) a += squares2(w[k]);
    return a;
}
__global__ __launch_bounds__(256) void k_vad_energy(const int16_t* __restrict__ pcm, long long n_samples, int frame_len,
                                                    long long n_frames, double thr_lin, float non_speech,
                                                    float* __restrict__ labels, unsigned char* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x / 64);
    const long long nwaves = (long long)gridDim.x * (blockDim.x / 64);
    const long long n_tasks = (n_frames + VAD_FPT - 1) / VAD_FPT;
    const int nvec = frame_len / 8;
    const bool vec_ok = (frame_len % 8 == 0) && nvec <= 64 && ((reinterpret_cast<uintptr_t>(pcm) & 15) == 0);
    const double thr_full = thr_lin * (double)frame_len;
    for (long long t = wave0; t < n_tasks; t += nwaves) {
        const long long f0 = t * VAD_FPT;
        unsigned word = 0;
#pragma unroll
        for (int g = 0; g < VAD_FPT; g += 4) {
            const long long fg = f0 + g;
            if (fg >= n_frames) break;
            if (vec_ok && (fg + 4) * frame_len <= n_samples) {  // four whole frames: wave-uniform fast path
                const vad_v4i* p = reinterpret_cast<const vad_v4i*>(pcm + fg * frame_len);
                vad_v4i w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    w[i] = lane < nvec ? __builtin_nontemporal_load(p + i * nvec + lane) : vad_v4i{0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned long long e = wave_total(squares8(w[i]));
                    word |= ((double)e >= thr_full ? 1u : 0u) << (g + i);
                }
            } else {  // short tail frame, unaligned buffer, odd frame length: one frame at a time, element loop
                for (int i = 0; i < 4 && fg + i < n_frames; ++i) {
                    const long long s0 = (fg + i) * frame_len;
                    const long long s1 = (s0 + frame_len) < n_samples ? (s0 + frame_len) : n_samples;
                    const int n = (int)(s1 - s0);
                    unsigned long long e = 0;
                    for (int base = 0; base < n; base += 512) {  // <= 8 samples per lane per round: part < 2^34
                        unsigned long long part = 0;
                        for (int j = base + lane; j < n && j < base + 512; j += 64) {
                            const int x = pcm[s0 + j];
                            part += (unsigned)(x * x);
                        }
                        e += wave_total(part);
                    }
                    word |= ((double)e >= thr_lin * (double)n ? 1u : 0u) << (g + i);
                }
            }
        }
        if (bits && lane == 0) bits[t] = (unsigned char)word;
        if (labels && lane < VAD_FPT && f0 + lane < n_frames) labels[f0 + lane] = ((word >> lane) & 1u) ? 1.0f : non_speech;
    }
}

// auditok-style token smoothing: the tokenizer is a sequential state machine per chunk, so one thread
// walks each chunk (chunks are independent; a 2 h file has 72 of them -- negligible work).
__global__ void k_vad_tokenize(const float* __restrict__ valid, long long n_frames, long long chunk, int min_len,
                               int max_len, int max_sil, float non_speech, float* __restrict__ out) {
    const long long ci = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long f0 = ci * chunk;
    if (f0 >= n_frames) return;
    const long long n = (f0 + chunk) < n_frames ? chunk : (n_frames - f0);
    const float* v = valid + f0;
    float* o = out + f0;
    // markers live in the output buffer itself (slot n, the one past the end, is tracked separately)
    for (long long i = 0; i < n; ++i) o[i] = 0.0f;
    enum { SILENCE = 0, POSSIBLE_SILENCE = 1, NOISE = 3 };
    int state = SILENCE, silence_len = 0;
    long long start = 0, len = 0;  // current token: frames [start, start + len)
    bool contiguous = false;
    auto deliver = [&](bool truncated, long long cur) {
        // _process_end_of_detection (default mode: trailing silence kept, min length not strict)
        if (len >= min_len || (len > 0 && contiguous)) {
            const long long end = start + len - 1;
            o[start] = 1.0f;
            if (end + 1 < n) o[end + 1] = non_speech - 1.0f;
            if (truncated) {
                start = cur + 1;
                contiguous = true;
            } else {
                contiguous = false;
            }
        } else {
            contiguous = false;
        }
        len = 0;
    };
    for (long long i = 0; i < n; ++i) {
        const bool ok = v[i] != 0.0f;
        if (state == SILENCE) {
            if (ok) {
                silence_len = 0;
                start = i;
                len = 1;
                state = NOISE;  // init_min = 0
                if (len >= max_len) deliver(true, i);
            }
        } else if (state == NOISE) {
            if (ok) {
                ++len;
                if (len >= max_len) deliver(true, i);
            } else if (max_sil <= 0) {
                deliver(false, i);
                state = SILENCE;
            } else {
                silence_len = 1;
                ++len;
                state = POSSIBLE_SILENCE;
                if (len == max_len) deliver(true, i);
            }
        } else {  // POSSIBLE_SILENCE
            if (ok) {
                ++len;
                silence_len = 0;
                state = NOISE;
                if (len >= max_len) deliver(true, i);
            } else if (silence_len >= max_sil) {
                if (silence_len < len)
                    deliver(false, i);
                else
                    len = 0;
                state = SILENCE;
                silence_len = 0;
            } else {
                ++len;
                ++silence_len;
                if (len >= max_len) deliver(true, i);
            }
        }
    }
    // _post_process: flush a token still open at the end of the chunk
    if ((state == NOISE || state == POSSIBLE_SILENCE) && len > 0 && len > silence_len) deliver(false, n - 1);
    // clip(cumsum(markers)[:-1], 0, 1)
    float acc = 0.0f;
    for (long long i = 0; i < n; ++i) {
        acc += o[i];
        o[i] = fminf(fmaxf(acc, 0.0f), 1.0f);
    }
}

// The same smoothing as a parallel kernel: one workgroup per chunk.  What makes that possible: the state of the
// tokenizer (SILENCE / NOISE / POSSIBLE_SILENCE and its silence counter) depends on the validity runs alone, never on
// token lengths -- truncation at max_len only cuts the frames of an "island" into pieces:
//   * an island starts at a valid frame that follows more than max_sil invalid ones (or none valid at all) and runs
//     until max_sil frames past its last valid frame (to the end of the chunk if no longer gap follows);
//   * inside an island a token is cut every max_len frames; full pieces are always delivered (max_len >= min_len, the
//     host falls back to k_vad_tokenize otherwise), the remainder of r frames iff max_sil < r (gap) or r > trailing
//     silence (end of chunk) and (r >= min_len or "contiguous": it follows a cut, or -- first piece -- the previous
//     island ended with a cut followed by at most max_sil frames, which leaves the tokenizer's flag set);
//   * markers: +1 at the first frame of a delivered piece, non_speech - 1 behind its last (the +1 wins where both
//     fall on one frame, as the reference's in-order assignments do), then clip(cumsum, 0, 1).
// Round 6 (second form): no per-frame index arrays.  Rounds 4-6 kept three of them (last valid frame, island start,
// island end) and filled them with per-thread serial passes over ~11 frames each plus block scans: 54 workgroups on
// 256 CUs, four waves per SIMD, every pass a chain of dependent LDS round trips -- 22.7 us per 90-minute file, bound by
// the latency of ONE wave's instruction stream (profiles/r06_runs_experiments.json).  Now:
//   P0  validity as 64-bit words V (one ballot per 64 coalesced frames);
//   P1  PL[w] = last valid frame in the words in front of w (one value per WORD: a DPP wave scan + seven wave totals),
//       which makes "last valid frame <= i" one or two LDS reads: the word's own bits, else PL;
//   P2  island starts as a second bit array S, one thread per word: frame i starts an island iff it is valid and no valid
//       frame lies in the max_sil + 1 frames in front of it -- the word's own bits smeared upwards (shift-or doubling)
//       and the low frames that PL still covers;
//   P3  SL[w] / NS[w] = last start in front of / first start behind word w (two scans, one pair of barriers);
//   P4  one thread per HALF WORD of S walks its island starts (typically none or one) and sets the marker BITS of the
//       island's pieces in two more bit arrays (P: +1, M: non_speech - 1; a frame in both carries the +1); the island's
//       end is min(n, last valid frame in front of the next start + max_sil + 1);
//   P5  clip(cumsum) per frame without a sum over frames: label(i) = clamp(#P(<= i) + #M(<= i) (non_speech - 1), 0, 1),
//       the counts from a packed prefix count per word + a popcount (fp64; equal to the sequential sum whenever that one
//       is exact -- every term is a float -- which is what rounds 4-6's blocked prefix sum relied on as well).
// Frames map to lanes the same way in P0 and P5: wave v owns the words v, v + 16, ...; lane k of the wave holds the
// wave's k-th word of every bit array in a register (v_readlane hands it to the whole wave: no LDS round trip per word),
// lane l of the wave is frame 64 w + l of the word being processed -- loads and stores are coalesced.
// oracle/vad_oracle.py::tokenize_chunk_words is the model of exactly this, tested against the state machine on the
// CPU.  LDS: 48 bytes per 64-frame word, nothing per frame.
// (FFS_TOK_STOP=k, never defined in the product build: section stop points for profiles/tok_sections.sh -- WRONG results.)
constexpr int TOK_SCAN_MAX = 28672;
constexpr int TOK_THREADS = 1024;
constexpr int TOK_WORDS = TOK_SCAN_MAX / 64;
constexpr int TOK_WPW = TOK_WORDS / (TOK_THREADS / 64);  // words per wave
constexpr int TOK_WORD_WAVES = TOK_SCAN_MAX / 64 / 64;  // waves whose threads own a word in the word scans
static_assert(TOK_SCAN_MAX % 1024 == 0 && TOK_SCAN_MAX / 32 <= TOK_THREADS && TOK_WPW <= 64 && TOK_SCAN_MAX < 65536, "one thread per half word, one lane per word of a wave, 16-bit counts");

// inclusive scan over the 64 lanes, DPP only; op(earlier, later), identity = left identity of op
template <class Op>
FFS_DEV int tok_wave_incl_scan(int v, int identity, Op op) {
#define FFS_TOK_STEP(ctrl, rows) v = op(__builtin_amdgcn_update_dpp(identity, v, ctrl, rows, 0xf, false), v)
    FFS_TOK_STEP(0x111, 0xf);  // row_shr:1 (a lane without a source receives the identity)
    FFS_TOK_STEP(0x112, 0xf);
    FFS_TOK_STEP(0x114, 0xf);
    FFS_TOK_STEP(0x118, 0xf);
    FFS_TOK_STEP(0x142, 0xa);  // row_bcast:15
    FFS_TOK_STEP(0x143, 0xc);  // row_bcast:31
#undef FFS_TOK_STEP
    return v;
}
// block-wide INCLUSIVE scan of one value per thread of the first TOK_WORD_WAVES waves (the word owners); s_w: one int per
// wave; the caller has a barrier between two uses of the same s_w
template <class Op>
FFS_DEV int tok_words_incl_scan(int v, int identity, Op op, int* s_w, int lane, int wave) {
    const int incl = tok_wave_incl_scan(v, identity, op);
    if (lane == 63 && wave < TOK_WORD_WAVES) s_w[wave] = incl;
    __syncthreads();
    int sw[TOK_WORD_WAVES];
#pragma unroll
    for (int w = 0; w < TOK_WORD_WAVES; ++w) sw[w] = s_w[w];
    int pre = identity;
#pragma unroll
    for (int w = 0; w < TOK_WORD_WAVES; ++w) pre = w < wave ? op(pre, sw[w]) : pre;
    return op(pre, incl);
}

struct TokWords {
    const unsigned long long* V;  // validity, bit i & 63 of word i >> 6
    const unsigned long long* S;  // island starts
    const int* PL;                // last valid frame in the words in front of w (-1: none)
    const int* SL;                // last island start in the words in front of w
    const int* NS;                // first island start in the words behind w (-1: none)
    int n, min_len, max_len, max_sil, ms;
    FFS_DEV static int last_le(const unsigned long long* B, const int* front, int i) {
        if (i < 0) return -1;
        const int w = i >> 6;
        const unsigned long long m = B[w] & (~0ull >> (63 - (i & 63)));
        return m ? (w << 6) + 63 - __clzll((long long)m) : front[w];
    }
    FFS_DEV int lastv(int i) const { return last_le(V, PL, i); }
    FFS_DEV int island_end(int i) const {  // i inside an island: its last frame
        const int w = i >> 6, b = i & 63;
        const unsigned long long m = b == 63 ? 0ull : S[w] & (~0ull << (b + 1));
        const int ns = m ? (w << 6) + __ffsll((long long)m) - 1 : NS[w];
        const int lv = lastv((ns >= 0 ? ns : n) - 1);
        return (lv + ms + 1 < n ? lv + ms + 1 : n) - 1;
    }
    FFS_DEV bool c_in(int s) const {
        if (max_sil <= 0 || s == 0) return false;
        const int lp = lastv(s - 1);
        if (lp < 0) return false;
        const int lenp = lp + max_sil - last_le(S, SL, lp) + 1;
        const int qd = lenp / max_len;
        return qd >= 1 && lenp - qd * max_len <= max_sil;
    }
    // the markers of the island that starts at frame s
    FFS_DEV void island(int s, unsigned long long* P, unsigned long long* M) const {
        const int e_isl = island_end(s);
        const int t_end = e_isl + 1 < n ? 0 : e_isl - lastv(e_isl);  // trailing silence of an island cut by the chunk's end
        int cin = -1;                                                // c_in(s), when a first piece needs it
        int j = 0;
        for (int i0 = s; i0 <= e_isl; i0 += max_len, ++j) {
            const int e = (i0 + max_len - 1) < e_isl ? (i0 + max_len - 1) : e_isl;
            const int r = e - i0 + 1;
            bool d = true;
            if (r != max_len) {
                bool ok_len = r >= min_len;
                if (!ok_len && r > 0) {
                    if (j >= 1) {
                        ok_len = true;
                    } else {
                        if (cin < 0) cin = c_in(s) ? 1 : 0;
                        ok_len = cin != 0;
                    }
                }
                if (e_isl + 1 < n)
                    d = max_sil <= 0 ? ok_len : (max_sil < r && ok_len);  // ended by a long gap
                else
                    d = r > 0 && r > t_end && ok_len;
            }
            if (d) {
                if (e + 1 < n) atomicOr(&M[(e + 1) >> 6], 1ull << ((e + 1) & 63));
                atomicOr(&P[i0 >> 6], 1ull << (i0 & 63));  // (a frame in both arrays carries the +1)
            }
        }
    }
};

__global__ __launch_bounds__(TOK_THREADS) void k_vad_tokenize_scan(const float* __restrict__ valid, long long n_frames, long long chunk,
                                                                  int min_len, int max_len, int max_sil, float non_speech,
                                                                  float* __restrict__ out) {
    __shared__ unsigned long long s_V[TOK_WORDS], s_S[TOK_WORDS], s_P[TOK_WORDS], s_M[TOK_WORDS];
    __shared__ int s_PL[TOK_WORDS], s_SL[TOK_WORDS], s_NS[TOK_WORDS], s_PP[TOK_WORDS];
    __shared__ int s_w[3][TOK_WORD_WAVES];
    const long long f0 = (long long)blockIdx.x * chunk;
    if (f0 >= n_frames) return;
    const int n = (int)((f0 + chunk) < n_frames ? chunk : (n_frames - f0));
    const float* v = valid + f0;
    float* o = out + f0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = (n + 63) >> 6;
    const int ms = max_sil > 0 ? max_sil : 0;
    constexpr int NWV = TOK_THREADS / 64;
    const int my_w = wave + NWV * lane;  // the word this lane keeps in registers (lanes < TOK_WPW)
    const bool own = lane < TOK_WPW && my_w < W;
    auto lane64 = [](unsigned long long x, int k) -> unsigned long long {  // (k uniform)
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, k), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), k);
        return ((unsigned long long)hi << 32) | lo;
    };
    // P0: validity words -- every load of the wave requested before the first ballot waits for one
    unsigned long long v_reg = 0ull;
    {
        float x[TOK_WPW];
#pragma unroll
        for (int k = 0; k < TOK_WPW; ++k) {
            x[k] = 0.0f;
            if (wave + NWV * k < W) {
                const int i = ((wave + NWV * k) << 6) + lane;
                if (i < n) x[k] = v[i];
            }
        }
#pragma unroll
        for (int k = 0; k < TOK_WPW; ++k) {
            if (wave + NWV * k < W) {
                const unsigned long long m = __ballot(x[k] != 0.0f);
                if (lane == k) v_reg = m;
            }
        }
    }
    if (own) s_V[my_w] = v_reg;
    if (tid < W) s_P[tid] = 0ull, s_M[tid] = 0ull;
    __syncthreads();
#if defined(FFS_TOK_STOP) && FFS_TOK_STOP == 1
    if (n >= 0) return;
#endif
    auto imax = [](int x, int y) { return x > y ? x : y; };
    auto later_known = [](int earlier, int later) { return later >= 0 ? later : earlier; };
    // P1: last valid frame in front of every word
    {
        int a = -1;
        if (tid < W) {
            const unsigned long long m = s_V[tid];
            a = m ? (tid << 6) + 63 - __clzll((long long)m) : -1;
        }
        const int incl = tok_words_incl_scan(a, -1, imax, s_w[0], lane, wave);
        if (tid == 0) s_PL[0] = -1;
        if (tid + 1 < W) s_PL[tid + 1] = incl;
    }
    __syncthreads();
#if defined(FFS_TOK_STOP) && FFS_TOK_STOP == 2
    if (n >= 0) return;
#endif
    // P2: island starts, one thread per word: a valid frame starts an island iff no valid frame lies in the K = max_sil + 1
    // frames in front of it -- the word's own bits smeared upwards by 1 .. K (doubling: log2 K shift-ors), and the low
    // frames the last valid frame of the words in front still covers
    if (tid < W) {
        const unsigned long long x = s_V[tid];
        const int K = ms + 1;
        unsigned long long g;
        if (K >= 64) {
            const unsigned long long low = x & (0ull - x);
            g = x ? ~(low | (low - 1ull)) : 0ull;  // every frame of the word behind its first valid one
        } else {
            g = x << 1;
            for (int c = 1; c < K;) {
                const int sh = c < K - c ? c : K - c;
                g |= g << sh;
                c += sh;
            }
        }
        const int pl = s_PL[tid];
        int cnt = pl >= 0 ? pl + K - (tid << 6) + 1 : 0;  // frames 64 tid .. pl + K are within K of frame pl
        cnt = cnt < 0 ? 0 : (cnt > 64 ? 64 : cnt);
        const unsigned long long inc = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
        s_S[tid] = x & ~(g | inc);
    }
    __syncthreads();
#if defined(FFS_TOK_STOP) && FFS_TOK_STOP == 3
    if (n >= 0) return;
#endif
    // P3: last start in front of / first start behind every word (the second scan runs over the words in reverse order)
    {
        int a = -1, b = -1;
        if (tid < W) {
            const unsigned long long m = s_S[tid], mr = s_S[W - 1 - tid];
            a = m ? (tid << 6) + 63 - __clzll((long long)m) : -1;
            b = mr ? ((W - 1 - tid) << 6) + __ffsll((long long)mr) - 1 : -1;
        }
        const int ia = tok_wave_incl_scan(a, -1, imax), ib = tok_wave_incl_scan(b, -1, later_known);
        if (lane == 63 && wave < TOK_WORD_WAVES) s_w[1][wave] = ia, s_w[2][wave] = ib;
        __syncthreads();
        int pa = -1, pb = -1;
#pragma unroll
        for (int w = 0; w < TOK_WORD_WAVES; ++w) {
            const int ta = s_w[1][w], tb = s_w[2][w];
            pa = w < wave ? imax(pa, ta) : pa;
            pb = w < wave ? later_known(pb, tb) : pb;
        }
        if (tid == 0) s_SL[0] = -1, s_NS[W - 1] = -1;
        if (tid + 1 < W) {
            s_SL[tid + 1] = imax(pa, ia);
            s_NS[W - 2 - tid] = later_known(pb, ib);  // words behind W - 2 - tid = the words W - 1 - tid .. W - 1
        }
    }
    __syncthreads();
#if defined(FFS_TOK_STOP) && FFS_TOK_STOP == 4
    if (n >= 0) return;
#endif
    // P4: marker bits, island by island
    if (tid < 2 * W) {
        const TokWords tw{s_V, s_S, s_PL, s_SL, s_NS, n, min_len, max_len, max_sil, ms};
        const int w = tid >> 1, h = tid & 1;
        unsigned m = (unsigned)(s_S[w] >> (32 * h));
        while (m) {
            const int b = __ffs((int)m) - 1;
            m &= m - 1;
            tw.island((w << 6) + 32 * h + b, s_P, s_M);
        }
    }
    __syncthreads();
#if defined(FFS_TOK_STOP) && FFS_TOK_STOP == 5
    if (n >= 0) return;
#endif
    // P5: markers in front of every word (packed: +1 markers in the low half, end markers in the high half), labels
    {
        int c = 0;
        if (tid < W) {
            const unsigned long long pw = s_P[tid], mw = s_M[tid] & ~pw;
            c = __popcll(pw) | (__popcll(mw) << 16);
        }
        const int incl = tok_words_incl_scan(c, 0, [](int x, int y) { return x + y; }, s_w[0], lane, wave);
        if (tid < W) s_PP[tid] = incl - c;
    }
    __syncthreads();
    {
        unsigned long long p_reg = 0ull, m_reg = 0ull;
        int pp_reg = 0;
        if (own) p_reg = s_P[my_w], m_reg = s_M[my_w] & ~p_reg, pp_reg = s_PP[my_w];
        const double m_end = (double)(non_speech - 1.0f);
        const unsigned long long le = ~0ull >> (63 - lane);  // this frame and the frames of its word in front of it
        for (int k = 0; wave + NWV * k < W; ++k) {
            const int i = ((wave + NWV * k) << 6) + lane;
            const unsigned long long pw = lane64(p_reg, k), mw = lane64(m_reg, k);
            const int pp = __builtin_amdgcn_readlane(pp_reg, k);
            const int cp = (pp & 0xffff) + __popcll(pw & le), cm = (pp >> 16) + __popcll(mw & le);
            const double run = (double)cp + (double)cm * m_end;
            if (i < n) o[i] = (float)fmin(fmax(run, 0.0), 1.0);
        }
    }
}

// ---- subtitle rasteriser arithmetic, shared by the host entry points and the batched kernel --------------------
// (fp64 IEEE operations only -- division, multiplication, modf, round, rint -- so host and device agree bit for bit;
// contraction is off inside them: nothing here may turn into an fma)
// Python slice semantics of  x[:stop] = v  /  x[start:] = v  on a length-n array
FFS_HD long long slice_clamp(long long i, long long n) {
    if (i < 0) i += n;
    if (i < 0) i = 0;
    if (i > n) i = n;
    return i;
}
// datetime.timedelta(seconds=x).total_seconds() for a float x >= 0, i.e. x rounded to whole
// microseconds the way CPython's delta_new/accum does it (integer part exact, fractional part
// times 1e6 split again, leftover rounded half-to-even on the accumulated parity).
FFS_HD long long timedelta_us(double x) {
#pragma clang fp contract(off)
    double ip;
    const double fr = modf(x, &ip);
    long long us = (long long)ip * 1000000;
    if (fr != 0.0) {
        double ip2;
        const double fr2 = modf(1e6 * fr, &ip2);
        us += (long long)ip2;
        if (fr2 != 0.0) {
            double whole = round(fr2);
            if (fabs(whole - fr2) == 0.5) {
                const int is_odd = (int)(us & 1);
                whole = 2.0 * round((fr2 + is_odd) * 0.5) - is_odd;
            }
            us += (long long)whole;
        }
    }
    return us;
}
FFS_HD double scaled_seconds(long long us, double ratio) {
#pragma clang fp contract(off)
    const double t = (double)us / 1e6;                  // timedelta.total_seconds()
    return (double)timedelta_us(t * ratio) / 1e6;       // SubtitleScaler: timedelta(seconds=t*ratio)
}
// one subtitle's clamped [a, b) sample interval of a raster of out_len samples (speech_transformers.py:968-975)
FFS_HD bool raster_interval(long long start_us, long long end_us, double ratio, double sample_rate, double start_seconds,
                            long long out_len, long long* a, long long* b) {
#pragma clang fp contract(off)
    const double ts = scaled_seconds(start_us, ratio), te = scaled_seconds(end_us, ratio);
    const long long start = (long long)rint((ts - start_seconds) * sample_rate);   // :968-972 (round half even)
    const long long end = start + (long long)rint((te - ts) * sample_rate);        // :974-975
    *a = slice_clamp(start, out_len);                                              // samples[start:end] = ...
    *b = slice_clamp(end, out_len);
    return *a < *b;
}

// OR the bits [a, b) of a word array (edge words masked); `lane` of `lanes` cooperating threads
FFS_DEV void or_bit_range(unsigned* __restrict__ out, long long a, long long b, int lane, int lanes) {
    const long long w_first = a >> 5, w_last = (b - 1) >> 5;
    for (long long w = w_first + lane; w <= w_last; w += lanes) {
        unsigned m = 0xffffffffu;
        if (w == w_first) m &= 0xffffffffu << (a & 31);
        if (w == w_last) m &= 0xffffffffu >> (31 - ((b - 1) & 31));
        atomicOr(&out[w], m);
    }
}

// Batched rasteriser (ffs_rasterize_batch_bits): vector v = the subtitles [sub_first, sub_first + sub_count) of the
// concatenated tracks, times scaled by `ratio`, as `len` bits starting at word `out_word` of the batch buffer.
struct RasterVec {
    long long sub_first, out_word;
    double ratio;
    int sub_count, len;
};
// grid = (blocks over the longest track, vectors); one thread per (vector, subtitle): the interval arithmetic above,
// then the subtitle's 5-20 words (a 2-6 s line at 100 Hz).
__global__ __launch_bounds__(256) void k_rasterize_batch(const long long* __restrict__ start_us, const long long* __restrict__ end_us,
                                                         const unsigned char* __restrict__ meta, const RasterVec* __restrict__ vecs,
                                                         int n_vec, double sample_rate, double start_seconds,
                                                         unsigned* __restrict__ out) {
    for (int v = blockIdx.y; v < n_vec; v += gridDim.y) {
        const RasterVec rv = vecs[v];
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rv.sub_count; i += gridDim.x * blockDim.x) {
            const long long k = rv.sub_first + i;
            if (meta && meta[k]) continue;  // speech_transformers.py:966-967
            long long a, b;
            if (raster_interval(start_us[k], end_us[k], rv.ratio, sample_rate, start_seconds, rv.len, &a, &b))
                or_bit_range(out + rv.out_word, a, b, 0, 1);
        }
    }
}

// subtitle rasteriser: one wave per [start, end) interval, byte stores of 1 (overlaps are unions)
__global__ __launch_bounds__(256) void k_fill_intervals(const int2* __restrict__ iv, int n, unsigned char* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x / 64) + (threadIdx.x / 64);
    const int nw = gridDim.x * (blockDim.x / 64);
    for (int i = wave; i < n; i += nw) {
        const int2 se = iv[i];
        for (int k = se.x + lane; k < se.y; k += 64) out[k] = 1;
    }
}

// bit-packed variant: one wave per interval, one word per lane and step; edge words are masked and every
// word is OR-ed in (overlapping subtitles share words)
__global__ __launch_bounds__(256) void k_fill_intervals_bits(const int2* __restrict__ iv, int n, unsigned* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x / 64) + (threadIdx.x / 64);
    const int nw = gridDim.x * (blockDim.x / 64);
    for (int i = wave; i < n; i += nw) {
        const int2 se = iv[i];
        if (se.y <= se.x) continue;
        const int w_first = se.x >> 5, w_last = (se.y - 1) >> 5;
        for (int w = w_first + lane; w <= w_last; w += 64) {
            unsigned m = 0xffffffffu;
            if (w == w_first) m &= 0xffffffffu << (se.x & 31);
            if (w == w_last) m &= 0xffffffffu >> (31 - ((se.y - 1) & 31));
            atomicOr(&out[w], m);
        }
    }
}

// Two-level vector -> bits: one thread per output word (32 samples).  SRC 0: bytes (!= 0), 1: floats (> thr).
template <int SRC>
__global__ __launch_bounds__(256) void k_pack_bits(const void* __restrict__ src, long long n, float thr,
                                                   unsigned* __restrict__ dst, long long n_words) {
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (long long)gridDim.x * blockDim.x) {
        const long long i0 = w * 32;
        unsigned out = 0;
        if (SRC == 0) {
            const unsigned char* p = reinterpret_cast<const unsigned char*>(src) + i0;
            if (i0 + 32 <= n) {
                unsigned wd[8];
                __builtin_memcpy(wd, p, 32);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    // non-zero flags of four bytes (bits 7, 15, 23, 31) gathered into one nibble: after >> 7 the
                    // flags sit at bits 0, 8, 16, 24 and the multiply lines them up at bits 21..24 (no carries)
                    const unsigned t = nz_flags(wd[k]) >> 7;
                    out |= (((t * 0x00204081u) >> 21) & 0xfu) << (4 * k);
                }
            } else {
                for (int k = 0; k < 32 && i0 + k < n; ++k) out |= (unsigned)(p[k] != 0) << k;
            }
        } else {
            const float* p = reinterpret_cast<const float*>(src) + i0;
            for (int k = 0; k < 32 && i0 + k < n; ++k) out |= (unsigned)(p[k] > thr) << k;
        }
        dst[w] = out;
    }
}

// Sparse reference assembly (MultiSegmentVideoSpeechTransformer, speech_transformers.py:871-890): label runs of
// the sampled windows copied to their place in the (zeroed) full-length vector, clipped at its end.
constexpr int SCATTER_MAX = 32;
struct ScatterSegs {
    long long src_off[SCATTER_MAX], dst_start[SCATTER_MAX], len[SCATTER_MAX];
    int n;
};
__global__ __launch_bounds__(256) void k_scatter_segments(const float* __restrict__ src, ScatterSegs segs, float* __restrict__ out,
                                                          long long out_len) {
    const int sgm = blockIdx.y;
    if (sgm >= segs.n) return;
    const long long dst = segs.dst_start[sgm];
    long long len = segs.len[sgm];
    if (dst + len > out_len) len = out_len - dst;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x)
        out[dst + i] = src[segs.src_off[sgm] + i];
}

__global__ void k_bounds_init(long long* b) {
    b[0] = 0x7fffffffffffffffLL;
    b[1] = -1;
}
__global__ __launch_bounds__(256) void k_speech_bounds(const float* __restrict__ x, long long n, long long* b) {
    long long lo = 0x7fffffffffffffffLL, hi = -1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (x[i] > 0.5f) {
            lo = lo < i ? lo : i;
            hi = hi > i ? hi : i;
        }
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        const long long ol = __shfl_xor(lo, sft, 64), oh = __shfl_xor(hi, sft, 64);
        lo = lo < ol ? lo : ol;
        hi = hi > oh ? hi : oh;
    }
    // one pair of atomics per block (every wave hitting the same two words serialises)
    __shared__ long long s_lo[4], s_hi[4];
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            lo = lo < s_lo[w] ? lo : s_lo[w];
            hi = hi > s_hi[w] ? hi : s_hi[w];
        }
        if (hi >= 0) {
            atomicMin(reinterpret_cast<long long*>(&b[0]), lo);
            atomicMax(reinterpret_cast<long long*>(&b[1]), hi);
        }
    }
}
__global__ void k_bounds_fix(long long* b) {
    if (b[1] < 0) b[0] = -1;
}

}  // namespace ffsa
