// ffs_runs.h -- run-boundary correlation: the EXACT correlation of two bit-packed two-level activity vectors from their
// run boundaries, no transform (gfx950).
//
// A speech-activity vector is a few thousand runs of ones in ~720 000 samples (2 h at 100 Hz).  For 0/1 vectors
// b (candidate, length S) and rho (reference, length R) the count n11(d) = sum_i b[i] * rho[i+d] -- the one
// data-dependent term of the reference's `convolve` (aligners.py:70-74; c(d) is an affine function of n11, the two
// one-sided counts n1x / nx1 and the overlap length, exact_score() in ffs_kernels.h) -- is piecewise linear in the lag:
// with db[p] = b[p] - b[p-1] (+1 where a run starts, -1 one past its end; P = sorted positions) and drho[q] likewise,
//
//     g(d) = n11(d) - n11(d-1) = sum_q drho[q] * b[q-d]
//     h(d) = g(d) - g(d+1)     = sum over boundary pairs (p, q) with q - p = d of db[p] * drho[q]
//
// so all lags of a tile [D0, D1] follow from n11(D0), g(D0) and the sparse second difference h by two running sums.
// Everything is integer arithmetic: the counts are exact, the scores are the same fp64 expression the transform path
// evaluates for its nominees, the maximum is taken over EVERY lag of the window (ties to the largest lag = the first
// k of np.argmax, aligners.py:45-48).  Work ~ |P| * |Q| * W / R boundary coincidences per candidate (W lags): 5e4 for
// subtitle-like vectors under the production window of +-60 s, against ~2e8 flops of the transform path.  Dense vectors
// (more than RUNS_CAP - 1 boundaries, or a coincidence count above the budget) go through the transforms as before,
// a whole sub-batch at a time; the rule is evaluated identically on the host and on the device (runs_over_budget,
// k_runs_chunk_flags).
//
// Index arithmetic modelled in oracle/runs_model.py (CPU-tested against a direct evaluation).
#pragma once
#include "ffs_kernels.h"

namespace ffsa {

constexpr int RUNS_T = 12288;            // lags per tile = per workgroup (the +-6000-lag production window is one tile)
constexpr int RUNS_LPT = RUNS_T / 256;   // consecutive lags per thread in the scan phase
constexpr int RUNS_QCAP = 3580;          // reference boundaries staged in LDS at a time (longer lists: slice by slice)
constexpr int RUNS_PC = 8;               // candidate boundaries a thread holds in registers per walk
constexpr int RUNS_CAP = 32768;          // boundary-list entries per vector incl. the sentinel: also keeps |h(d)| < 2^15
static_assert(RUNS_LPT % 2 == 0 && RUNS_LPT <= 64, "two lags per LDS word, one 64-bit mask per chunk");

struct RunsVec {  // one vector of the call
    const unsigned* words;  // bit-packed samples (bit i = (words[i >> 5] >> (i & 31)) & 1)
    int32_t len;
    int32_t pad;
};

struct RunsBest {  // best lag of one (candidate, tile)
    double score;
    int32_t d;
    unsigned n11, n1x, nx1;
    int32_t pad[2];
};

struct PackVec {  // one 0/1 byte vector of the call and where its bit-packed image goes
    const unsigned char* src;
    unsigned* dst;
    int32_t len;
    int32_t pad;
};

// FFS_DTYPE_U8 vectors of a call -> bit-packed images (bit = byte != 0), all vectors in one launch: grid =
// vectors * chunks_per_vec workgroups, a thread makes one word from 32 bytes.  The packed copy then takes the same
// path as FFS_DTYPE_U1 input (an eighth of the bytes for every later pass).
__global__ __launch_bounds__(256) void k_pack_bytes_batch(const PackVec* __restrict__ vecs, int chunks_per_vec) {
    const int v = blockIdx.x / chunks_per_vec, c = blockIdx.x - v * chunks_per_vec;
    const PackVec pv = vecs[v];
    const long long n = pv.len;
    const long long w = (long long)c * 256 + threadIdx.x;
    const long long n_words = (n + 31) >> 5;
    if (w >= n_words) return;
    const long long i0 = w * 32;
    const unsigned char* __restrict__ p = pv.src + i0;
    unsigned out = 0;
    if (i0 + 32 <= n) {
        unsigned wd[8];
        __builtin_memcpy(wd, p, 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned t = nz_flags(wd[k]) >> 7;  // non-zero flags of four bytes at bits 0, 8, 16, 24
            out |= (((t * 0x00204081u) >> 21) & 0xfu) << (4 * k);
        }
    } else {
        for (int k = 0; k < 32 && i0 + k < n; ++k) out |= (unsigned)(p[k] != 0) << k;
    }
    pv.dst[w] = out;
}

// True when a candidate stays with the transform path: a truncated boundary list, or more expected boundary
// coincidences inside its lag window than `budget`.  Integer arithmetic only -- host and device must agree.
FFS_HD bool runs_over_budget(int n_p, int n_q, long long W, long long R, int cap, long long budget) {
    if (n_p >= cap || n_q >= cap) return true;
    const long long pairs = (long long)n_p * (long long)n_q;  // < 2^30
    return pairs * W / (R > 0 ? R : 1) > budget;
}

// ---------------------------------------------------------------------------------------------------------------
// Boundary lists.  One workgroup per vector; per sweep every thread takes two 16-byte groups, group g of thread t =
// words base + (256 g + t) * 4 .. + 3 -- a wave's load instruction reads 1 KB of consecutive bytes -- and the loads of
// the NEXT sweep are issued before the current one is scanned (a sweep is load latency + scan + writes; eight resident
// blocks per CU in different phases keep the HBM reads going).  Boundary bits e = x ^ (x << 1 | previous bit) (the
// previous word comes from the neighbouring lane), one block scan per sweep of the per-group (boundaries, ones) counts
// packed 2 x 16 bits (a field sums to at most 256 * 128; DPP row shifts + row broadcasts inside a wave, the four wave
// totals through LDS), then every thread writes its own boundaries: q[k] = position, cq[k] = ones of the vector in front
// of it.  q[n] = INT_MAX, cq[n] = all ones.
FFS_DEV unsigned wave_incl_scan_u32(unsigned v) {  // inclusive prefix sum over the 64 lanes
#define FFS_DPP_ADD(ctrl, rows) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false)
    FFS_DPP_ADD(0x111, 0xf);  // row_shr:1
    FFS_DPP_ADD(0x112, 0xf);  // row_shr:2
    FFS_DPP_ADD(0x114, 0xf);  // row_shr:4
    FFS_DPP_ADD(0x118, 0xf);  // row_shr:8   -> inclusive inside every row of 16 lanes
    FFS_DPP_ADD(0x142, 0xa);  // row_bcast:15 -> rows 1 and 3 add the total of the row in front
    FFS_DPP_ADD(0x143, 0xc);  // row_bcast:31 -> rows 2 and 3 add the total of rows 0 + 1
    // (the same six steps as one inline-asm v_add_u32_dpp each -- with the s_nop the DPP hazard needs -- measured 10 % slower)
#undef FFS_DPP_ADD
    return v;
}

__global__ __launch_bounds__(256, 8) void k_runs_extract(const RunsVec* __restrict__ vecs, int* __restrict__ rq,
                                                         int* __restrict__ rc, int2* __restrict__ rn, int cap) {
    constexpr int G = 2, SWEEP = 256 * 4 * G;  // words per sweep
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned* __restrict__ w = vecs[v].words;
    const int len = vecs[v].len;
    int* __restrict__ q = rq + (size_t)v * cap;
    int* __restrict__ cq = rc + (size_t)v * cap;
    const gptr qb = (gptr)q, cqb = (gptr)cq;
    const int nw = (len + 31) >> 5;     // words that hold samples
    const int n_proc = (len >> 5) + 1;  // word len/32 holds position `len`, where a run that reaches the end closes
    const unsigned tail = (len & 31) ? ((1u << (len & 31)) - 1u) : 0xffffffffu;  // valid bits of word nw - 1
    __shared__ unsigned s_e[2][4], s_o[2][4];
    unsigned n_bound = 0, n_ones = 0;  // boundaries / ones in front of this sweep
    unsigned xn[G][4], pn[G];          // the next sweep's words; word in front of each group (lane 0 of a wave only)
    auto request = [&](int base) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int w0 = base + (g * 256 + tid) * 4;
            if (w0 + 4 < nw) {  // in front of the (masked) last word
                uint4 t;
                __builtin_memcpy(&t, w + w0, 16);
                xn[g][0] = t.x, xn[g][1] = t.y, xn[g][2] = t.z, xn[g][3] = t.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = w0 + k;
                    unsigned t = i < nw ? w[i] : 0u;
                    if (i == nw - 1) t &= tail;
                    xn[g][k] = t;
                }
            }
            pn[g] = 0;
            if (lane == 0 && w0 > 0 && w0 - 1 < nw) {
                pn[g] = w[w0 - 1];
                if (w0 - 1 == nw - 1) pn[g] &= tail;
            }
        }
    };
    request(0);
    int buf = 0;
    for (int base = 0; base < n_proc; base += SWEEP, buf ^= 1) {
        unsigned x[G][4], e[G][4], pv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            pv[g] = pn[g];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[g][k] = xn[g][k];
        }
        if (base + SWEEP < n_proc) request(base + SWEEP);
        unsigned pe = 0, po = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            unsigned prev = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x[g][3], 0x138, 0xf, 0xf, false);  // wave_shr:1
            if (lane == 0) prev = pv[g];
            prev >>= 31;
            unsigned ne = 0, no = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                e[g][k] = x[g][k] ^ ((x[g][k] << 1) | prev);
                prev = x[g][k] >> 31;
                ne += __popc(e[g][k]);
                no += __popc(x[g][k]);
            }
            pe |= ne << (16 * g);
            po |= no << (16 * g);
        }
        const unsigned ie = wave_incl_scan_u32(pe), io = wave_incl_scan_u32(po);
        if (lane == 63) s_e[buf][wave] = ie, s_o[buf][wave] = io;
        __syncthreads();  // (two buffers: the next sweep's totals cannot overwrite these while they are read)
        unsigned xe = ie - pe, xo = io - po, te = 0, to = 0;  // exclusive in-block prefixes, block totals
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned a = s_e[buf][i], b = s_o[buf][i];
            if (i < wave) xe += a, xo += b;
            te += a, to += b;
        }
        unsigned gb = n_bound, go = n_ones;  // boundaries / ones in front of group g of thread 0
#pragma unroll
        for (int g = 0; g < G; ++g) {
            int k_out = (int)(gb + ((xe >> (16 * g)) & 0xffffu));
            int ones = (int)(go + ((xo >> (16 * g)) & 0xffffu));
            gb += (te >> (16 * g)) & 0xffffu;
            go += (to >> (16 * g)) & 0xffffu;
            if ((pe >> (16 * g)) & 0xffffu) {
                const int w0 = base + (g * 256 + tid) * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned ee = e[g][k];
                    while (ee) {
                        const int b = __builtin_ctz(ee);
                        if (k_out < cap) {  // (scalar list base + 32-bit byte offset: no 64-bit address arithmetic per store)
                            *(__attribute__((address_space(1))) int*)(qb + 4u * (unsigned)k_out) = 32 * (w0 + k) + b;
                            *(__attribute__((address_space(1))) int*)(cqb + 4u * (unsigned)k_out) = ones + __popc(x[g][k] & ((1u << b) - 1u));
                        }
                        ++k_out;
                        ee &= ee - 1;
                    }
                    ones += __popc(x[g][k]);
                }
            }
        }
        n_bound = gb, n_ones = go;
    }
    if (tid == 0) {
        rn[v] = make_int2((int)n_bound, (int)n_ones);
        if ((int)n_bound < cap) {
            q[n_bound] = INT32_MAX;
            cq[n_bound] = (int)n_ones;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bits [start, start + 64) of a bit-packed vector, zero outside [0, len)
FFS_DEV unsigned long long fetch64(const unsigned* __restrict__ w, int len, long long start) {
    const int nw = (len + 31) >> 5;
    const long long wi = start >> 5;  // floor
    unsigned v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const long long i = wi + k;
        unsigned t = (i >= 0 && i < nw) ? w[i] : 0u;
        if (i == nw - 1 && (len & 31)) t &= (1u << (len & 31)) - 1u;
        v[k] = t;
    }
    const unsigned sh = (unsigned)(start & 31);
    const unsigned lo = __builtin_amdgcn_alignbit(v[1], v[0], sh), hi = __builtin_amdgcn_alignbit(v[2], v[1], sh);
    return ((unsigned long long)hi << 32) | lo;
}

FFS_DEV int lower_bound_i32(const int* a, int n, int x) {  // first k with a[k] >= x
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// the same search by a whole wave (every lane passes the same arguments): 64 probes per step, three dependent loads
// for a list of 32 768 entries instead of fifteen
FFS_DEV int wave_lower_bound_i32(const int* __restrict__ a, int n, int x) {
    const int lane = threadIdx.x & 63;
    int lo = 0, len = n;  // the answer lies in [lo, lo + len]
    while (len > 0) {
        const int step = (len + 63) >> 6;
        const int idx = lo + (lane + 1) * step - 1;  // last element of this lane's sub-range
        const bool below = idx < lo + len && a[idx] < x;
        const int c = __popcll(__ballot(below));  // sub-ranges that lie entirely below x (a prefix of the lanes)
        const int end = lo + len;
        lo += c * step;
        len = (step - 1) < (end - lo) ? (step - 1) : (end - lo);
    }
    return lo;
}

// exclusive scan over the 256 threads of a block of three ints at once (s_tmp: 4 x 3 ints); wrap-around arithmetic
FFS_DEV void block_excl_scan3(int& a, int& b, int& c, int* s_tmp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ia = a, ib = b, ic = c;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int oa = __shfl_up(ia, s, 64), ob = __shfl_up(ib, s, 64), oc = __shfl_up(ic, s, 64);
        if (lane >= s) ia += oa, ib += ob, ic += oc;
    }
    __syncthreads();
    if (lane == 63) s_tmp[wave * 3] = ia, s_tmp[wave * 3 + 1] = ib, s_tmp[wave * 3 + 2] = ic;
    __syncthreads();
    int pa = ia - a, pb = ib - b, pc = ic - c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (i < wave) pa += s_tmp[i * 3], pb += s_tmp[i * 3 + 1], pc += s_tmp[i * 3 + 2];
    a = pa, b = pb, c = pc;
}

// One workgroup per sub-batch (pairs_per_chunk pairs): flag[chunk] = some candidate of it is over budget -- the host
// reaches the same verdict from the list lengths and sends the whole sub-batch through the transforms, so k_runs_corr
// leaves every candidate of a flagged sub-batch alone (no work that would be thrown away).
__global__ __launch_bounds__(256) void k_runs_chunk_flags(const CandDesc* __restrict__ cands, int n_pairs, int n_cand,
                                                          int pairs_per_chunk, const int2* __restrict__ rn, int cap,
                                                          long long budget, int* __restrict__ flags) {
    const int ch = blockIdx.x;
    const int p0 = ch * pairs_per_chunk, p1 = (p0 + pairs_per_chunk) < n_pairs ? (p0 + pairs_per_chunk) : n_pairs;
    int over = 0;
    for (int i = p0 * n_cand + (int)threadIdx.x; i < p1 * n_cand; i += 256) {
        const CandDesc& cd = cands[i];
        if (cd.flags & CAND_NO_LAGS) continue;
        const int pair = i / n_cand;
        const int vr = pair * (n_cand + 1), vs = vr + 1 + (i - pair * n_cand);
        over |= runs_over_budget(rn[vs].x, rn[vr].x, (long long)cd.d_hi - cd.d_lo + 1, cd.R, cap, budget) ? 1 : 0;
    }
    over = __syncthreads_or(over);
    if (threadIdx.x == 0) flags[ch] = over;
}

// grid = (candidates of the call, tiles_max); block = 256 threads; tile t of a candidate covers the lags
// [d_lo + t*RUNS_T, ...] of its window.
//   1. zero the tile's second-difference array (16 bits per lag in 32-bit LDS words: the sum of all additions to a
//      word is v_lo + 65536 * v_hi as an integer, so both halves are recovered exactly whatever the borrows did; one
//      word per lag was measured slower -- 64 KB of LDS leave two blocks per CU instead of four, and the kernel lives
//      off the LDS atomic rate), stage the reference's boundary list in LDS;
//   2. every thread takes a contiguous share of the candidate's boundaries p: one lower-bound search (then linear
//      steps) into the reference's list gives the first boundary q >= p + D0 AND the ones of the reference in front of
//      p + D0 -- summed over p that is n11(D0), and g(D0) is the sum of the parities -- then a walk over the q's up to
//      p + D1 adds +-1 at lag q - p (ds_add_u32);
//   3. two block scans turn h into g and n11 for the thread's RUNS_LPT consecutive lags; the one-sided counts follow
//      from their values at D0 and four 64-bit windows of the two bit vectors (the bits that enter / leave the overlap);
//   4. every lag is scored with exact_score()'s expression; block argmax, ties to the largest lag.
__global__ __launch_bounds__(256) void k_runs_corr(const CandDesc* __restrict__ cands, int n_cand,
                                                   const int* __restrict__ rq, const int* __restrict__ rc,
                                                   const int2* __restrict__ rn, int cap,
                                                   NomList* __restrict__ noms, RescoreAcc* __restrict__ acc,
                                                   RunsBest* __restrict__ best, int tiles_max,
                                                   const int* __restrict__ chunk_flags, int pairs_per_chunk) {
    __shared__ unsigned hist[RUNS_T / 2 + 2];  // second difference h of the tile's lags, 16 bits per lag
    __shared__ int q_lds[RUNS_QCAP + 2];
    __shared__ int s_tmp[16];
    __shared__ double s_sc[4];
    __shared__ int s_d[4];
    const int ci = blockIdx.x, tile = blockIdx.y, tid = threadIdx.x;
    const CandDesc cd = cands[ci];
    if (cd.flags & CAND_NO_LAGS) {
        if (tile == 0 && tid == 0) noms[ci].count = 0, noms[ci].flags = 1, noms[ci].gmax = -INFINITY;
        return;
    }
    const int W = cd.d_hi - cd.d_lo + 1;
    const int n_tiles = (W + RUNS_T - 1) / RUNS_T;
    if (tile >= n_tiles) return;
    const int pair = ci / n_cand;
    if (chunk_flags[pair / pairs_per_chunk]) return;  // this sub-batch goes through the transforms (k_runs_chunk_flags)
    const int vr = pair * (n_cand + 1), vs = vr + 1 + (ci - pair * n_cand);
    const int2 ns = rn[vs], nr = rn[vr];
    const int n_p = ns.x, n_q = nr.x, S = cd.S, R = cd.R;
    const int D0 = cd.d_lo + tile * RUNS_T;
    const int Wt = (cd.d_hi - D0 + 1) < RUNS_T ? (cd.d_hi - D0 + 1) : RUNS_T;
    const int* __restrict__ Pg = rq + (size_t)vs * cap;
    const int* __restrict__ Qg = rq + (size_t)vr * cap;
    const int* __restrict__ CQg = rc + (size_t)vr * cap;
    const bool whole = n_q <= RUNS_QCAP;  // the reference's whole list fits the staging area
    // ones of rho in [0, x) = sum_k sgn_k * min(x, Q[k]) (sgn = -1 at run starts, +1 at run ends): the two positions
    // that bound the overlap at lag D0
    const int r_lo = D0 > 0 ? D0 : 0, r_hi = (S + D0) < R ? (S + D0) : R;
    int rsum = 0;
    for (int k = tid; k <= n_q; k += 256) {
        const int qv = Qg[k];
        if (whole) q_lds[k] = qv;
        if (k < n_q) {
            const int m_hi = qv < r_hi ? qv : r_hi, m_lo = qv < r_lo ? qv : r_lo;
            rsum += (k & 1) ? (m_hi - m_lo) : (m_lo - m_hi);
        }
    }
    for (int i = tid; i < RUNS_T / 2 + 2; i += 256) hist[i] = 0u;
    __syncthreads();
    const int i0 = D0 < 0 ? -D0 : 0, i1 = (R - D0) < S ? (R - D0) : S;
    const int wmax = Wt - 2;                                 // h is needed for the lags D0 .. D1 - 1
    const unsigned wlim = wmax >= 0 ? (unsigned)wmax : 0u;  // (a one-lag tile never reads h: a stray add at 0 is harmless)
    int n11p = 0, gp = 0, bsum = 0;
    // One group = RUNS_PC consecutive boundaries p of the candidate, positions in registers: the boundaries q that any of
    // them can meet inside the tile form ONE stretch of the reference's list, so every q is read once per group and tested
    // against the RUNS_PC positions without a dependent load in between; the additions are fire-and-forget LDS atomics.
    // The same walk counts, for every p, the q's in front of p + D0 -- its lower bound, which gives the ones of the
    // reference in front of p + D0 (n11(D0)) and their parity (g(D0)).  Q is indexed absolutely; [lo, hi] is the stretch
    // that is readable through it (the staged slice, or the whole list in global memory).
    auto group = [&](auto Q, int kc, int k1, int lo, int hi) {
        int x[RUNS_PC], cnt[RUNS_PC];
        int x_last = 0;
#pragma unroll
        for (int i = 0; i < RUNS_PC; ++i) {
            const bool valid = kc + i < k1;
            x[i] = valid ? Pg[kc + i] + D0 : 0x3fffffff;  // never met: q - x < 0 for every real boundary
            cnt[i] = 0;
            if (valid) x_last = x[i];
        }
        const int lb0 = lo + lower_bound_i32(Q + lo, hi - lo, x[0]);
        const int xe = x_last + wmax;
        int qq = Q[lb0];
        int j = lb0;
        // boundaries in front of the group's last position also count towards the lower bounds of its members
        for (; qq < x_last; ++j) {
            const int qn = Q[j + 1];
            const int sq = (j & 1) ? -1 : 1;
#pragma unroll
            for (int i = 0; i < RUNS_PC; ++i) {
                const int d = qq - x[i];
                cnt[i] += d < 0;
                if ((unsigned)d <= wlim) atomicAdd(&hist[d >> 1], (unsigned)((i & 1) ? -sq : sq) << ((d & 1) * 16));
            }
            qq = qn;
        }
        for (; qq <= xe; ++j) {
            const int qn = Q[j + 1];
            const int sq = (j & 1) ? -1 : 1;
#pragma unroll
            for (int i = 0; i < RUNS_PC; ++i) {
                const int d = qq - x[i];
                if ((unsigned)d <= wlim) atomicAdd(&hist[d >> 1], (unsigned)((i & 1) ? -sq : sq) << ((d & 1) * 16));
            }
            qq = qn;
        }
#pragma unroll
        for (int i = 0; i < RUNS_PC; ++i) {
            if (kc + i < k1) {
                const int lb = lb0 + cnt[i];
                const int sp = (i & 1) ? -1 : 1;  // db[p]: groups start at even indices
                int ones = CQg[lb];
                if (lb & 1) ones -= Q[lb] - x[i];  // inside a run: the run's ones from x on are not in front of x
                n11p -= sp * ones;
                gp -= sp * (lb & 1);
                const int p = x[i] - D0;
                const int m1 = p < i1 ? p : i1, m0 = p < i0 ? p : i0;
                bsum -= sp * (m1 - m0);  // ones of b in [i0, i1) = sum_k sgn_k * (min(i1, P[k]) - min(i0, P[k]))
            }
        }
    };
    // Rounds of 256 groups (2048 consecutive candidate boundaries).  A reference list that does not fit the staging area
    // is staged slice by slice: the q's a round can meet are one stretch [first q >= P[first] + D0, first q beyond
    // P[last] + D1], found by two wave-wide 64-ary searches; only a round whose stretch is still too long (a reference
    // far denser than the candidate) walks the list in global memory.
    constexpr int ROUND = 256 * RUNS_PC;
    int st_lo = 0, st_hi = n_q;
    bool staged_ok = whole;
    for (int rs = 0; rs < n_p; rs += ROUND) {
        const int re = (rs + ROUND) < n_p ? (rs + ROUND) : n_p;
        if (!whole) {
            __syncthreads();  // the previous round is done with the staged slice
            if (tid < 128) {
                const int xq = (tid < 64) ? Pg[rs] + D0 : Pg[re - 1] + D0 + (int)wlim + 1;
                const int v = wave_lower_bound_i32(Qg, n_q, xq);
                if ((tid & 63) == 0) s_tmp[tid >> 6] = v;
            }
            __syncthreads();
            st_lo = s_tmp[0], st_hi = s_tmp[1];
            staged_ok = st_hi - st_lo <= RUNS_QCAP;
            if (staged_ok)
                for (int k = st_lo + tid; k <= st_hi; k += 256) q_lds[k - st_lo] = Qg[k];
            __syncthreads();
        }
        const int kc = rs + tid * RUNS_PC;
        if (kc < re) {
            if (staged_ok)
                group((const int*)q_lds - st_lo, kc, re, st_lo, st_hi);
            else
                group(Qg, kc, re, 0, n_q);
        }
    }
    // block sums of (n11p, gp, bsum, rsum)
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        n11p += __shfl_xor(n11p, s, 64);
        gp += __shfl_xor(gp, s, 64);
        bsum += __shfl_xor(bsum, s, 64);
        rsum += __shfl_xor(rsum, s, 64);
    }
    if ((tid & 63) == 0) {
        const int wv = tid >> 6;
        s_tmp[wv * 4] = n11p, s_tmp[wv * 4 + 1] = gp, s_tmp[wv * 4 + 2] = bsum, s_tmp[wv * 4 + 3] = rsum;
    }
    __syncthreads();  // also: every addition to hist has landed
    const int n11_0 = s_tmp[0] + s_tmp[4] + s_tmp[8] + s_tmp[12];
    const int g_0 = s_tmp[1] + s_tmp[5] + s_tmp[9] + s_tmp[13];
    const int n1x_0 = s_tmp[2] + s_tmp[6] + s_tmp[10] + s_tmp[14];
    const int nx1_0 = s_tmp[3] + s_tmp[7] + s_tmp[11] + s_tmp[15];
    __syncthreads();  // s_tmp is reused by the scans

    const int c = tid * RUNS_LPT;  // this thread's lags: D0 + c .. D0 + c + RUNS_LPT - 1
    auto h_pair = [&](int i, int& h0, int& h1) {
        const unsigned wv = hist[(c >> 1) + i];
        h0 = (int)(short)(wv & 0xffffu);
        h1 = (int)(wv - (unsigned)h0) >> 16;
    };
    int hs = 0;
#pragma unroll 4
    for (int i = 0; i < RUNS_LPT / 2; ++i) {
        int h0, h1;
        h_pair(i, h0, h1);
        hs += h0 + h1;
    }
    // the bits that enter (+) and leave (-) the two one-sided counts when the lag grows by one, lag c + i = bit i:
    //   n1x(d+1) = n1x(d) + b[-d-1] - b[R-d-1],   nx1(d+1) = nx1(d) - rho[d] + rho[S+d]   (zero outside the vectors)
    const long long dc = (long long)D0 + c;
    const unsigned* sw = reinterpret_cast<const unsigned*>(cd.s);
    const unsigned* rw = reinterpret_cast<const unsigned*>(cd.r);
    const unsigned long long m_in1x = __brevll(fetch64(sw, S, -dc - 64));
    const unsigned long long m_out1x = __brevll(fetch64(sw, S, (long long)R - dc - 64));
    const unsigned long long m_outx1 = fetch64(rw, R, dc);
    const unsigned long long m_inx1 = fetch64(rw, R, (long long)S + dc);
    const unsigned long long lpt_mask = RUNS_LPT == 64 ? ~0ull : ((1ull << RUNS_LPT) - 1ull);
    int d1x = __popcll(m_in1x & lpt_mask) - __popcll(m_out1x & lpt_mask);
    int dx1 = __popcll(m_inx1 & lpt_mask) - __popcll(m_outx1 & lpt_mask);
    int hpre = hs, z0 = 0, z1 = 0;
    block_excl_scan3(hpre, z0, z1, s_tmp);
    const int g_c = g_0 - hpre;  // g at the thread's first lag
    int gs = 0;
    {
        int g = g_c;
#pragma unroll 4
        for (int i = 0; i < RUNS_LPT / 2; ++i) {
            int h0, h1;
            h_pair(i, h0, h1);
            g -= h0;
            gs += g;
            g -= h1;
            gs += g;
        }
    }
    block_excl_scan3(gs, d1x, dx1, s_tmp);  // now exclusive prefixes
    int n11 = n11_0 + gs, n1x = n1x_0 + d1x, nx1 = nx1_0 + dx1;
    double bs = -INFINITY;
    int bd = INT32_MIN;
    unsigned b11 = 0, b1x = 0, bx1 = 0;
    if (c < Wt) {
        int g = g_c;
        const int lim = (Wt - c) < RUNS_LPT ? (Wt - c) : RUNS_LPT;
        for (int i = 0; i < lim; i += 2) {
            int h0, h1;
            h_pair(i >> 1, h0, h1);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ii = i + e;
                if (ii < lim) {
                    const int d = D0 + c + ii;
                    const double sc = two_level_score(cd, n11, n1x, nx1, d);
                    if (sc >= bs) bs = sc, bd = d, b11 = (unsigned)n11, b1x = (unsigned)n1x, bx1 = (unsigned)nx1;
                    g -= e ? h1 : h0;
                    n11 += g;
                    n1x += (int)((m_in1x >> ii) & 1ull) - (int)((m_out1x >> ii) & 1ull);
                    nx1 += (int)((m_inx1 >> ii) & 1ull) - (int)((m_outx1 >> ii) & 1ull);
                }
            }
        }
    }
    // block argmax: larger score, then larger lag
    double ws = bs;
    int wd = bd;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const double os = __shfl_xor(ws, s, 64);
        const int od = __shfl_xor(wd, s, 64);
        if (os > ws || (os == ws && od > wd)) ws = os, wd = od;
    }
    if ((tid & 63) == 0) s_sc[tid >> 6] = ws, s_d[tid >> 6] = wd;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (s_sc[i] > ws || (s_sc[i] == ws && s_d[i] > wd)) ws = s_sc[i], wd = s_d[i];
    if (bd == wd && wd != INT32_MIN) {  // the one thread that owns the winning lag
        if (n_tiles == 1) {
            NomList& nl = noms[ci];
            nl.count = 1;
            nl.flags = 0;
            nl.gmax = (float)bs;
            nl.d[0] = bd;
            nl.val[0] = (float)bs;
            RescoreAcc& a = acc[(size_t)ci * KNOM];
            a.n11 = b11, a.n1x = b1x, a.nx1 = bx1;
        } else {
            RunsBest& o = best[(size_t)ci * tiles_max + tile];
            o.score = bs, o.d = bd, o.n11 = b11, o.n1x = b1x, o.nx1 = bx1;
        }
    }
}

// candidates whose window spans several tiles: best tile result (larger score, then larger lag = later tile)
__global__ void k_runs_pick(const CandDesc* __restrict__ cands, int n, int n_cand, const RunsBest* __restrict__ best,
                            int tiles_max, NomList* __restrict__ noms, RescoreAcc* __restrict__ acc,
                            const int* __restrict__ chunk_flags, int pairs_per_chunk) {
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= n) return;
    const CandDesc& cd = cands[ci];
    if (cd.flags & CAND_NO_LAGS) return;
    const int W = cd.d_hi - cd.d_lo + 1;
    const int n_tiles = (W + RUNS_T - 1) / RUNS_T;
    if (n_tiles <= 1) return;
    if (chunk_flags[(ci / n_cand) / pairs_per_chunk]) return;
    RunsBest b = best[(size_t)ci * tiles_max];
    for (int t = 1; t < n_tiles; ++t) {
        const RunsBest& o = best[(size_t)ci * tiles_max + t];
        if (o.score >= b.score) b = o;
    }
    NomList& nl = noms[ci];
    nl.count = 1;
    nl.flags = 0;
    nl.gmax = (float)b.score;
    nl.d[0] = b.d;
    nl.val[0] = (float)b.score;
    RescoreAcc& a = acc[(size_t)ci * KNOM];
    a.n11 = b.n11, a.n1x = b.n1x, a.nx1 = b.nx1;
}

}  // namespace ffsa
