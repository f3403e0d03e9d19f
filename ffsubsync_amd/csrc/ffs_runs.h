// ffs_runs.h -- run-boundary correlation: the EXACT correlation of two two-level activity vectors from their run
// boundaries, no transform (gfx950).
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
// subtitle-like vectors under the production window of +-60 s, against ~2e8 flops of the transform path.
//
// A vector reaches the kernels as a BOUNDARY LIST (RunsRef, round 5): either extracted here from its bit-packed samples
// (k_runs_extract) or handed over by its producer -- the subtitle rasteriser knows its intervals (k_rasterize_runs: no
// bitmap is ever written or read), a bit-packed label vector is converted once (ffs_runs_from_bits) and reused by
// every later solve.  Dense vectors (a list of RUNS_CAP or more entries, or a coincidence count above the budget) go
// through the transforms, a whole sub-batch at a time (k_runs_chunk_flags decides on the device, the host reads one int
// per sub-batch); list-only vectors are expanded to bits for that (k_runs_expand).
//
// Index arithmetic modelled in oracle/runs_model.py (CPU-tested against a direct evaluation).
#pragma once
#include "ffs_kernels.h"

namespace ffsa {

#ifndef FFS_RUNS_THREADS  // (A/B builds only: profiles/r06_runs_experiments.json, "seven-wave workgroups")
#define FFS_RUNS_THREADS 512
#define FFS_RUNS_T 12288
#define FFS_RUNS_QCAP 3070
#endif
constexpr int RUNS_THREADS = FFS_RUNS_THREADS;    // k_runs_corr workgroup
constexpr int RUNS_WAVES = RUNS_THREADS / 64;
constexpr int RUNS_T = FFS_RUNS_T;                // lags per tile = per workgroup (the +-6000-lag production window is one tile)
constexpr int RUNS_LPT = RUNS_T / RUNS_THREADS;   // consecutive lags per thread in the scan phase
constexpr int RUNS_QCAP = FFS_RUNS_QCAP;          // reference boundaries staged in LDS at a time (longer lists: slice by slice)
static_assert(RUNS_T % RUNS_THREADS == 0 && RUNS_LPT % 4 == 0 && RUNS_LPT <= 32, "whole groups of four lags per thread, 32-bit edge windows");
constexpr int RUNS_CAP = 32768;                   // boundary-list entries per vector incl. the sentinel (plan-owned lists)
#ifndef FFS_RUNS_TPW
#define FFS_RUNS_TPW 2
#endif
constexpr int RUNS_TPW = FFS_RUNS_TPW;             // wave tasks (64 candidate RUNS each, one per lane) a wave advances together
constexpr int RUNS_QSENT = 0x1fffffff;            // staged sentinel: beyond every position (a plan's vectors are shorter than 2^24); DOUBLED in
                                                  // LDS, and sentinel - position must not overflow 32 bits for any position > -2^26
static_assert(RUNS_LPT <= 32 && RUNS_LPT % 4 == 0 && RUNS_QCAP % 2 == 0, "one 32-bit mask per thread, 8-byte histogram loads");

// One vector of a call as the run-boundary kernels see it.  e[k] = (position of boundary k, ones of the vector in front of
// it), k < n, sorted; e[n] = (INT32_MAX, all ones); n is even (a run that reaches the end closes at position len).
// hdr->x = n (>= the list's capacity when it was truncated), hdr->y = ones.
// A plan-owned list of a CANDIDATE that arrived as bits (bits != null, cap > 0, the vector is not its pair's reference) is
// COMPACT: e holds 4-byte positions only (nothing reads a candidate's ones-in-front column: k_runs_corr counts the
// candidate's ones from the positions) -- the 8-byte entry stores are what holds k_runs_extract below the read ceiling, and
// seven of a pair's eight lists are candidates'.  References, threshold planes and every caller-owned list keep
// (position, ones in front).
struct RunsRef {
    const int2* e;
    const int2* hdr;
    const unsigned* bits;  // the bit-packed samples (bit i = (bits[i >> 5] >> (i & 31)) & 1); null: the list is all there is
    int32_t len;           // samples
    int32_t cap;           // entries e[] has room for (incl. the sentinel)
};

// The kernels read lists and bitmaps through GLOBAL-address-space pointers: a pointer fetched from a RunsRef in memory is
// generic to the compiler, and generic (flat) loads may alias LDS -- they would be ordered against every histogram
// atomic and staging store, one load latency at a time.
struct RunEntry {
    int pos, ones;
};
typedef const __attribute__((address_space(1))) RunEntry* GEntries;
typedef const __attribute__((address_space(1))) int* GInts;
typedef const __attribute__((address_space(1))) unsigned* GWords;

struct RunsBest {  // best lag of one (candidate, tile)
    double score;
    int32_t d;
    int32_t pad;
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
FFS_HD bool runs_over_budget(long long n_p, long long n_q, long long cap_p, long long cap_q, long long W, long long R, long long budget) {
    if (n_p >= cap_p || n_q >= cap_q) return true;
    const long long pairs = n_p * n_q;  // < 2^30
    return pairs * W / (R > 0 ? R : 1) > budget;
}

// ---------------------------------------------------------------------------------------------------------------
// Boundary lists from bit-packed samples.  One workgroup per vector; per sweep every thread takes two 16-byte groups,
// group g of thread t = words base + (NT g + t) * 4 .. + 3 -- a wave's load instruction reads 1 KB of consecutive bytes
// -- and the loads of the NEXT sweep are issued before the current one is scanned.  Boundary bits e = x ^ (x << 1 |
// previous bit) (the previous word comes from the neighbouring lane), one block scan per sweep of the per-group
// (boundaries, ones) counts (256 threads: packed 2 x 16 bits, a field sums to at most 256 * 128; DPP row shifts + row
// broadcasts inside a wave, the wave totals through LDS); what happens to the boundary words then is described at
// runs_extract_body (round 6: compacted into per-wave rings, one list entry per lane).  e[k] = (position, ones of the
// vector in front of it).  The sweep loop stops as soon as the list is full (a vector that dense goes through the
// transforms anyway -- its remaining words are never read).
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

// maximum over the 64 lanes of a wave, DPP only (no ds_bpermute round trips): the same six steps as the prefix sum; every
// lane receives the result of lane 63 (a uniform value: v_readlane)
FFS_DEV float wave_max_f32(float v) {
#define FFS_DPP_MAX(ctrl, rows) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rows, 0xf, false)))
    FFS_DPP_MAX(0x111, 0xf);  // row_shr:1 (lanes without a source keep their own value: old = v)
    FFS_DPP_MAX(0x112, 0xf);
    FFS_DPP_MAX(0x114, 0xf);
    FFS_DPP_MAX(0x118, 0xf);
    FFS_DPP_MAX(0x142, 0xa);  // row_bcast:15
    FFS_DPP_MAX(0x143, 0xc);  // row_bcast:31
#undef FFS_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Round 6: the words that HOLD a boundary (one in eleven for subtitle-like vectors) are compacted before anything is done
// bit by bit.  Round 5 ran the per-bit loop of every word slot on the whole wave -- with 64 lanes some lane nearly always
// has a boundary in the slot, so every wave paid eight loop bodies per sweep for ~45 boundaries (17 VALU operations per
// word, 0.55 of 8 TB/s).  Now a lane appends each of its boundary words as one 16-byte item (word index + the bit in front,
// the word, ones and boundaries of the vector in front of it -- all known after the sweep's block scan) to a ring in LDS
// that belongs to its WAVE (ranks from one DPP scan; LDS operations of a wave complete in order, so no barrier), and
// whenever 64 items are waiting the wave turns them into list entries, one item per lane.
constexpr int RUNS_XQ = 256;  // items per wave ring (a group of a sweep adds at most 256 per wave)
// NT threads per workgroup: 256 (eight workgroups per CU: throughput, calls with thousands of vectors) or 1024 (a sweep is
// 32 KB: a 90 KB vector takes 3 dependent memory round trips instead of 11 -- small calls, where the latency of ONE
// vector is what the caller waits for).
template <int NT>
FFS_DEV void runs_extract_lds(unsigned (*&s_e)[NT / 64 * 2], unsigned (*&s_o)[NT / 64 * 2], uint4 (*&s_q)[RUNS_XQ]) {
    __shared__ unsigned l_e[2][NT / 64 * 2], l_o[2][NT / 64 * 2];
    __shared__ __attribute__((aligned(16))) uint4 l_q[NT / 64][RUNS_XQ];
    s_e = l_e, s_o = l_o, s_q = l_q;
}
template <int NT, bool compact = false>
FFS_DEV void runs_extract_body(const unsigned* __restrict__ w_generic, const int len, int2* __restrict__ e, int2* __restrict__ hdr,
                               const int cap) {
    constexpr int NWV = NT / 64;                // waves
    constexpr int G = 2, SWEEP = NT * 4 * G;  // words per sweep
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const gptr eb = (gptr)e;
    const int nw = (len + 31) >> 5;     // words that hold samples
    const int n_proc = (len >> 5) + 1;  // word len/32 holds position `len`, where a run that reaches the end closes
    const unsigned tail = (len & 31) ? ((1u << (len & 31)) - 1u) : 0xffffffffu;  // valid bits of word nw - 1
    // (LDS through a helper that is a template of NT alone: the two instantiations of this body for one NT share it)
    unsigned(*s_e)[NWV * 2], (*s_o)[NWV * 2];
    uint4(*s_q)[RUNS_XQ];
    runs_extract_lds<NT>(s_e, s_o, s_q);
    uint4* const ring = s_q[wave];
    unsigned q_head = 0, q_tail = 0;   // items [head, tail) of this wave's ring are waiting (wave-uniform)
    unsigned n_bound = 0, n_ones = 0;  // boundaries / ones in front of this sweep
    unsigned xn[G][4], pn[G];          // the next sweep's words; word in front of each group (lane 0 of a wave only)
    // The loads go through a buffer resource that covers exactly the vector's words: a word behind the end reads as 0 (and
    // so does "the word in front of word 0"), no fault, NO BRANCH -- round 5's request() chose between a 16-byte load and
    // four guarded ones, and the compiler waited for the data at the join of the two paths: the "loads of the next sweep"
    // were waited for on the spot, every sweep paid the full memory latency (0.55 of 8 TB/s at eight workgroups per CU).
    // Only the partial last word needs its mask, applied where the words are used.
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(w_generic), 0, nw * 4, 0x00020000);
    auto request = [&](int base) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int w0 = base + (g * NT + tid) * 4;
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            // (nontemporal: the samples are read exactly once -- 0.165 -> 0.158 us per pair, profiles/r06_runs_experiments.json)
            const v4u t = __builtin_amdgcn_raw_buffer_load_b128(vrs, w0 * 4, 0, 2);
            xn[g][0] = t.x, xn[g][1] = t.y, xn[g][2] = t.z, xn[g][3] = t.w;
            pn[g] = 0;
            if (lane == 0) pn[g] = __builtin_amdgcn_raw_buffer_load_b32(vrs, w0 * 4 - 4, 0, 0);
        }
    };
    // up to 64 waiting items -> list entries, one item per lane: entry k = (position, ones of the vector in front of it)
    auto drain = [&]() {
        const unsigned n = (q_tail - q_head) < 64u ? (q_tail - q_head) : 64u;
        if ((unsigned)lane < n) {
            const uint4 it = ring[(q_head + (unsigned)lane) & (RUNS_XQ - 1)];
            unsigned eb_k = it.y ^ __builtin_amdgcn_alignbit(it.y, it.x, 31);  // x ^ (x << 1 | bit in front)
            const int pos0 = (int)((it.x & 0x7fffffffu) << 5);
            int k_out = (int)it.w;
            while (eb_k) {
                const int b = __builtin_ctz(eb_k);
                if (k_out < cap) {  // (scalar list base + 32-bit byte offset: no 64-bit address arithmetic per store)
                    typedef int v2i __attribute__((ext_vector_type(2)));
                    if (compact)  // positions only (see RunsRef): half the bytes of the stores that cost this kernel most
                        *(__attribute__((address_space(1))) int*)(eb + 4u * (unsigned)k_out) = pos0 + b;
                    else
                        *(__attribute__((address_space(1))) v2i*)(eb + 8u * (unsigned)k_out) =
                            (v2i){pos0 + b, (int)it.z + (int)__popc(it.y & ((1u << b) - 1u))};
                }
                ++k_out;
                eb_k &= eb_k - 1;
            }
        }
        q_head += n;
    };
    request(0);
    int buf = 0;
    for (int base = 0; base < n_proc && n_bound < (unsigned)cap; base += SWEEP, buf ^= 1) {
        unsigned x[G][4], pv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            pv[g] = pn[g];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[g][k] = xn[g][k];
            // the partial last word: its bits behind the vector's end do not count (one thread of one sweep)
            const int w0 = base + (g * NT + tid) * 4;
            if ((unsigned)(nw - 1 - w0) < 4u) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (w0 + k == nw - 1) x[g][k] &= tail;
            }
            if (w0 == nw) pv[g] &= tail;  // (lane 0's word in front is the last word)
        }
        // The list entries of the items that are waiting are stored BEFORE the next sweep's loads are requested: gfx950 counts
        // loads and stores in one in-order counter, so a store issued behind the loads would have to be acknowledged before
        // the loads' data may be used -- a sweep then pays the write latency on top of the read latency.
        while (q_tail - q_head >= 64u) drain();
        if (base + SWEEP < n_proc) request(base + SWEEP);
        unsigned ne[G], no[G], pc = 0;  // per group: boundaries and ones of this thread; its boundary WORDS (16 bits per group)
        unsigned fr[G];                  // bit 31 of the word in front of each group
#pragma unroll
        for (int g = 0; g < G; ++g) {
            unsigned prev = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x[g][3], 0x138, 0xf, 0xf, false);  // wave_shr:1
            if (lane == 0) prev = pv[g];
            fr[g] = prev;
            unsigned nc = 0;
            ne[g] = no[g] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned e_k = x[g][k] ^ __builtin_amdgcn_alignbit(x[g][k], prev, 31);  // x ^ (x << 1 | bit 31 of the word in front)
                prev = x[g][k];
                ne[g] += __popc(e_k);
                no[g] += __popc(x[g][k]);
                nc += e_k != 0u ? 1u : 0u;
            }
            pc |= nc << (16 * g);
        }
        const unsigned ic = wave_incl_scan_u32(pc);
        // exclusive in-block prefixes and block totals of (boundaries, ones) per group
        unsigned qe = 0, qo = 0, se = 0, so = 0;  // NT <= 256: both groups packed 2 x 16 bits (a field sums to at most 256 * 128)
        unsigned ie[G], io[G];                    // wider workgroups: one scan per group and quantity
        if (NT <= 256) {
            static_assert(G == 2, "two groups per packed scan");
            const unsigned pe = ne[0] | (ne[1] << 16), po = no[0] | (no[1] << 16);
            const unsigned je = wave_incl_scan_u32(pe), jo = wave_incl_scan_u32(po);
            if (lane == 63) s_e[buf][wave] = je, s_o[buf][wave] = jo;
            __syncthreads();  // (two buffers: the next sweep's totals cannot overwrite these while they are read)
            qe = je - pe, qo = jo - po;
#pragma unroll
            for (int i = 0; i < NWV; ++i) {
                const unsigned a = s_e[buf][i], b = s_o[buf][i];
                if (i < wave) qe += a, qo += b;
                se += a, so += b;
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                ie[g] = wave_incl_scan_u32(ne[g]), io[g] = wave_incl_scan_u32(no[g]);
                if (lane == 63) s_e[buf][wave * G + g] = ie[g], s_o[buf][wave * G + g] = io[g];
                ie[g] -= ne[g], io[g] -= no[g];
            }
            __syncthreads();
        }
        const unsigned wc = (unsigned)__builtin_amdgcn_readlane((int)ic, 63);  // this wave's boundary words, per group
        const unsigned rc = ic - pc;                                           // ranks inside the wave
        unsigned gb = n_bound, go = n_ones;  // boundaries / ones in front of group g of thread 0
#pragma unroll
        for (int g = 0; g < G; ++g) {
            unsigned xe_g, xo_g, te_g = 0, to_g = 0;
            if (NT <= 256) {
                xe_g = (qe >> (16 * g)) & 0xffffu, xo_g = (qo >> (16 * g)) & 0xffffu, te_g = (se >> (16 * g)) & 0xffffu, to_g = (so >> (16 * g)) & 0xffffu;
            } else {
                xe_g = ie[g], xo_g = io[g];
                for (int i = 0; i < NWV; ++i) {
                    const unsigned a = s_e[buf][i * G + g], b = s_o[buf][i * G + g];
                    if (i < wave) xe_g += a, xo_g += b;
                    te_g += a, to_g += b;
                }
            }
            unsigned k_out = gb + xe_g;
            unsigned ones = go + xo_g;
            gb += te_g;
            go += to_g;
            const unsigned add = (wc >> (16 * g)) & 0xffffu;  // (<= 256 = the ring)
            {
            while (q_tail - q_head + add > (unsigned)RUNS_XQ) drain();
            unsigned slot = q_tail + ((rc >> (16 * g)) & 0xffffu);
            const unsigned w0 = (unsigned)(base + (g * NT + tid) * 4);
            unsigned prev = fr[g];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned e_k = x[g][k] ^ __builtin_amdgcn_alignbit(x[g][k], prev, 31);  // (again: cheaper than keeping eight words alive)
                if (e_k) {
                    ring[slot & (RUNS_XQ - 1)] = make_uint4((w0 + k) | (prev & 0x80000000u), x[g][k], ones, k_out);
                    ++slot;
                }
                k_out += __popc(e_k);
                ones += __popc(x[g][k]);
                prev = x[g][k];
            }
            q_tail += add;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the wave's own ring: written and read in order, no barrier)
            }
        }
        n_bound = gb, n_ones = go;
    }
    while (q_tail != q_head) drain();
    if (tid == 0) {
        *hdr = make_int2((int)n_bound, (int)n_ones);
        if ((int)n_bound < cap) {
            if (compact)
                reinterpret_cast<int*>(e)[n_bound] = INT32_MAX;
            else
                e[n_bound] = make_int2(INT32_MAX, (int)n_ones);
        }
    }
}

// Density probe (a stream of dense calls: FFS_ALGO_AUTO would extract lists only to throw them away): one wave per
// vector reads RUNS_PROBE_WORDS words spread evenly over it and scales their boundary bits up to the whole vector --
// hdr = (estimated boundaries, 0).  k_runs_chunk_flags on the estimates then says which sub-batches are far over budget.
constexpr int RUNS_PROBE_WORDS = 256;
__global__ __launch_bounds__(256) void k_runs_probe(const RunsRef* __restrict__ refs, int n_vec) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= n_vec) return;
    const RunsRef r = refs[v];
    if (!r.bits) return;  // (a list-only vector has its true count in the header already)
    const GWords w = (GWords)r.bits;
    const int nw = r.len >> 5;  // whole words only
    unsigned cnt = 0;
    if (nw > 0) {
#pragma unroll
        for (int u = 0; u < RUNS_PROBE_WORDS / 64; ++u) {
            const long long k = ((long long)(u * 64 + lane) * nw) / RUNS_PROBE_WORDS;
            const unsigned x = w[k];
            cnt += __popc((x ^ (x << 1)) & 0xfffffffeu);  // value changes between neighbouring samples inside the word: 31 places
        }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) cnt += __shfl_xor(cnt, s, 64);
    if (lane == 0) {
        const long long sampled = (long long)(nw < RUNS_PROBE_WORDS ? (nw > 0 ? nw : 1) : RUNS_PROBE_WORDS) * 31;
        long long est = (long long)cnt * (long long)r.len / sampled;
        if (est > 0x3fffffff) est = 0x3fffffff;
        const_cast<int2*>(r.hdr)[0] = make_int2((int)est, 0);
    }
}

// every vector of a call that arrived as bits (list-only vectors are skipped)
// (`zero`, `zero_words`: the call's sub-batch flags + statistics, cleared by workgroup 0 -- the kernels that use them run behind
// this one; a separate hipMemsetAsync is two ~4.5 us fill kernels on the stream, a tenth of a 128-pair step)
FFS_DEV void runs_extract_zero(int* __restrict__ zero, int zero_words) {
    if (zero && blockIdx.x == 0)
        for (int i = threadIdx.x; i < zero_words; i += blockDim.x) zero[i] = 0;
}
// (`cstride`, `n_plain`: vectors v < n_plain with v % cstride != 0 are the CANDIDATES of a call: their plan-owned lists are
// written compact, see RunsRef; cstride = 0: none)
FFS_DEV bool runs_compact_vec(int cstride, int n_plain) {
    return cstride > 0 && (int)blockIdx.x < n_plain && (int)blockIdx.x % cstride != 0;
}
__global__ __launch_bounds__(256, 8) void k_runs_extract(const RunsRef* __restrict__ refs, int* __restrict__ zero, int zero_words,
                                                         int cstride, int n_plain) {
    runs_extract_zero(zero, zero_words);
    const RunsRef r = refs[blockIdx.x];
    if (!r.bits) return;
    if (runs_compact_vec(cstride, n_plain))
        runs_extract_body<256, true>(r.bits, r.len, const_cast<int2*>(r.e), const_cast<int2*>(r.hdr), r.cap);
    else
        runs_extract_body<256, false>(r.bits, r.len, const_cast<int2*>(r.e), const_cast<int2*>(r.hdr), r.cap);
}
// The same for calls with few vectors: a vector's latency, not the chip's throughput, is what such a call waits for.  At
// most 256 vectors: 1024-thread workgroups (one per CU, 3 sweeps per 90 KB vector); at most 768: 512 threads (three per CU,
// 6 sweeps); more: 256 threads (eight per CU, 11 sweeps -- the wide instantiations need more than 64 registers, so they
// only pay while every vector of the call is resident at once).  See runs_extract_launch().
__global__ __launch_bounds__(512, 4) void k_runs_extract_512(const RunsRef* __restrict__ refs, int with_len_cap, int* __restrict__ zero,
                                                             int zero_words, int cstride, int n_plain) {
    runs_extract_zero(zero, zero_words);
    const RunsRef r = refs[blockIdx.x];
    if (!r.bits) return;
    if (with_len_cap && threadIdx.x == 0) const_cast<int2*>(r.hdr)[1] = make_int2(r.len, r.cap);
    if (runs_compact_vec(cstride, n_plain))
        runs_extract_body<512, true>(r.bits, r.len, const_cast<int2*>(r.e), const_cast<int2*>(r.hdr), r.cap);
    else
        runs_extract_body<512, false>(r.bits, r.len, const_cast<int2*>(r.e), const_cast<int2*>(r.hdr), r.cap);
}
__global__ __launch_bounds__(1024, 4) void k_runs_extract_1024(const RunsRef* __restrict__ refs, int with_len_cap, int* __restrict__ zero,
                                                               int zero_words, int cstride, int n_plain) {
    runs_extract_zero(zero, zero_words);
    const RunsRef r = refs[blockIdx.x];
    if (!r.bits) return;
    if (with_len_cap && threadIdx.x == 0) const_cast<int2*>(r.hdr)[1] = make_int2(r.len, r.cap);
    if (runs_compact_vec(cstride, n_plain))
        runs_extract_body<1024, true>(r.bits, r.len, const_cast<int2*>(r.e), const_cast<int2*>(r.hdr), r.cap);
    else
        runs_extract_body<1024, false>(r.bits, r.len, const_cast<int2*>(r.e), const_cast<int2*>(r.hdr), r.cap);
}

// vectors into caller-owned list blocks (ffs_runs_from_bits_batch): the block header also gets (len, cap)
__global__ __launch_bounds__(256, 8) void k_runs_extract_lists(const RunsRef* __restrict__ refs) {
    const RunsRef r = refs[blockIdx.x];
    if (threadIdx.x == 0) const_cast<int2*>(r.hdr)[1] = make_int2(r.len, r.cap);
    runs_extract_body<256>(r.bits, r.len, const_cast<int2*>(r.e), const_cast<int2*>(r.hdr), r.cap);
}

// one vector into a caller-owned list (ffs_runs_from_bits)
__global__ __launch_bounds__(1024, 4) void k_runs_extract_one(const unsigned* __restrict__ bits, int len, int2* __restrict__ e,
                                                             int2* __restrict__ hdr, int cap) {
    if (threadIdx.x == 0) hdr[1] = make_int2(len, cap);
    runs_extract_body<1024>(bits, len, e, hdr, cap);
}

// every vector of `refs` that arrived as bits -> its list, with the workgroup size that suits the number of vectors
// (`with_len_cap`: caller-owned blocks, whose header also carries (len, cap))
// (`cstride`, `n_plain`: the plan-owned lists of a call -- candidates compact; never with caller-owned blocks)
static inline void runs_extract_launch(const RunsRef* refs, size_t n_vec, bool with_len_cap, hipStream_t st, int* zero = nullptr,
                                       int zero_words = 0, int cstride = 0, int n_plain = 0) {
    if (with_len_cap) cstride = 0;
    if (n_vec <= 256)
        hipLaunchKernelGGL(k_runs_extract_1024, dim3((unsigned)n_vec), dim3(1024), 0, st, refs, with_len_cap ? 1 : 0, zero, zero_words, cstride, n_plain);
    else if (n_vec <= 768)  // (72 registers: three 512-thread workgroups per CU)
        hipLaunchKernelGGL(k_runs_extract_512, dim3((unsigned)n_vec), dim3(512), 0, st, refs, with_len_cap ? 1 : 0, zero, zero_words, cstride, n_plain);
    else if (with_len_cap)
        hipLaunchKernelGGL(k_runs_extract_lists, dim3((unsigned)n_vec), dim3(256), 0, st, refs);
    else
        hipLaunchKernelGGL(k_runs_extract, dim3((unsigned)n_vec), dim3(256), 0, st, refs, zero, zero_words, cstride, n_plain);
}

// ---------------------------------------------------------------------------------------------------------------
// 32 bits [start, start + 32) of a bit-packed vector, zero outside [0, len)
FFS_DEV unsigned fetch32(GWords w, int len, long long start) {
    const int nw = (len + 31) >> 5;
    const long long wi = start >> 5;  // floor
    unsigned v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const long long i = wi + k;
        unsigned t = (i >= 0 && i < nw) ? w[i] : 0u;
        if (i == nw - 1 && (len & 31)) t &= (1u << (len & 31)) - 1u;
        v[k] = t;
    }
    return __builtin_amdgcn_alignbit(v[1], v[0], (unsigned)(start & 31));
}

// the same 32 samples from the vector's boundary list: the value at position x is the parity of the boundaries <= x
FFS_DEV unsigned list_bits32(GEntries e, int n, long long start) {
    if (start + 32 <= 0 || n <= 0) return 0u;
    int lo = 0, hi = n;  // boundaries at positions <= start
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)e[mid].pos <= start)
            lo = mid + 1;
        else
            hi = mid;
    }
    unsigned m = (lo & 1) ? 0xffffffffu : 0u;
    for (int k = lo; k < n; ++k) {
        const long long off = (long long)e[k].pos - start;  // >= 1
        if (off >= 32) break;
        m ^= 0xffffffffu << (int)off;
    }
    return m;
}

// first k in [0, n] with e[k].x >= x, by a whole wave (every lane passes the same arguments): 64 probes per step, three
// dependent loads for a list of 32 768 entries instead of fifteen
FFS_DEV int wave_lower_bound_e(GEntries a, int n, int x) {
    const int lane = threadIdx.x & 63;
    int lo = 0, len = n;  // the answer lies in [lo, lo + len]
    while (len > 0) {
        const int step = (len + 63) >> 6;
        const int idx = lo + (lane + 1) * step - 1;  // last element of this lane's sub-range
        const bool below = idx < lo + len && a[idx].pos < x;
        const int c = __popcll(__ballot(below));  // sub-ranges that lie entirely below x (a prefix of the lanes)
        const int end = lo + len;
        lo += c * step;
        len = (step - 1) < (end - lo) ? (step - 1) : (end - lo);
    }
    return lo;
}

// exclusive scan over the NW * 64 threads of a block of three ints at once (s_tmp: NW x 3 ints); wrap-around arithmetic
template <int NW>
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
    for (int i = 0; i < NW - 1; ++i)
        if (i < wave) pa += s_tmp[i * 3], pb += s_tmp[i * 3 + 1], pc += s_tmp[i * 3 + 2];
    a = pa, b = pb, c = pc;
}

// RUNS_FLAG_SPLIT workgroups per sub-batch (pairs_per_chunk pairs; grid = (sub-batches, RUNS_FLAG_SPLIT), flags zeroed by the
// caller): flags[chunk] = 1 when some candidate of it is over budget (the whole sub-batch then goes through the transforms
// and k_runs_corr leaves it alone), 2 when a list-only vector of it is truncated (nothing can solve it: the host reports
// the error).  stats[0] += boundaries of the sub-batch's vectors.  (One workgroup per sub-batch took 32 us for 512 x 7
// candidates -- three dependent global loads per candidate -- on the stream between the extraction and k_runs_corr.)
constexpr int RUNS_FLAG_SPLIT = 8;
// (round 6: at least RUNS_FLAG_SPLIT, and one workgroup per 448 candidates of a sub-batch -- a call solved as ONE sub-batch of
// 8192 pairs kept eight workgroups busy for 60 us)
static inline unsigned runs_flag_split(long long cands_per_chunk) {
    const long long want = (cands_per_chunk + 447) / 448;
    return (unsigned)(want < RUNS_FLAG_SPLIT ? RUNS_FLAG_SPLIT : (want > 128 ? 128 : want));
}
__global__ __launch_bounds__(256) void k_runs_chunk_flags(const CandDesc* __restrict__ cands, int n_pairs, int n_cand,
                                                          int pairs_per_chunk, const RunsRef* __restrict__ refs, long long budget,
                                                          int* __restrict__ flags, unsigned long long* __restrict__ stats,
                                                          int estimates) {
    const int ch = blockIdx.x;
    const int p0 = ch * pairs_per_chunk, p1 = (p0 + pairs_per_chunk) < n_pairs ? (p0 + pairs_per_chunk) : n_pairs;
    int over = 0, bad = 0;
    unsigned long long nb = 0;
    int longest = 0;  // longest list among the vectors that arrived as bits (plan-owned lists: the host sizes their stride by it)
    for (int i = p0 * n_cand + (int)(blockIdx.y * 256 + threadIdx.x); i < p1 * n_cand; i += 256 * (int)gridDim.y) {
        const CandDesc& cd = cands[i];
        const int pair = i / n_cand, j = i - pair * n_cand;
        const int vr = pair * (n_cand + 1), vs = vr + 1 + j;
        const RunsRef rr = refs[vr], rs = refs[vs];
        const GInts hq = (GInts)rr.hdr, hp = (GInts)rs.hdr;
        const int n_q = hq[0], n_p = hp[0];
        // (a caller's block carries its capacity in the header: n, ones, len, cap)
        // `estimates` (k_runs_probe): the headers of vectors that arrived as bits hold estimated counts, not list lengths
        const int cap_q = (estimates && rr.bits) ? 0x7fffffff : (rr.cap > 0 ? rr.cap : hq[3]);
        const int cap_p = (estimates && rs.bits) ? 0x7fffffff : (rs.cap > 0 ? rs.cap : hp[3]);
        nb += (unsigned)n_p + (j == 0 ? (unsigned)n_q : 0u);
        if (rs.bits && n_p > longest) longest = n_p;
        if (rr.bits && n_q > longest) longest = n_q;
        bad |= (!rs.bits && n_p >= cap_p) || (!rr.bits && n_q >= cap_q);
        if (cd.flags & CAND_NO_LAGS) continue;
        // (16-bit histogram cells: lists of 32 768 entries or more never take the run-boundary kernel, whoever owns them)
        const int lim_q = (estimates && rr.bits) ? cap_q : (cap_q < RUNS_CAP ? cap_q : RUNS_CAP);
        const int lim_p = (estimates && rs.bits) ? cap_p : (cap_p < RUNS_CAP ? cap_p : RUNS_CAP);
        over |= runs_over_budget(n_p, n_q, lim_p, lim_q, (long long)cd.d_hi - cd.d_lo + 1, cd.R, budget) ? 1 : 0;
    }
    over = __syncthreads_or(over);
    bad = __syncthreads_or(bad);
    __shared__ unsigned long long s_nb[4];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) nb += __shfl_xor(nb, s, 64);
    if ((threadIdx.x & 63) == 0) s_nb[threadIdx.x >> 6] = nb;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int f = bad ? 2 : over;
        if (f) atomicMax(&flags[ch], f);
        if (!estimates) atomicAdd(stats, s_nb[0] + s_nb[1] + s_nb[2] + s_nb[3]);
    }
    if (!estimates) {
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const int o = __shfl_xor(longest, s, 64);
            longest = o > longest ? o : longest;
        }
        if ((threadIdx.x & 63) == 0 && longest > 0) atomicMax(stats + 1, (unsigned long long)longest);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The candidate's record, written by the kernel itself (k_finalize_cands is for the transform path's nominee lists).  The
// pair records stay with k_finalize_pairs, one tiny launch behind the kernel: finishing a pair inside k_runs_corr -- the
// last of its candidates to arrive reads the others' records -- needs a device-scope release per workgroup, which on this
// chip (eight XCDs, one L2 each) is an L2 write-back: measured 0.35 us per pair, more than the rest of the kernel.
FFS_DEV void runs_write_cand(const CandDesc& cd, CandResult* __restrict__ cres, int ci, double score, int d, bool none) {
    CandResult r;
    if (none) {  // every lag masked: np.argmax of all -inf is k=0 (aligners.py:45-48)
        r.score = -INFINITY;
        r.offset = (long long)cd.n_ref - 1 - cd.S;
        r.score_f32 = -INFINITY;
        r.flags = 1;
    } else {
        r.score = score;
        r.offset = d;
        r.score_f32 = (float)score;
        r.flags = 0;
    }
    apply_zero_rule(cd, r);
    cres[ci] = r;
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-level references (round 5).  The `weighted` fused VAD hands the aligner 0.6 * silero + 0.4 * webrtc
// (speech_transformers.py:290-293): four levels {l, .4 + .6 l, .6 + .4 l, 1} (l = non_speech_label), a float vector.
// A vector with levels lam_0 < ... < lam_{L-1}, L <= 4, is  lam_0 + sum_k (lam_k - lam_{k-1}) [r >= lam_k]  -- L - 1
// two-level THRESHOLD vectors; when the steps are small integer multiples m_k of one quantum q (2 : 1 : 2 above), the
// integer-valued vector M = sum_k m_k [r >= lam_k] satisfies r = lam_0 + q M and its correlation with a two-level
// candidate follows from the SAME histogram, every coincidence with threshold list k added m_k-fold:
//     sum_i s'[i] (2 r[i+d] - 1) = (2 lam_0 - 1) (s0 ov + (s1 - s0) n1x) + 2 q (s0 Mx1 + (s1 - s0) M11),
// M11 = sum_i b[i] M[i+d] and Mx1 = sum over the overlap of M exact integers.  Anything else (more levels, steps that are
// not commensurable, a sample that is none of the levels) is flagged and takes the transforms.
struct LevelInfo {
    double lam[4];  // the distinct sample values, ascending
    double q;       // quantum of the level steps
    int32_t n_levels;  // 1 .. 4; 5 = more than four
    int32_t m[3];      // lam[k + 1] - lam[k] = m[k] * q
    int32_t ok;        // usable by the run-boundary path (cleared again when a sample matches no level)
    int32_t pad[3];
};
constexpr int LEVELS_MAX_MULT = 8;

template <class T>
FFS_DEV void levels_insert(double (&v)[5], int& n, T x) {  // keep up to five distinct values (the fifth means "too many")
    const double d = (double)x;
    bool seen = false;
#pragma unroll
    for (int i = 0; i < 5; ++i) seen |= (i < n && v[i] == d);
    if (!seen && n < 5) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if (i == n) v[i] = d;
        ++n;
    }
}

// One workgroup per reference vector: 2048 samples spread evenly -> the distinct values among them (a level that fills
// less than ~0.3 % of a vector may be missed: k_levels_bits then meets an unknown sample and clears `ok`).
template <class T>
__global__ __launch_bounds__(256) void k_levels_sample(const T* const* __restrict__ vec, const int* __restrict__ len, LevelInfo* __restrict__ out) {
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    typedef const __attribute__((address_space(1))) T* GT;
    const GT x = (GT)vec[v];
    const int n = len[v];
    double vals[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long long k = ((long long)(u * 256 + tid) * n) / 2048;
        if (k < n) levels_insert(vals, cnt, x[k]);
    }
    // merge: six butterfly steps leave the wave's distinct values in every lane, thread 0 merges the four waves'
    __shared__ double s_v[4][5];
    __shared__ int s_n[4];
    for (int s = 1; s < 64; s <<= 1) {
        const int cn = __shfl_xor(cnt, s, 64);
        double pv[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) pv[i] = __shfl_xor(vals[i], s, 64);
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if (i < cn) levels_insert(vals, cnt, pv[i]);
    }
    double (&wv)[5] = vals;
    const int wn = cnt;
    if (lane == 0) {
        s_n[wave] = wn;
#pragma unroll
        for (int i = 0; i < 5; ++i) s_v[wave][i] = wv[i];
    }
    __syncthreads();
    if (tid == 0) {
        double all[5];
        int an = 0;
        for (int w = 0; w < 4; ++w)
            for (int i = 0; i < s_n[w] && i < 5; ++i) levels_insert(all, an, s_v[w][i]);
        for (int i = 1; i < an && i < 5; ++i)  // ascending (insertion sort of at most five)
            for (int j = i; j > 0 && all[j] < all[j - 1]; --j) {
                const double t = all[j];
                all[j] = all[j - 1];
                all[j - 1] = t;
            }
        LevelInfo li;
        li.n_levels = an;
        li.ok = 0;
        li.q = 0.0;
        li.pad[0] = li.pad[1] = li.pad[2] = 0;
        for (int i = 0; i < 4; ++i) li.lam[i] = i < an ? all[i] : 0.0;
        for (int i = 0; i < 3; ++i) li.m[i] = 0;
        if (an >= 2 && an <= 4 && all[0] == all[0] && all[an - 1] - all[0] < 1e300) {
            double q = all[1] - all[0];
            for (int i = 2; i < an; ++i) q = (all[i] - all[i - 1]) < q ? (all[i] - all[i - 1]) : q;
            // the steps as multiples of the smallest one over 1, 2, 3, 4: 2 : 1 : 2 needs the divisor 1, 3 : 2 the divisor 2 ...
            for (int dv = 1; dv <= 4 && !li.ok; ++dv) {
                const double qq = q / dv;
                bool good = true;
                int mm[3] = {0, 0, 0};
                for (int i = 1; i < an; ++i) {
                    const double r = (all[i] - all[i - 1]) / qq, rr = rint(r);
                    good = good && fabs(r - rr) <= 1e-9 * rr && rr >= 1.0 && rr <= (double)LEVELS_MAX_MULT;
                    mm[i - 1] = (int)rr;
                }
                if (good) {
                    li.ok = 1;
                    li.q = qq;
                    for (int i = 0; i < 3; ++i) li.m[i] = mm[i];
                }
            }
        }
        out[v] = li;
    }
}

// Threshold planes of a multi-level vector: plane k (k = 0 .. 2) bit i = [x[i] >= lam[k + 1]] (all zero for k >= L - 1),
// `plane_words` 32-bit words apart.  grid = (chunks of 16 384 samples, vectors); a wave turns 64 samples into two words per
// plane with a ballot.  A sample that equals none of the levels clears info->ok (the vector then takes the transforms).
// (s_bitreplicate_b64_b32: every bit of a 32-bit scalar twice -- with the masks below two lane masks interleave into the
// sample order of lanes that hold two consecutive samples each)
FFS_DEV unsigned long long levels_interleave(unsigned even, unsigned odd) {
    unsigned long long re, ro;
    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(re) : "s"(even));
    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(ro) : "s"(odd));
    return (re & 0x5555555555555555ull) | (ro & 0xaaaaaaaaaaaaaaaaull);
}

// Threshold planes of a multi-level vector: plane k (k = 0 .. 2) bit i = [x[i] >= lam[k + 1]] (all zero for k >= L - 1),
// `plane_words` 32-bit words apart.  grid = (chunks of 16 384 samples, vectors).  A lane holds TWO consecutive samples (one
// 16-byte load for float64), a wave step covers 128: the compares' lane masks ARE the ballots, two of them interleave into
// four words per plane on the scalar unit, lanes 0-3 store them.  A sample that equals none of the levels clears
// info->ok (the vector then takes the transforms).
template <class T>
__global__ __launch_bounds__(256) void k_levels_bits(const T* const* __restrict__ vec, const int* __restrict__ len,
                                                     LevelInfo* __restrict__ info, unsigned* const* __restrict__ planes,
                                                     const int* __restrict__ plane_words) {
    const int v = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = len[v];
    const long long c0 = (long long)blockIdx.x * 16384;
    if (c0 >= n) return;
    const LevelInfo li = info[v];
    if (!li.ok) return;
    typedef T T2 __attribute__((ext_vector_type(2), aligned(sizeof(T))));
    typedef const __attribute__((address_space(1))) T2* GT2;
    typedef const __attribute__((address_space(1))) T* GT;
    const GT x = (GT)vec[v];
    __attribute__((address_space(1))) unsigned* out = (__attribute__((address_space(1))) unsigned*)planes[v];
    const int pw = plane_words[v];
    const double nan = __builtin_nan("");
    const double l0 = li.lam[0], l1 = li.lam[1], l2 = li.n_levels > 2 ? li.lam[2] : nan, l3 = li.n_levels > 3 ? li.lam[3] : nan;
    unsigned long long unknown = 0;  // lane mask: a sample of this lane matched no level
    constexpr int LU = 8;  // wave steps whose loads are in flight together (4 and 16: the same time)
    for (int it = 0; it < 128; it += 4 * LU) {
        const long long base0 = c0 + (long long)(it + wave * LU) * 128;  // this wave's LU consecutive steps
        if (base0 >= n) break;
        T2 xv[LU];
        if (base0 + (long long)LU * 128 <= n) {  // (wave-uniform) all of it inside the vector: plain loads, all in flight
#pragma unroll
            for (int u = 0; u < LU; ++u) xv[u] = __builtin_nontemporal_load((GT2)(x + base0 + u * 128 + 2 * lane));
        } else {
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const long long i = base0 + u * 128 + 2 * lane;
                xv[u].x = i < n ? x[i] : (T)l0;
                xv[u].y = i + 1 < n ? x[i + 1] : (T)l0;
            }
        }
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const long long base = base0 + u * 128;
            const double da = (double)xv[u].x, db = (double)xv[u].y;
            unknown |= ~(__ballot(da == l0 || da == l1 || da == l2 || da == l3) & __ballot(db == l0 || db == l1 || db == l2 || db == l3));
            const double lk[3] = {l1, l2, l3};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const unsigned long long ea = __ballot(da >= lk[k]), eb = __ballot(db >= lk[k]);
                const unsigned long long w01 = levels_interleave((unsigned)ea, (unsigned)eb);
                const unsigned long long w23 = levels_interleave((unsigned)(ea >> 32), (unsigned)(eb >> 32));
                const long long w = (base >> 5) + lane;
                if (lane < 4 && w < pw) {  // (pw = words of the vector: nothing behind its end)
                    const unsigned long long ww = (lane & 2) ? w23 : w01;
                    out[(size_t)k * pw + w] = (lane & 1) ? (unsigned)(ww >> 32) : (unsigned)ww;
                }
            }
        }
    }
    if (__syncthreads_or(unknown != 0 ? 1 : 0) && tid == 0) info[v].ok = 0;
}

// inclusive prefix sum over the 64 lanes of three ints at once, DPP only (no LDS round trips)
FFS_DEV void wave_incl_scan3(int& a, int& b, int& c) {
    a = (int)wave_incl_scan_u32((unsigned)a);
    b = (int)wave_incl_scan_u32((unsigned)b);
    c = (int)wave_incl_scan_u32((unsigned)c);
}

// exclusive scan over the NW * 64 threads of a block of three ints at once (s_tmp: NW x 3 ints); wrap-around arithmetic;
// ONE barrier (the caller alternates between two s_tmp buffers, so a second scan cannot overwrite totals still being read)
template <int NW>
FFS_DEV void block_excl_scan3_dpp(int& a, int& b, int& c, int* s_tmp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ia = a, ib = b, ic = c;
    wave_incl_scan3(ia, ib, ic);
    if (lane == 63) s_tmp[wave * 3] = ia, s_tmp[wave * 3 + 1] = ib, s_tmp[wave * 3 + 2] = ic;
    __syncthreads();
    int pa = ia - a, pb = ib - b, pc = ic - c;
#pragma unroll
    for (int i = 0; i < NW - 1; ++i)
        if (i < wave) pa += s_tmp[i * 3], pb += s_tmp[i * 3 + 1], pc += s_tmp[i * 3 + 2];
    a = pa, b = pb, c = pc;
}

// 32 samples [start, start + 32) of a vector from (a stretch of) its boundary list held in LDS: ent[0 .. cnt) = the
// positions of the list's entries k0 .. k0 + cnt - 1 (ascending); every entry of the list in [start, start + 32) must
// be among them, and entries in front of k0 lie in front of `start`.
FFS_DEV unsigned lds_list_bits32(const int* ent, int k0, int cnt, long long start) {
    if (start + 32 <= 0) return 0u;
    int lo = 0, hi = cnt;  // staged entries at positions <= start
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)ent[mid] <= start)
            lo = mid + 1;
        else
            hi = mid;
    }
    unsigned m = ((k0 + lo) & 1) ? 0xffffffffu : 0u;
    for (int k = lo; k < cnt; ++k) {
        const long long off = (long long)ent[k] - start;  // >= 1
        if (off >= 32) break;
        m ^= 0xffffffffu << (int)off;
    }
    return m;
}

// grid = (pairs * split, tiles_max); block = 512 threads, four workgroups per CU (<= 64 VGPRs, 39 KB of LDS).  Workgroup
// (pair, j0) solves the candidates j0, j0 + split, ... of its pair one after the other (split = n_cand: one candidate per
// workgroup; split = 1: the whole pair) -- the reference's boundary list is staged in LDS ONCE for all of them; tile t of
// a candidate covers the lags [d_lo + t*RUNS_T, ...] of its window.  Per candidate:
//   0. the four 32-sample windows of the two vectors a thread needs for the one-sided counts (the samples that enter /
//      leave the overlap at its RUNS_LPT lags) are requested first -- from the bits, or, for a vector that exists as a
//      list only, from the stretch of its list that the tile's lags can touch (four waves stage one stretch each);
//   1. zero the tile's second-difference array (16 bits per lag in 32-bit LDS words: the sum of all additions to a
//      word is v_lo + 65536 * v_hi as an integer, so both halves are recovered exactly whatever the borrows did); the
//      reference's list sits in LDS with its positions DOUBLED (lag differences then come out doubled: the histogram
//      word's byte address is d2 & ~3 and the 16-bit half (d2 & 2) << 3 -- one instruction each);
//   2. ONE candidate boundary p per lane, 64 consecutive ones per wave task, RUNS_TPW tasks of a wave advanced together,
//      UNIFORM CONTROL FLOW (round 6): a binary search of a fixed number of steps gives the first boundary q >= p + D0 AND
//      (through the list's ones-in-front column) the ones of the reference in front of p + D0 -- summed over p that is
//      n11(D0), and g(D0) is the sum of the parities -- then every lane walks the q's two at a time (an aligned 8-byte LDS
//      read = one run of the reference: start +, end -) and adds +-1 at lag q - p when that lag lies in the tile
//      (ds_add_u32, fire and forget); a lane that has passed its window keeps reading (its lags are out of range, its
//      list index stops at the sentinel pair) until no lane of the wave's tasks has anything left -- no per-task masks,
//      no per-lane branches: the round-5 loop spent two scalar instructions per vector instruction on exec-mask
//      bookkeeping (98 k SALU per pair).  An LDS atomic to random words costs ~7.4 cycles per wave-instruction (bank
//      conflicts; 4.1 conflict-free: profiles/lds_issue_rates.hip), so what matters is that the lanes of a wave have
//      windows of similar length -- consecutive p's do;
//   3. ONE block scan (DPP inside a wave, one barrier) turns h into g and n11 at every thread's first lag: with
//      hs = sum of the thread's h, ws = sum_k (LPT - k) h[k] and the thread index t,
//          g_c = g_0 - A,   n11_c = n11_0 + LPT t g_0 - LPT ((t - 1) A - B) - C,
//      A / B / C the exclusive prefixes of hs / t hs / ws (round 5: two dependent scans and two passes over h); the
//      one-sided counts follow from their values at D0 and the windows of step 0 (same scan);
//   4. every lag is scored in fp32 (4 fused multiply-adds, branch-free); lags within the rounding-error bound of the
//      block's best fp32 value are re-evaluated with exact_score()'s fp64 expression; block argmax over those, ties to the
//      largest lag;
//   5. the winning thread writes the candidate's record.
// (FFS_RUNS_STOP=n, never defined in the product build: section stop points for the A/B harness profiles/runs_ab.sh --
// the kernel returns early with WRONG results so that the sections' shares of its time can be read off.)
constexpr int RUNS_EDGE = 96;  // list entries staged per edge window stretch (denser: read from global memory)
#ifndef FFS_RUNS_WPS
#define FFS_RUNS_WPS 8
#endif
constexpr int RUNS_X_NONE = -(1 << 26);  // doubled position of a lane without a run: far in front of every list (every lag of its walk
                                         // lies beyond the 2 * RUNS_T doubled lags of a tile; vectors are shorter than 2^24 samples on this path)

// reference boundaries as the walk reads them, positions DOUBLED: staged in LDS (pointer already moved back by the slice's
// first index) ...
struct QLds2 {
    const int* p;
    FFS_DEV int one(int k) const { return p[k]; }
    FFS_DEV int2 two(int k) const { return *reinterpret_cast<const int2*>(p + k); }  // k even
};
// ... or straight from the list in global memory (a reference far denser than the candidate)
struct QGlb2 {
    GEntries e;
    int n;
    FFS_DEV int one(int k) const {
        const int v = k <= n ? e[k].pos : RUNS_QSENT;
        return 2 * (v < RUNS_QSENT ? v : RUNS_QSENT);
    }
    FFS_DEV int2 two(int k) const { return make_int2(one(k), one(k + 1)); }
};

// exclusive scan over the NW * 64 threads of a block of five ints at once (s_tmp: NW x 8 ints); wrap-around arithmetic; ONE
// barrier (the caller alternates between two s_tmp buffers, so a second scan cannot overwrite totals still being read)
template <int NW>
FFS_DEV void block_excl_scan5_dpp(int (&v)[5], int* s_tmp, int lane, int wave) {  // (wave: uniform, in a scalar register)
    int in[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) in[k] = (int)wave_incl_scan_u32((unsigned)v[k]);
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 5; ++k) s_tmp[wave * 8 + k] = in[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = in[k] - v[k];
#pragma unroll
    for (int i = 0; i < NW - 1; ++i)
        if (i < wave) {
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] += s_tmp[i * 8 + k];
        }
}

// ML: the reference is a multi-level vector given as up to three threshold lists refs[n_vec_all + 3 pair + k] with the
// multiplicities of linfo[pair] (see LevelInfo); the levels are processed one after the other into the same histogram.
template <bool ML>
FFS_DEV void runs_corr_body(const CandDesc* __restrict__ cands, int n_cand, const RunsRef* __restrict__ refs,
                            CandResult* __restrict__ cres, RunsBest* __restrict__ best, int tiles_max,
                            const int* __restrict__ chunk_flags, int pairs_per_chunk, const LevelInfo* __restrict__ linfo,
                            int n_vec_all, int split) {
    __shared__ __attribute__((aligned(16))) unsigned hist[RUNS_T / 2 + 4];  // second difference h of the tile's lags, 16 bits per lag
    __shared__ __attribute__((aligned(16))) int q_lds[RUNS_QCAP + 4];       // DOUBLED positions
    __shared__ int s_edge[4][RUNS_EDGE];
    __shared__ int s_ek0[4], s_ecnt[4];
    __shared__ __attribute__((aligned(16))) int s_tmp[2][RUNS_WAVES * 8];
    __shared__ double s_sc[RUNS_WAVES];
    __shared__ int s_d[RUNS_WAVES];
    __shared__ float s_m[RUNS_WAVES];
    const int tile = blockIdx.y, tid0 = threadIdx.x;
    // workgroup b runs on XCD b % 8 (one L2 each): every XCD takes a contiguous stretch of the (pair, candidate) indices, so
    // the workgroups of one pair -- which all stage the same reference list -- follow each other on ONE XCD
    int bx;
    {
        const int nb = (int)gridDim.x, x = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3, base = nb >> 3, rem = nb & 7;
        bx = x * base + (x < rem ? x : rem) + i;  // (-1.3 % at 8192 pairs: the six other stagings of a list hit the L2)
    }
    const int pair = bx / split, j_first = bx - pair * split;
    if (chunk_flags[pair / pairs_per_chunk]) return;  // this sub-batch goes through the transforms (k_runs_chunk_flags)
    const int vr = pair * (n_cand + 1);
    const int vr0 = ML ? n_vec_all + 3 * pair : vr;  // (first) list of the reference
    RunsRef rr = refs[vr0];
    // (the LevelInfo fields as scalars, the multiplicity of level k by selects: a struct copy indexed with a loop variable
    // lives in scratch memory -- round 5's 80 bytes per lane)
    double lev_lam0 = 0.0, lev_q = 0.0;
    int n_lv = 1, lev_m0 = 0, lev_m1 = 0, lev_m2 = 0;
    if (ML) {
        const LevelInfo* lp = linfo + pair;
        lev_lam0 = lp->lam[0], lev_q = lp->q;
        n_lv = lp->n_levels - 1;
        lev_m0 = lp->m[0], lev_m1 = lp->m[1], lev_m2 = lp->m[2];
    }
    auto lev_m = [&](int k) -> int { return k == 0 ? lev_m0 : (k == 1 ? lev_m1 : lev_m2); };
    GEntries Qe = (GEntries)rr.e;
    GWords rbits = (GWords)rr.bits;
    int n_q = ((GInts)rr.hdr)[0];
    // the reference's whole list in LDS, once for every candidate of this workgroup (ML: one list per level and candidate)
    const bool pair_staged = !ML && n_q <= RUNS_QCAP;
    auto stage_whole = [&](int tid) {
        constexpr int NST = (RUNS_QCAP + 2 + RUNS_THREADS - 1) / RUNS_THREADS;
        int qv[NST];
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int k = tid + u * RUNS_THREADS;
            qv[u] = k <= n_q ? Qe[k].pos : RUNS_QSENT;
        }
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int k = tid + u * RUNS_THREADS;
            if (k <= n_q + 1) q_lds[k] = 2 * (qv[u] < RUNS_QSENT ? qv[u] : RUNS_QSENT);
        }
    };
    if (pair_staged) stage_whole(tid0);  // (the first candidate's barrier behind the zeroing covers it)
    bool first_cand = true;
    for (int jc = j_first; jc < n_cand; jc += split) {
    // (the thread index is made opaque once per candidate: everything derived from it -- LDS addresses, lane masks of the
    // wave-prefix sums -- would otherwise be hoisted out of this loop and spilled: 38 VGPRs + 48 SGPRs)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci = pair * n_cand + jc, vs = vr + 1 + jc;
    const RunsRef rs_ = refs[vs];
    const CandDesc cd = cands[ci];
    if (cd.flags & CAND_NO_LAGS) {
        if (tile == 0 && tid == 0) {
            runs_write_cand(cd, cres, ci, 0.0, 0, true);
        }
        continue;
    }
    const int W = cd.d_hi - cd.d_lo + 1;
    const int n_tiles = (W + RUNS_T - 1) / RUNS_T;
    if (tile >= n_tiles) continue;
    const int S = cd.S, R = cd.R;
    const int D0 = cd.d_lo + tile * RUNS_T;
    const int Wt = (cd.d_hi - D0 + 1) < RUNS_T ? (cd.d_hi - D0 + 1) : RUNS_T;
    const GEntries Pe = (GEntries)rs_.e;
    const GWords sbits = (GWords)rs_.bits;
    // position of the candidate's boundary k = Pc[k << p_sh]: a plan-owned candidate list holds positions only (RunsRef),
    // every other list (position, ones in front) -- one code path, no branch
    // (calls with multi-level references keep full candidate lists: k_runs_corr_ml sits at its register budget)
    const int p_sh = (!ML && rs_.bits != nullptr && rs_.cap > 0) ? 0 : 1;
    const GInts Pc = (GInts)rs_.e;
    const int n_p = ((GInts)rs_.hdr)[0];
    const int c = tid * RUNS_LPT;  // this thread's lags: D0 + c .. D0 + c + RUNS_LPT - 1
    // the samples that enter (+) and leave (-) the two one-sided counts when the lag grows by one, lag c + i = bit i:
    //   n1x(d+1) = n1x(d) + b[-d-1] - b[R-d-1],   nx1(d+1) = nx1(d) - rho[d] + rho[S+d]   (zero outside the vectors)
    const long long dc = (long long)D0 + c;
    unsigned m_in1x = 0, m_out1x = 0, m_outx1 = 0, m_inx1 = 0;
    if (sbits) {
        m_in1x = __brev(fetch32(sbits, S, -dc - 32));
        m_out1x = __brev(fetch32(sbits, S, (long long)R - dc - 32));
    }
    // ML: per group of four lags, sum over the threshold planes of m_k (samples entering - leaving nx1), |value| <= 4 * 3 *
    // LEVELS_MAX_MULT = 96: one signed BYTE per group, packed in two registers (round 5 kept six ints, in scratch memory)
    static_assert(RUNS_LPT / 4 <= 8 && 4 * 3 * LEVELS_MAX_MULT < 128, "one signed byte per group of four lags");
    unsigned long long wx1p = 0ull;
    int wx1_sum = 0;
    auto wx1_of = [&](int g4) -> int {  // sign-extended byte g4 (g4 may be a loop variable: two 64-bit shifts, no array)
        return (int)((long long)(wx1p << (56 - 8 * g4)) >> 56);
    };
    if (ML) {
        int acc[RUNS_LPT / 4];
#pragma unroll
        for (int g4 = 0; g4 < RUNS_LPT / 4; ++g4) acc[g4] = 0;
        for (int k = 0; k < n_lv; ++k) {
            const GWords pb = (GWords)refs[vr0 + k].bits;
            const unsigned mo = fetch32(pb, R, dc), mi = fetch32(pb, R, (long long)S + dc);
            const int mk = lev_m(k);
#pragma unroll
            for (int g4 = 0; g4 < RUNS_LPT / 4; ++g4)
                acc[g4] += mk * (__popc((mi >> (4 * g4)) & 15u) - __popc((mo >> (4 * g4)) & 15u));
        }
#pragma unroll
        for (int g4 = 0; g4 < RUNS_LPT / 4; ++g4) {
            wx1p |= (unsigned long long)((unsigned)acc[g4] & 0xffu) << (8 * g4);
            wx1_sum += acc[g4];
        }
        if (!sbits) {  // a list-only candidate against a multi-level reference: its windows straight from the list
            m_in1x = __brev(list_bits32(Pe, n_p, -dc - 32));
            m_out1x = __brev(list_bits32(Pe, n_p, (long long)R - dc - 32));
        }
    } else if (rbits) {
        m_outx1 = fetch32(rbits, R, dc);
        m_inx1 = fetch32(rbits, R, (long long)S + dc);
    }
    if (!first_cand) __syncthreads();  // the previous candidate's readers of hist / s_edge / s_tmp / s_sc are done
    first_cand = false;
    if (!ML && (!sbits || !rbits) && wave < 4) {
        // list-only vectors: wave w stages the stretch of the list that window w of ANY thread of the tile can touch
        // (windows: 0 b[-d-1], 1 b[R-d-1], 2 rho[d], 3 rho[S+d] over the tile's lags D0 .. D0 + RUNS_T - 1)
        const bool of_b = wave < 2;
        if (of_b ? !sbits : !rbits) {
            const GEntries E = of_b ? Pe : Qe;
            const int n_e = of_b ? n_p : n_q;
            const long long t0 = (long long)D0, t1 = (long long)D0 + RUNS_T - 1;  // lags of the tile
            const long long lo_pos = wave == 0 ? -t1 - 32 : wave == 1 ? (long long)R - t1 - 32 : wave == 2 ? t0 : (long long)S + t0;
            const long long hi_pos = wave == 0 ? -t0 : wave == 1 ? (long long)R - t0 : wave == 2 ? t1 + 32 : (long long)S + t1 + 32;
            const int clo = lo_pos < 0 ? 0 : (lo_pos > 0x3fffffff ? 0x3fffffff : (int)lo_pos);
            const int chi = hi_pos < 0 ? 0 : (hi_pos > 0x3fffffff ? 0x3fffffff : (int)hi_pos);
            const int k0 = wave_lower_bound_e(E, n_e, clo), k1 = wave_lower_bound_e(E, n_e, chi + 1);
            const int cnt = k1 - k0;
            for (int k = lane; k < cnt && k < RUNS_EDGE; k += 64) s_edge[wave][k] = E[k0 + k].pos;
            if (lane == 0) s_ek0[wave] = k0, s_ecnt[wave] = cnt;
        }
    }
    {
        uint4* hz = reinterpret_cast<uint4*>(hist);
        unsigned z = 0;
        asm volatile("" : "+v"(z));  // (a zero the compiler does not keep alive -- and spill -- across the whole candidate)
        for (int i = tid; i < (RUNS_T / 2 + 4) / 4; i += RUNS_THREADS) hz[i] = make_uint4(z, z, z, z);
    }
    const int i0 = D0 < 0 ? -D0 : 0, i1 = (R - D0) < S ? (R - D0) : S;
    const int wmax = Wt - 2;                      // h is needed for the lags D0 .. D1 - 1
    const int wlim = wmax >= 0 ? wmax : 0;        // (a one-lag tile never reads h: a stray add at 0 is harmless)
    const unsigned wlim2 = 2u * (unsigned)wlim;   // (positions are doubled in the walk)
    int n11p = 0, gp = 0, bsum = 0, rsum_all = 0;
    bool any_sliced = false;
    for (int lv = 0; lv < n_lv; ++lv) {  // (one pass unless ML)
    int wk = 1;  // multiplicity of this level's coincidences
    if (ML) {
        if (lv > 0) {
            __syncthreads();  // every wave is done with the previous level's staged list
            rr = refs[vr0 + lv];
            Qe = (GEntries)rr.e;
            rbits = (GWords)rr.bits;
            n_q = ((GInts)rr.hdr)[0];
        }
        wk = lev_m(lv);
    }
    const bool whole = n_q <= RUNS_QCAP;  // the reference's whole list fits the staging area
    any_sliced |= !whole;
    if (whole && !pair_staged) stage_whole(tid);
    __syncthreads();  // hist is zero, the list is staged
    // ones of rho in [0, x) = sum_k sgn_k * min(x, Q[k]) (sgn = -1 at run starts, +1 at run ends): the two positions
    // that bound the overlap at lag D0
    const int r_lo = D0 > 0 ? D0 : 0, r_hi = (S + D0) < R ? (S + D0) : R;
    int rsum = 0;
    if (whole) {
        for (int k = tid; k < n_q; k += RUNS_THREADS) {
            const int q1 = q_lds[k] >> 1;
            const int m_hi = q1 < r_hi ? q1 : r_hi, m_lo = q1 < r_lo ? q1 : r_lo;
            rsum += (k & 1) ? (m_hi - m_lo) : (m_lo - m_hi);
        }
    } else {
        for (int k = tid; k < n_q; k += RUNS_THREADS) {
            const int q1 = Qe[k].pos;
            const int m_hi = q1 < r_hi ? q1 : r_hi, m_lo = q1 < r_lo ? q1 : r_lo;
            rsum += (k & 1) ? (m_hi - m_lo) : (m_lo - m_hi);
        }
    }
    rsum_all += wk * rsum;
#if defined(FFS_RUNS_STOP) && FFS_RUNS_STOP == 1
    if (n_q >= 0) return;
#endif
    if (!ML && !sbits) {
        if (s_ecnt[0] <= RUNS_EDGE && s_ecnt[1] <= RUNS_EDGE) {
            m_in1x = __brev(lds_list_bits32(s_edge[0], s_ek0[0], s_ecnt[0], -dc - 32));
            m_out1x = __brev(lds_list_bits32(s_edge[1], s_ek0[1], s_ecnt[1], (long long)R - dc - 32));
        } else {
            m_in1x = __brev(list_bits32(Pe, n_p, -dc - 32));
            m_out1x = __brev(list_bits32(Pe, n_p, (long long)R - dc - 32));
        }
    }
    if (!ML && !rbits) {
        if (s_ecnt[2] <= RUNS_EDGE && s_ecnt[3] <= RUNS_EDGE) {
            m_outx1 = lds_list_bits32(s_edge[2], s_ek0[2], s_ecnt[2], dc);
            m_inx1 = lds_list_bits32(s_edge[3], s_ek0[3], s_ecnt[3], (long long)S + dc);
        } else {
            m_outx1 = list_bits32(Qe, n_q, dc);
            m_inx1 = list_bits32(Qe, n_q, (long long)S + dc);
        }
    }
    // RUNS_TPW wave tasks at a time -- 64 consecutive candidate RUNS each, one run (start boundary, end boundary) per lane
    // (round 6; round 5: one boundary per lane), task t of a wave = runs r0 + (t * RUNS_WAVES + wave) * 64 + lane -- advanced
    // TOGETHER: every step of the binary searches and of the walks issues the tasks' LDS reads back to back and waits once.
    // A lane reads every reference run of its window ONCE (one aligned 8-byte LDS read) and adds the run's four boundary
    // coincidences: + at (start, start) and (end, end), - at (start, end) and (end, start) -- half the list reads, half
    // the searches and half the loop trips of the boundary-per-lane walk for the same scatter-adds.
    // [lo, hi] = stretch of the reference's list readable through Q (Q(hi) lies beyond every window of the round), jmax =
    // even index of a readable pair beyond every window.  Doubled positions throughout (x2 = 2 (p + D0)); a lane without a
    // run carries RUNS_X_NONE, a position far in front of the list: its search ends at `lo`, every lag of its walk is far
    // beyond the tile, nothing keeps the loops alive.
    auto tasks = [&](auto Q, int r0, int r1, int lo, int hi, int jmax) {
        int xs2[RUNS_TPW], xe2[RUNS_TPW], j[RUNS_TPW], ones_d[RUNS_TPW], a[RUNS_TPW], b[RUNS_TPW];
        typedef int v4i __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) v4i* GRunPairs;
#pragma unroll
        for (int t = 0; t < RUNS_TPW; ++t) {
            const int i = r0 + (t * RUNS_WAVES + wave) * 64 + lane;
            if (ML) {
                v4i se = {0, 0, 0, 0};
                if (i < r1) se = *(GRunPairs)(Pe + 2 * i);  // (start, ones in front, end, ones in front): one 16-byte load
                xs2[t] = se.x, xe2[t] = se.z;
            } else {
                xs2[t] = xe2[t] = 0;
                if (i < r1) xs2[t] = Pc[(2 * i) << p_sh], xe2[t] = Pc[(2 * i + 1) << p_sh];  // (start, end) of the lane's run
            }
        }
#pragma unroll
        for (int t = 0; t < RUNS_TPW; ++t) {
            const int i = r0 + (t * RUNS_WAVES + wave) * 64 + lane;
            const bool valid = i < r1;
            if (!ML || lv == 0) {  // (the candidate's own count: once, without the multiplicity)
                // ones of b in [i0, i1) = sum over its runs of (min(i1, end) - min(i0, end)) - (min(i1, start) - min(i0, start))
                const int ps = xs2[t], pe = xe2[t];
                const int ds = (ps < i1 ? ps : i1) - (ps < i0 ? ps : i0), de = (pe < i1 ? pe : i1) - (pe < i0 ? pe : i0);
                if (valid) bsum += de - ds;
            }
            xs2[t] = valid ? 2 * (xs2[t] + D0) : RUNS_X_NONE;
            xe2[t] = valid ? 2 * (xe2[t] + D0) : RUNS_X_NONE;
            a[t] = lo, b[t] = hi;  // first k in [lo, hi] with Q(k) >= xs
        }
        const int steps = hi > lo ? 32 - __clz(hi - lo) : 0;  // (uniform: no per-lane exit test)
        for (int s_ = 0; s_ < steps; ++s_) {
            int m[RUNS_TPW], qm[RUNS_TPW];
#pragma unroll
            for (int t = 0; t < RUNS_TPW; ++t) {
                m[t] = (a[t] + b[t]) >> 1;
                qm[t] = Q.one(m[t]);
            }
#pragma unroll
            for (int t = 0; t < RUNS_TPW; ++t) {
                const bool lt = qm[t] < xs2[t];
                a[t] = lt ? m[t] + 1 : a[t];
                b[t] = lt ? b[t] : m[t];
            }
        }
        // first k >= lb(start) with Q(k) >= xe: a few entries further (a candidate run rarely spans many reference runs)
#pragma unroll
        for (int t = 0; t < RUNS_TPW; ++t) b[t] = a[t];
        for (int guard = hi - lo + 1; guard > 0; --guard) {
            bool any = false;
#pragma unroll
            for (int t = 0; t < RUNS_TPW; ++t) {
                const bool adv = b[t] < hi && Q.one(b[t]) < xe2[t];
                b[t] += adv ? 1 : 0;
                any |= adv;
            }
            if (!__any(any)) break;
        }
#pragma unroll
        for (int t = 0; t < RUNS_TPW; ++t) {
            const int i = r0 + (t * RUNS_WAVES + wave) * 64 + lane;
            const bool valid = i < r1;
            const int ls = a[t], le = b[t];
            // ones of the reference in front of xe minus in front of xs (the sentinel entry holds all ones); the loads are
            // consumed after the walks
            ones_d[t] = valid ? Qe[le].ones - Qe[ls].ones : 0;
            // inside a run: the run's ones from x on are not in front of x
            const int cs = (ls & 1) ? (Q.one(ls) - xs2[t]) >> 1 : 0, ce = (le & 1) ? (Q.one(le) - xe2[t]) >> 1 : 0;
            if (valid) {
                n11p += wk * (cs - ce);           // sum over boundaries of db[p] (corr - ones), db = +1 at the start, -1 at the end
                gp -= wk * ((ls & 1) - (le & 1));
            }
            j[t] = ls & ~1;  // the reference's run that contains or follows xs: (start, end) = entries (j, j + 1)
            j[t] = j[t] < jmax ? j[t] : jmax;
        }
        const unsigned vp = (unsigned)wk, vm = (unsigned)(-wk);
        // (every trip moves each lane two entries on until it sits on the pair at jmax: at most (jmax - lo) / 2 + 1 trips do
        // anything -- the bound only guards against a kernel that cannot end)
        for (int trips = ((jmax - (lo & ~1)) >> 1) + 2; trips > 0; --trips) {
            int2 qq[RUNS_TPW];
#pragma unroll
            for (int t = 0; t < RUNS_TPW; ++t) qq[t] = Q.two(j[t]);
            bool any = false;
            int lag[RUNS_TPW][4];  // doubled lags of the four coincidences: (start, start), (end q, start p), (start q, end p), (end, end)
#pragma unroll
            for (int t = 0; t < RUNS_TPW; ++t) {
                lag[t][0] = qq[t].x - xs2[t], lag[t][1] = qq[t].y - xs2[t], lag[t][2] = qq[t].x - xe2[t], lag[t][3] = qq[t].y - xe2[t];
                // the smallest of the four lags (reference start - candidate end) beyond the tile: so is everything behind it
                any |= lag[t][2] <= (int)wlim2;
                const int jn = j[t] + 2;
                j[t] = jn < jmax ? jn : jmax;
            }
            // (the scatter-adds behind every subtraction: nothing in this trip waits for an atomic to complete)
#pragma unroll
            for (int t = 0; t < RUNS_TPW; ++t) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    // lag d = bits 16 (d & 1) .. of word d >> 1: byte address d2 & ~3, shift (d2 & 2) << 3 = the low five bits of d2 << 3
                    const unsigned d2 = (unsigned)lag[t][u];
                    if (d2 <= wlim2) atomicAdd(&hist[d2 >> 2], ((u == 0 || u == 3) ? vp : vm) << ((d2 << 3) & 31u));
                }
            }
            if (!__any(any)) break;
        }
#pragma unroll
        for (int t = 0; t < RUNS_TPW; ++t) n11p += wk * ones_d[t];  // - sum db[p] ones(p): + at the end boundary, - at the start
    };
    // Rounds of RUNS_ROUND consecutive candidate runs.  A reference list that does not fit the staging area is staged
    // slice by slice: the q's a round can meet are one stretch [first q >= first start + D0, first q beyond last end + D1],
    // found by two wave-wide 64-ary searches; only a round whose stretch is still too long (a reference far denser than
    // the candidate) walks the list in global memory.
    constexpr int RUNS_ROUND = RUNS_THREADS * RUNS_TPW;
    const int n_runs = n_p >> 1;  // (n_p is even: a run that reaches the end closes at position len)
    int st_lo = 0, st_hi = n_q;
    bool staged_ok = whole;
    for (int r0 = 0; r0 < n_runs; r0 += RUNS_ROUND) {
        const int r1 = (r0 + RUNS_ROUND) < n_runs ? (r0 + RUNS_ROUND) : n_runs;
        if (!whole) {
            __syncthreads();  // the previous round is done with the staged slice
            if (tid < 128) {
                int xq;
                if (ML) {
                    xq = (tid < 64) ? Pe[2 * r0].pos + D0 : Pe[2 * r1 - 1].pos + D0 + wlim + 1;
                } else {
                    const int kq = (tid < 64) ? 2 * r0 : 2 * r1 - 1;
                    xq = Pc[kq << p_sh] + ((tid < 64) ? D0 : D0 + wlim + 1);
                }
                const int v = wave_lower_bound_e(Qe, n_q, xq);
                if ((tid & 63) == 0) s_tmp[0][tid >> 6] = v;
            }
            __syncthreads();
            st_lo = s_tmp[0][0] & ~1, st_hi = s_tmp[0][1];
            staged_ok = st_hi - st_lo <= RUNS_QCAP;
            if (staged_ok)
                for (int k = st_lo + tid; k <= st_hi + 2; k += RUNS_THREADS) {  // (two entries beyond: the pair at jmax)
                    const int qv = k <= n_q ? Qe[k].pos : RUNS_QSENT;
                    q_lds[k - st_lo] = 2 * (qv < RUNS_QSENT ? qv : RUNS_QSENT);
                }
            __syncthreads();
        }
        if (staged_ok)
            tasks(QLds2{q_lds - st_lo}, r0, r1, st_lo, st_hi, whole ? n_q : ((st_hi + 1) & ~1));
        else
            tasks(QGlb2{Qe, n_q}, r0, r1, 0, n_q, n_q);
    }
    }  // levels
    int rsum = rsum_all;
#if defined(FFS_RUNS_STOP) && FFS_RUNS_STOP == 2
    if (n_q >= 0) return;
#endif
    // block sums of (n11p, gp, bsum, rsum)
    {
        int z = 0;
        wave_incl_scan3(n11p, gp, bsum);
        wave_incl_scan3(rsum, z, z);
    }
    if (any_sliced) __syncthreads();  // (s_tmp[0] held the slice bounds)
    if (lane == 63) s_tmp[0][wave * 8] = n11p, s_tmp[0][wave * 8 + 1] = gp, s_tmp[0][wave * 8 + 2] = bsum, s_tmp[0][wave * 8 + 3] = rsum;
    __syncthreads();  // also: every addition to hist has landed
    int n11_0 = 0, g_0 = 0, n1x_0 = 0, nx1_0 = 0;
#pragma unroll
    for (int wv = 0; wv < RUNS_WAVES; ++wv)
        n11_0 += s_tmp[0][wv * 8], g_0 += s_tmp[0][wv * 8 + 1], n1x_0 += s_tmp[0][wv * 8 + 2], nx1_0 += s_tmp[0][wv * 8 + 3];

    // The thread's lags are walked four at a time (one aligned 8-byte LDS read = two words = four lags) in ROLLED loops:
    // unrolled, the passes below keep dozens of unpacked values alive and the kernel spills at 64 registers.
    const uint2* hq = reinterpret_cast<const uint2*>(hist + (c >> 1));
    auto unpack4 = [&](int q4, int (&h)[4]) {
        const uint2 w = hq[q4];
        const int a0 = (int)(short)(w.x & 0xffffu), b0 = (int)(short)(w.y & 0xffffu);
        h[0] = a0, h[1] = (int)(w.x - (unsigned)a0) >> 16, h[2] = b0, h[3] = (int)(w.y - (unsigned)b0) >> 16;
    };
    int hs = 0, ws = 0;  // sum of the thread's h; sum_k (LPT - k) h[k]
#pragma unroll 1
    for (int q4 = 0; q4 < RUNS_LPT / 4; ++q4) {
        int h[4];
        unpack4(q4, h);
        const int s4 = (h[0] + h[1]) + (h[2] + h[3]);
        hs += s4;
        ws += (RUNS_LPT - 4 * q4) * s4 - (h[1] + 2 * h[2] + 3 * h[3]);
    }
    constexpr unsigned lpt_mask = RUNS_LPT == 32 ? 0xffffffffu : ((1u << RUNS_LPT) - 1u);
    int sc5[5];
    sc5[0] = hs, sc5[1] = tid * hs, sc5[2] = ws;
    sc5[3] = __popc(m_in1x & lpt_mask) - __popc(m_out1x & lpt_mask);
    sc5[4] = __popc(m_inx1 & lpt_mask) - __popc(m_outx1 & lpt_mask);
    if (ML) sc5[4] = wx1_sum;
    block_excl_scan5_dpp<RUNS_WAVES>(sc5, s_tmp[1], lane, wave);  // exclusive prefixes A, B, C and of the one-sided counts' changes
    const int g_c = g_0 - sc5[0];  // g at the thread's first lag
    // n11 at the thread's first lag (wrap-around arithmetic: exact whenever the result fits, which it does)
    const int n11_c = n11_0 + RUNS_LPT * tid * g_0 - RUNS_LPT * ((tid - 1) * sc5[0] - sc5[1]) - sc5[2];
    const int n1x_c = n1x_0 + sc5[3], nx1_c = nx1_0 + sc5[4];  // the counts at the thread's first lag
#if defined(FFS_RUNS_STOP) && FFS_RUNS_STOP == 4
    if (n_q >= 0) {
        if (n11_c == 0x7fffffff) best[0].d = n1x_c + nx1_c;
        return;
    }
#endif
    const int lim = (Wt - c) < RUNS_LPT ? (Wt - c) : RUNS_LPT;               // lags of this thread inside the tile (<= 0: none)
    // score(d) = c0 ov + c1x n1x + cx1 nx1 + c11 n11 (two_level_score() multiplied out; ML: the weighted counts M11 / Mx1
    // with the coefficients of the LevelInfo comment); fp32 first
    double k0 = cd.s0 * cd.r0, k1x = cd.r0 * (cd.s1 - cd.s0), kx1 = cd.s0 * (cd.r1 - cd.r0), k11 = (cd.s1 - cd.s0) * (cd.r1 - cd.r0);
    if (ML) {
        const double base = 2.0 * lev_lam0 - 1.0, two_q = 2.0 * lev_q;
        k0 = base * cd.s0, k1x = base * (cd.s1 - cd.s0), kx1 = two_q * cd.s0, k11 = two_q * (cd.s1 - cd.s0);
    }
    const float f0 = (float)k0, f1x = (float)k1x, fx1 = (float)kx1, f11 = (float)k11;
    // (ML: the counts are up to LEVELS_MAX_MULT * 3 times a sample count: the same bound on the sums holds with that factor)
    const float count_max = (float)(R > S ? R : S) * (ML ? (float)(3 * LEVELS_MAX_MULT) : 1.0f);
    // fp32 prefilter.  a(d) = f11 n11(d) + E(d), E = f1x n1x + fx1 nx1 + f0 ov the part that only moves where the
    // overlap's ends pass a sample: |E(d+1) - E(d)| <= |f1x| + |fx1| + |f0| =: step.  E is refreshed every FOUR lags (exact
    // integer counts, one popcount of four mask bits each) and used unchanged for the three lags behind -- at most
    // 3 step off.  |fp32 value - exact| <= 12 roundings of 2^-24 on sums of at most (|k0|+|k1x|+|kx1|+|k11|) max(R, S);
    // a lag whose prefilter value lies within 2 (rounding bound + 3 step) of the block's best may hold the maximum.
    const float estep = fabsf(f1x) + fabsf(fx1) + fabsf(f0);
    const float estep_w = fabsf(f1x) + (ML ? (float)(3 * LEVELS_MAX_MULT) : 1.0f) * fabsf(fx1) + fabsf(f0);  // (one lag moves Mx1 by up to sum m_k)
    const float margin = 24.0f * 1.1920929e-7f * (estep + fabsf(f11)) * count_max * 1.01f + 6.0f * estep_w * 1.01f + 1e-3f;
    auto edge32 = [&](int a1x, int ax1, int d) -> float {
        const int j0 = d < 0 ? -d : 0;
        const int j1 = (R - d) < S ? (R - d) : S;
        const int ov = j1 > j0 ? (j1 - j0) : 0;
        return fmaf(f1x, (float)a1x, fmaf(fx1, (float)ax1, f0 * (float)ov));
    };
    float tmax = -INFINITY;
    auto prefilter = [&](auto masked) {  // (masked: the thread's lags beyond `lim` do not count -- the tile's last threads only)
        int g = g_c, a11 = n11_c, a1x = n1x_c, ax1 = nx1_c;
        unsigned mi1 = m_in1x, mo1 = m_out1x, mix = m_inx1, mox = m_outx1;
#pragma unroll 1
        for (int q4 = 0; q4 < RUNS_LPT / 4; ++q4) {
            int h[4];
            unpack4(q4, h);
            const float e32 = edge32(a1x, ax1, D0 + c + 4 * q4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f = fmaf(f11, (float)a11, e32);
                if (decltype(masked)::value)
                    tmax = fmaxf(tmax, (4 * q4 + e) < lim ? f : -INFINITY);
                else
                    tmax = fmaxf(tmax, f);
                g -= h[e];
                a11 += g;
            }
            a1x += __popc(mi1 & 15u) - __popc(mo1 & 15u);
            if (ML) {
                ax1 += wx1_of(q4);
            } else {
                ax1 += __popc(mix & 15u) - __popc(mox & 15u);
            }
            mi1 >>= 4, mo1 >>= 4, mix >>= 4, mox >>= 4;
        }
    };
    if (lim >= RUNS_LPT)
        prefilter(std::false_type{});
    else if (lim > 0)
        prefilter(std::true_type{});
    const float wm = wave_max_f32(tmax);
    if (lane == 0) s_m[wave] = wm;
    __syncthreads();
    float bm = s_m[0];
#pragma unroll
    for (int wv = 1; wv < RUNS_WAVES; ++wv) bm = fmaxf(bm, s_m[wv]);
    const float thr = bm - margin;
#if defined(FFS_RUNS_STOP) && FFS_RUNS_STOP == 5
    if (n_q >= 0) {
        if (thr == 123.25f) best[0].d = 1;
        return;
    }
#endif
    double bs = -INFINITY;
    int bd = INT32_MIN;
    if (tmax >= thr) {  // exact re-evaluation of the lags that may hold the maximum
        int g = g_c, a11 = n11_c, a1x = n1x_c, ax1 = nx1_c;
        unsigned mi1 = m_in1x, mo1 = m_out1x, mix = m_inx1, mox = m_outx1;
#pragma unroll 1
        for (int q4 = 0; q4 < RUNS_LPT / 4; ++q4) {
            int h[4];
            unpack4(q4, h);
            const float e32 = edge32(a1x, ax1, D0 + c + 4 * q4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * q4 + e;
                if (i < lim && fmaf(f11, (float)a11, e32) >= thr) {
                    const unsigned low = (1u << e) - 1u;  // the one-sided counts at this lag: e mask bits further
                    const int b1x = a1x + __popc(mi1 & low) - __popc(mo1 & low);
                    const int d = D0 + c + i;
                    double sc;
                    if (ML) {
                        int bx1 = ax1;  // (rare: the planes' windows are fetched again for the partial group)
                        for (int k = 0; k < n_lv; ++k) {
                            const GWords pb = (GWords)refs[vr0 + k].bits;
                            const unsigned mo = fetch32(pb, R, dc) >> (4 * q4), mi = fetch32(pb, R, (long long)S + dc) >> (4 * q4);
                            bx1 += lev_m(k) * (__popc(mi & low) - __popc(mo & low));
                        }
                        const int j0 = d < 0 ? -d : 0;
                        const int j1 = (R - d) < S ? (R - d) : S;
                        const int ov = j1 > j0 ? (j1 - j0) : 0;
                        sc = (double)ov * k0;
                        sc = __builtin_fma((double)b1x, k1x, sc);
                        sc = __builtin_fma((double)bx1, kx1, sc);
                        sc = __builtin_fma((double)a11, k11, sc);
                    } else {
                        const int bx1 = ax1 + __popc(mix & low) - __popc(mox & low);
                        sc = two_level_score(cd, a11, b1x, bx1, d);
                    }
                    if (sc >= bs) bs = sc, bd = d;
                }
                g -= h[e];
                a11 += g;
            }
            a1x += __popc(mi1 & 15u) - __popc(mo1 & 15u);
            if (ML) {
                ax1 += wx1_of(q4);
            } else {
                ax1 += __popc(mix & 15u) - __popc(mox & 15u);
            }
            mi1 >>= 4, mo1 >>= 4, mix >>= 4, mox >>= 4;
        }
    }
#if defined(FFS_RUNS_STOP) && FFS_RUNS_STOP == 6
    if (n_q >= 0) {
        if (bs == 123.25) best[0].d = bd;
        return;
    }
#endif
    // block argmax: larger score, then larger lag.  Inside a wave only the few lanes that re-evaluated a lag take part (most
    // waves have none): their values are read lane by lane into scalar registers -- no shuffle butterfly over 64 lanes
    double ws_ = -INFINITY;
    int wd = INT32_MIN;
    for (unsigned long long mask = __ballot(bd != INT32_MIN); mask; mask &= mask - 1) {
        const int l = __builtin_ctzll(mask);
        const unsigned lo32 = (unsigned)__builtin_amdgcn_readlane((int)__double2loint(bs), l);
        const int hi32 = __builtin_amdgcn_readlane(__double2hiint(bs), l);
        const double os = __hiloint2double(hi32, (int)lo32);
        const int od = __builtin_amdgcn_readlane(bd, l);
        if (os > ws_ || (os == ws_ && od > wd)) ws_ = os, wd = od;
    }
    if (lane == 0) s_sc[wave] = ws_, s_d[wave] = wd;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RUNS_WAVES; ++i)
        if (s_sc[i] > ws_ || (s_sc[i] == ws_ && s_d[i] > wd)) ws_ = s_sc[i], wd = s_d[i];
    if (bd == wd && wd != INT32_MIN) {  // the one thread that owns the winning lag
        if (n_tiles == 1) {
            runs_write_cand(cd, cres, ci, bs, bd, false);
        } else {
            RunsBest& o = best[(size_t)ci * tiles_max + tile];
            o.score = bs, o.d = bd;
        }
    }
    }  // candidates of this workgroup
}

__global__ __launch_bounds__(RUNS_THREADS, FFS_RUNS_WPS) void k_runs_corr(
    const CandDesc* __restrict__ cands, int n_cand, const RunsRef* __restrict__ refs, CandResult* __restrict__ cres,
    RunsBest* __restrict__ best, int tiles_max, const int* __restrict__ chunk_flags, int pairs_per_chunk, int split) {
    runs_corr_body<false>(cands, n_cand, refs, cres, best, tiles_max, chunk_flags, pairs_per_chunk, nullptr, 0, split);
}
// multi-level references (threshold lists + LevelInfo); a few more registers: three workgroups per CU
__global__ __launch_bounds__(RUNS_THREADS, 6) void k_runs_corr_ml(
    const CandDesc* __restrict__ cands, int n_cand, const RunsRef* __restrict__ refs, CandResult* __restrict__ cres,
    RunsBest* __restrict__ best, int tiles_max, const int* __restrict__ chunk_flags, int pairs_per_chunk,
    const LevelInfo* __restrict__ linfo, int n_vec_all) {
    runs_corr_body<true>(cands, n_cand, refs, cres, best, tiles_max, chunk_flags, pairs_per_chunk, linfo, n_vec_all, n_cand);
}

// Multi-level flags: sub-batch = 1 when a reference of it is not usable (LevelInfo.ok == 0, a truncated threshold list, a
// histogram cell that could overflow its 16 bits) or some candidate's coincidences (summed over the threshold lists)
// exceed the budget.
__global__ __launch_bounds__(256) void k_runs_chunk_flags_ml(const CandDesc* __restrict__ cands, int n_pairs, int n_cand,
                                                             int pairs_per_chunk, const RunsRef* __restrict__ refs,
                                                             const LevelInfo* __restrict__ linfo, int n_vec_all, long long budget,
                                                             int* __restrict__ flags, unsigned long long* __restrict__ stats) {
    const int ch = blockIdx.x;
    const int p0 = ch * pairs_per_chunk, p1 = (p0 + pairs_per_chunk) < n_pairs ? (p0 + pairs_per_chunk) : n_pairs;
    int over = 0, longest = 0;
    unsigned long long nb = 0;
    for (int i = p0 * n_cand + (int)threadIdx.x; i < p1 * n_cand; i += 256) {
        const CandDesc& cd = cands[i];
        const int pair = i / n_cand, j = i - pair * n_cand;
        const RunsRef rs = refs[pair * (n_cand + 1) + 1 + j];
        const LevelInfo li = linfo[pair];
        const int n_p = ((GInts)rs.hdr)[0];
        const int cap_p = rs.cap > 0 ? rs.cap : ((GInts)rs.hdr)[3];
        nb += (unsigned)n_p;
        if (!li.ok || n_p >= cap_p || n_p >= RUNS_CAP) {
            if (n_p > longest) longest = n_p;
            over = 1;
            continue;
        }
        long long coinc = 0, cell = 0;
        if (n_p > longest) longest = n_p;
        for (int k = 0; k < li.n_levels - 1; ++k) {
            const RunsRef rk = refs[n_vec_all + 3 * pair + k];
            const int n_q = ((GInts)rk.hdr)[0];
            if (n_q > longest) longest = n_q;
            if (j == 0) nb += (unsigned)n_q;
            if (n_q >= rk.cap) over = 1;
            coinc += (long long)n_p * n_q;
            cell += (long long)li.m[k] * (n_p < n_q ? n_p : n_q);
        }
        if (cell >= 32768) over = 1;  // |h(d)| < 2^15 for the packed histogram
        if (!(cd.flags & CAND_NO_LAGS) && coinc * ((long long)cd.d_hi - cd.d_lo + 1) / (cd.R > 0 ? cd.R : 1) > budget) over = 1;
    }
    over = __syncthreads_or(over);
    __shared__ unsigned long long s_nb[4];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) nb += __shfl_xor(nb, s, 64);
    if ((threadIdx.x & 63) == 0) s_nb[threadIdx.x >> 6] = nb;
    __syncthreads();
    if (threadIdx.x == 0) {
        flags[ch] = over;
        atomicAdd(stats, s_nb[0] + s_nb[1] + s_nb[2] + s_nb[3]);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const int o = __shfl_xor(longest, s, 64);
        longest = o > longest ? o : longest;
    }
    if ((threadIdx.x & 63) == 0 && longest > 0) atomicMax(stats + 1, (unsigned long long)longest);
}

// candidates whose window spans several tiles: best tile result (larger score, then larger lag = later tile)
__global__ void k_runs_pick(const CandDesc* __restrict__ cands, int n, int n_cand, const RunsBest* __restrict__ best,
                            int tiles_max, CandResult* __restrict__ cres, const int* __restrict__ chunk_flags, int pairs_per_chunk) {
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= n) return;
    const CandDesc& cd = cands[ci];
    if (chunk_flags[(ci / n_cand) / pairs_per_chunk]) return;
    if (cd.flags & CAND_NO_LAGS) return;
    const int W = cd.d_hi - cd.d_lo + 1;
    const int n_tiles = (W + RUNS_T - 1) / RUNS_T;
    if (n_tiles <= 1) return;
    RunsBest b = best[(size_t)ci * tiles_max];
    for (int t = 1; t < n_tiles; ++t) {
        const RunsBest& o = best[(size_t)ci * tiles_max + t];
        if (o.score >= b.score) b = o;
    }
    runs_write_cand(cd, cres, ci, b.score, b.d, false);
}

// ---------------------------------------------------------------------------------------------------------------
// List-only vectors of a sub-batch that goes through the transforms after all: their samples as bits.
struct ExpandVec {
    const int2* e;
    const int2* hdr;
    unsigned* dst;
    int32_t len;
    int32_t pad;
};
// A caller-owned block is read through its own header (n, ones, len, cap): a truncated list (n >= cap) or one made for
// another vector length is never followed past the block -- n is clamped to the entries the block holds and `err`
// (optional) is raised so that the entry point can report FFS_E_INVALID like the run-boundary path's flags kernel does.
__global__ __launch_bounds__(256) void k_runs_expand(const ExpandVec* __restrict__ vecs, int chunks_per_vec, int* __restrict__ err) {
    const int v = blockIdx.x / chunks_per_vec, c = blockIdx.x - v * chunks_per_vec;
    const ExpandVec ev = vecs[v];
    if (!ev.dst) return;
    const int n_words = (ev.len + 31) >> 5;
    const int w = c * 256 + (int)threadIdx.x;
    if (w >= n_words) return;
    const GInts hd = (GInts)ev.hdr;
    int n = hd[0];
    const int cap = hd[3];
    const bool broken = n < 0 || n >= cap || hd[2] != ev.len;
    if (n >= cap) n = cap > 0 ? cap - 1 : 0;
    if (n < 0) n = 0;
    if (broken && err && w == 0) atomicOr(err, 1);
    unsigned m = list_bits32((GEntries)ev.e, n, (long long)w * 32);
    if (w == n_words - 1 && (ev.len & 31)) m &= (1u << (ev.len & 31)) - 1u;
    ev.dst[w] = m;
}

// ---------------------------------------------------------------------------------------------------------------
// Subtitle rasteriser that never makes a bitmap (SubtitleScaler + SubtitleSpeechTransformer, subtitle_transformers.py:
// 35-47, speech_transformers.py:957-980): the union of a track's scaled [start, end) sample intervals IS the boundary
// list.  One workgroup per vector; the track's subtitles arrive sorted by start time (the entry point sorts a copy when
// they are not), the scaling is monotone and -- start_seconds <= 0, checked by the entry point -- no start sample is
// negative, so the rasterised starts are non-decreasing and a run begins exactly where a start lies beyond every earlier
// end (touching intervals merge: a boundary that closes and opens at the same sample is none).  Per chunk of 1024
// subtitles: interval arithmetic (raster_interval, shared with the host), a block-wide running maximum of the ends, a
// block scan of (new runs, their starts, the ends in front of them) -- ones in front of run r = ends of runs < r minus
// starts of runs < r -- and every run start writes its own entry and the end entry of the run in front of it.
struct RasterRunsVec {
    long long sub_first, out_off;  // first subtitle of the track in the concatenated arrays; byte offset of the list block
    double ratio;
    int32_t sub_count, len, cap, pad;  // cap: entries the block has room for (>= 2 * sub_count + 1)
};
__global__ __launch_bounds__(256) void k_rasterize_runs(const long long* __restrict__ start_us, const long long* __restrict__ end_us,
                                                        const unsigned char* __restrict__ meta, const RasterRunsVec* __restrict__ vecs,
                                                        double sample_rate, double start_seconds, char* __restrict__ out) {
    constexpr int IT = 4, CHUNK = 256 * IT;
    const RasterRunsVec rv = vecs[blockIdx.x];
    int2* hdr = reinterpret_cast<int2*>(out + rv.out_off);
    int2* e = reinterpret_cast<int2*>(out + rv.out_off + 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int s_max[4];
    __shared__ unsigned s_cnt[4], s_sa[4], s_se[4];
    int cmax = -1;                             // largest end so far (-1: none yet)
    unsigned n_runs = 0, sum_a = 0, sum_e = 0;  // runs begun so far, sum of their starts, sum of the ends in front of them
    for (int base = 0; base < rv.sub_count; base += CHUNK) {
        int a[IT], b[IT];
        bool ok[IT];
        int tmax = -1;
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = base + tid * IT + k;
            ok[k] = false;
            a[k] = b[k] = 0;
            if (i < rv.sub_count && !(meta && meta[rv.sub_first + i])) {  // speech_transformers.py:966-967
                long long la, lb;
                if (raster_interval(start_us[rv.sub_first + i], end_us[rv.sub_first + i], rv.ratio, sample_rate, start_seconds, rv.len,
                                    &la, &lb)) {
                    ok[k] = true;
                    a[k] = (int)la, b[k] = (int)lb;
                    tmax = tmax > b[k] ? tmax : b[k];
                }
            }
        }
        // exclusive running maximum of the ends over the threads of the block
        int im = tmax;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int o = __shfl_up(im, s, 64);
            if (lane >= s) im = im > o ? im : o;
        }
        int pm = __shfl_up(im, 1, 64);
        if (lane == 0) pm = -1;
        __syncthreads();  // (the previous chunk's readers of s_* are done)
        if (lane == 63) s_max[wave] = im;
        __syncthreads();
        int run_max = cmax;
        for (int wv = 0; wv < 4; ++wv) {
            if (wv < wave) pm = pm > s_max[wv] ? pm : s_max[wv];
            run_max = run_max > s_max[wv] ? run_max : s_max[wv];
        }
        pm = pm > cmax ? pm : cmax;  // largest end in front of this thread's first subtitle
        // runs that begin in this thread
        unsigned cnt = 0, sa = 0, se = 0;
        bool flag[IT];
        int em[IT];
        {
            int m = pm;
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                em[k] = m;
                flag[k] = ok[k] && a[k] > m;
                if (flag[k]) {
                    cnt += 1;
                    sa += (unsigned)a[k];
                    se += m >= 0 ? (unsigned)m : 0u;
                }
                if (ok[k]) m = m > b[k] ? m : b[k];
            }
        }
        unsigned ic = cnt, ia = sa, ie = se;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const unsigned oc = __shfl_up(ic, s, 64), oa = __shfl_up(ia, s, 64), oe = __shfl_up(ie, s, 64);
            if (lane >= s) ic += oc, ia += oa, ie += oe;
        }
        if (lane == 63) s_cnt[wave] = ic, s_sa[wave] = ia, s_se[wave] = ie;
        __syncthreads();
        unsigned r = n_runs + ic - cnt, xa = sum_a + ia - sa, xe = sum_e + ie - se;  // exclusive prefixes incl. earlier chunks
        unsigned tc = 0, ta = 0, te = 0;
        for (int wv = 0; wv < 4; ++wv) {
            if (wv < wave) r += s_cnt[wv], xa += s_sa[wv], xe += s_se[wv];
            tc += s_cnt[wv], ta += s_sa[wv], te += s_se[wv];
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            if (flag[k]) {
                const unsigned endp = em[k] >= 0 ? (unsigned)em[k] : 0u;
                xe += endp;  // ends of the runs in front of run r, the one that closes here included
                const int ones = (int)(xe - xa);
                if (2 * r < (unsigned)rv.cap) e[2 * r] = make_int2(a[k], ones);
                if (r >= 1 && 2 * r - 1 < (unsigned)rv.cap) e[2 * r - 1] = make_int2(em[k], ones);
                xa += (unsigned)a[k];
                r += 1;
            }
        }
        n_runs += tc, sum_a += ta, sum_e += te;
        cmax = run_max;
    }
    if (tid == 0) {
        const unsigned n = 2 * n_runs;
        const int ones = n_runs ? (int)(sum_e + (unsigned)cmax - sum_a) : 0;
        if (n_runs && n - 1 < (unsigned)rv.cap) e[n - 1] = make_int2(cmax, ones);
        if (n < (unsigned)rv.cap) e[n] = make_int2(INT32_MAX, ones);
        hdr[0] = make_int2((int)n, ones);
        hdr[1] = make_int2(rv.len, rv.cap);
    }
}

}  // namespace ffsa
