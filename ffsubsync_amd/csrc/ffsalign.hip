// ffsalign.hip -- host side of libffsalign.so: plans, descriptor building, kernel dispatch and
// the extern "C" entry points declared in include/ffsubsync_amd.h.
//
// Written for gfx950 (MI355X) only: hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define FFS_HOST_AVX2 1  // host pass only: the device pass parses host functions too, without the x86 headers
#else
#define FFS_HOST_AVX2 0
#endif
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/ffsubsync_amd.h"
#include "ffs_kernels.h"
#include "ffs_runs.h"

using namespace ffsa;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(FFS_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// run `stmt` with a compile-time DT equal to the runtime element type (0 = bytes, 1 = float, 2 = bit-packed, 3 = double)
#define FFS_BY_DTYPE(dtype, stmt)              \
    do {                                       \
        if ((dtype) == FFS_DTYPE_U8) {         \
            constexpr int DT = 0;              \
            stmt;                              \
        } else if ((dtype) == FFS_DTYPE_F32) { \
            constexpr int DT = 1;              \
            stmt;                              \
        } else if ((dtype) == FFS_DTYPE_F64) { \
            constexpr int DT = 3;              \
            stmt;                              \
        } else {                               \
            constexpr int DT = 2;              \
            stmt;                              \
        }                                      \
    } while (0)

// the exact re-evaluation kernels: DT as above, or 4 when the reference and the candidates differ in element type
#define FFS_BY_RESCORE_DTYPE(mixed, dtype, stmt)  \
    do {                                          \
        if (mixed) {                              \
            constexpr int DT = 4;                 \
            stmt;                                 \
        } else if ((dtype) == FFS_DTYPE_U8) {     \
            constexpr int DT = 0;                 \
            stmt;                                 \
        } else if ((dtype) == FFS_DTYPE_F32) {    \
            constexpr int DT = 1;                 \
            stmt;                                 \
        } else if ((dtype) == FFS_DTYPE_F64) {    \
            constexpr int DT = 3;                 \
            stmt;                                 \
        } else {                                  \
            constexpr int DT = 2;                 \
            stmt;                                 \
        }                                         \
    } while (0)

constexpr int64_t kMinFftN = 4096;  // shorter problems go to the exact direct kernel
constexpr int64_t kMaxFftN = 1 << 24;
constexpr unsigned kPoolCapacity = 1u << 20;
// FFS_ALGO_AUTO: boundary coincidences per candidate, per point of the plan's transform length and packed transform slot
// the candidate occupies, above which the transforms take over.  Measured break-even (profiles/budget_probe.py, round 5):
// +-60 s window: ~20 per point of N = 3 * 2^18 (lists of 17 k x 20 k boundaries); no window: ~13 per point of N = 3 * 2^19
// (118 lag tiles per candidate cost as much as 6 M coincidences).  Twelve keeps the choice on the winning side in both.
constexpr long long kRunsBudgetPerPoint = 12;
// Plan-owned boundary lists (vectors that arrive as bits) start with room for this many entries each -- subtitle-like vectors
// have ~2 000 -- and the stride grows (x 4, up to RUNS_CAP) when a call meets a longer list: that call's sub-batch goes through
// the transforms (a list that fills its slot counts as over budget), the next call has the room.  A 256 KB stride for 16 KB
// lists cost 4 % of k_runs_extract (65 536 lists spread over 16 GB of address space).
constexpr int kRunsStride0 = 4096;
// calls of at most this many candidates upload their descriptors once and launch the extraction behind them (see late_extract)
constexpr size_t kSmallCallCands = 4096;
constexpr int kSegBlocks = 3;    // blocks per candidate in block-segmented mode (n_fft = 3 * block transform length)
constexpr int kCollectRows = 4;  // grid rows of the exhaustive last pass (each walks the flagged-candidate list)

int ilog2(int64_t x) {
    int p = 0;
    while ((int64_t(1) << p) < x) ++p;
    return p;
}

// column-tile width for a column transform of length L
int tile_cols(int L) {
    if (L % 3 == 0) {  // 3 * 2^k columns: 48, 96, 192, 384, 768
        const int c = 3072 / L;
        return c < 16 ? 16 : c;
    }
    if (L <= 16) return 256;
    if (L <= 128) return 4096 / L;
    if (L <= 1024) return 16;
    if (L == 2048) return 8;
    return 4;
}

size_t col_lds_bytes(int L) {
    size_t t = (L > 16) ? (size_t)L * tile_cols(L) * sizeof(cf) : 0;
    if (L % 3 == 0) t += (size_t)L * sizeof(cf);  // W_L^k for the radix-3 combine
    return t < 2048 ? 2048 : t;  // the byte / bit input staging of pass A needs up to 2 KB on its own
}
size_t row_lds_bytes(int L) {
    const int rows = 256 / (L / 16);
    return (size_t)rows * (L + L / 16) * sizeof(cf);
}

// stage tables of Shape<L>: stage 1 (Ns = 16) then stage 2 (Ns = 256).  A radix-16 stage stores the
// six powers e in {1,2,3,4,8,12} of w = exp(-2*pi*i*jm/(Ns*16)) as [6][Ns]; a smaller radix R stores
// w^r as [R][Ns].
std::vector<cf> make_stage_tables(int L) {
    const int LT = L / 16;
    const int R1 = LT >= 16 ? 16 : LT;
    const int R2 = L / (16 * R1);
    std::vector<cf> t;
    auto add = [&](int R, int Ns) {
        static const int pw16[6] = {1, 2, 3, 4, 8, 12};
        const int rows = (R == 16) ? 6 : R;
        for (int i = 0; i < rows; ++i)
            for (int jm = 0; jm < Ns; ++jm) {
                const int e = (R == 16) ? pw16[i] : i;
                const double a = -2.0 * M_PI * (double)e * (double)jm / ((double)Ns * (double)R);
                t.push_back(mk((float)cos(a), (float)sin(a)));
            }
    };
    if (R1 > 1) add(R1, 16);
    if (R2 > 1) add(R2, 256);
    if (t.empty()) t.push_back(mk(1.f, 0.f));
    return t;
}

// Transform lengths 3 * 2^k the kernels can run: N = N1 * N2 with N1 = 3 * 2^j in [48, 768] columns.
bool radix3_length(int64_t n) {
    if (n % 3) return false;
    const int64_t m = n / 3;
    return (m & (m - 1)) == 0 && n >= 48 * 256 && n <= 768 * 4096;
}

cf wn(int64_t N, int64_t p) {
    p %= N;
    const double a = -2.0 * M_PI * (double)p / (double)N;
    return mk((float)cos(a), (float)sin(a));
}

}  // namespace

struct ffs_plan {
    int device = 0;
    int64_t N = 0;
    int N1 = 0, N2 = 0, C = 0, log2C = 0;
    int log2CL = 0;  // tile layout T[x/CL][k1][x%CL]: CL = max(C, 64) columns (512-byte row chunks)
    int pairs_in_flight = 0, max_cand = 0, max_slots = 0;
    int pass_a_prefetch = 3;  // grid rows the input prefetch blocks of pass A run ahead, byte inputs (FFS_PASS_A_PREFETCH, 0 = off)
    int pass_a_prefetch_bits = 12;  // the same for bit-packed inputs, in rows of a 2^18-point transform (scaled by length)
    bool direct_only = false;
    // the five run-time knobs; none of them can change a result
    bool allow_pruned = true;       // FFS_DISABLE_PRUNED_PASS_C=1 forces the full last pass
    bool allow_half_last = true;    // FFS_DISABLE_HALF_LAST=1: store all rows of a single-candidate last slot
    int lab_flags = 0;              // lab build only (make lab): DBG_* section switches, timing only
    // device tables
    cf *tw1 = nullptr, *tw2 = nullptr;        // stage tables for N1 / N2
    cf *tbA = nullptr, *tsA = nullptr;        // pass A inter twiddles: [N1/16][N2], [16][N2]
    cf* tw1h = nullptr;                       // stage tables of the 256-row sub-transforms of a 512-row column
    cf *tbR = nullptr, *tsR = nullptr, *thR = nullptr;  // pass-A twiddles of the three-sub-transforms-per-thread columns (k_pass_a3)
    cf *tbM = nullptr, *tsM = nullptr;        // mid inter twiddles:    [N1][N2/16], [N1][16]
    cf* twn1 = nullptr;                       // W_N1^k, k < N1 (pruned pass C)
    cf* work = nullptr;                       // [pairs_in_flight][max_slots][N]
    BlockNom* bnom = nullptr;                 // [pairs_in_flight*n_packed*2][tiles]
    PoolEntry* pool_entries = nullptr;        // [kPoolCapacity] exhaustive fallback for flagged candidates
    int* xlist = nullptr;                     // [1 + pairs_in_flight*max_cand] flagged candidates of the sub-batch
    // Block-segmented mode (n_fft = 3*M, M = 2^k >= 2^16): a complete power-of-two plan of length M with
    // room for three blocks per pair; used when the lag window is narrow enough (ffs_align_batch)
    ffs_plan* seg = nullptr;
    bool allow_seg = true;                    // FFS_DISABLE_SEGMENTED=1
    // per-call descriptor storage (grown on demand)
    void* dev_desc = nullptr;
    size_t dev_desc_bytes = 0;
    void* host_desc = nullptr;                // pinned
    size_t host_desc_bytes = 0;
    hipEvent_t upload_done = nullptr;
    // Run-boundary calls that extract lists upload their descriptors on a COPY STREAM (one per device, shared by the plans)
    // into one of TWO device blocks (dev_desc / dev_desc2, alternating): the vector table of call k + 1 goes up while
    // call k's correlation runs, a large call's candidate descriptors while the extraction of their own call runs -- on
    // one stream the uploads sat between the kernels with the device idle (126 + 70 us of a 3.3 ms step of 8192 pairs,
    // ~20 us of a 111 us step of 128: profiles/large_step_timeline.py, small_step_timeline.py).
    void* dev_desc2 = nullptr;
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_ev = nullptr;                 // the vector table of the current call has arrived
    hipEvent_t half_done[2] = {nullptr, nullptr};  // the last call that used block h has finished
    bool half_used[2] = {false, false};
    int cur_half = 0;                             // the block the current (or most recent) call uses
    // The workspace, descriptor and nominee buffers are reused by every call: a call on another stream
    // than the previous one first waits for that one's last kernel (same-stream calls are ordered anyway).
    hipEvent_t last_done = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_last = false;
    int64_t workspace_bytes = 0;
    bool radix3_off = false;                  // FFS_DISABLE_RADIX3=1 when the plan was created (ffs_plan_length)
    bool workspace_ready = false;             // every buffer of ensure_workspace is allocated (all or nothing)
    // Run-boundary path (ffs_runs.h): FFS_ALGO_AUTO sends every sub-batch of bit-packed two-level vectors whose boundary
    // lists are short enough through it (one event wait per call to read the list lengths), the others through the
    // transforms; FFS_ALGO_FFT never uses it; FFS_ALGO_RUNS ignores the coincidence budget (truncated lists still go
    // through the transforms).  Buffers grown on demand.
    int algo = FFS_ALGO_AUTO;
    int runs_split = 0;                 // FFS_RUNS_SPLIT: workgroups per pair in k_runs_corr (0: the rule at the launch site)
    long long runs_budget = -1;         // FFS_RUNS_BUDGET: boundary coincidences per candidate above which the transforms take over (-1: the rule in ffs_plan_create)
    int2* runs_e = nullptr;             // [vectors][runs_stride] (boundary position, ones in front of it) of the vectors that arrive as bits
    int runs_stride = 4096;             // entries per plan-owned list (kRunsStride0; FFS_RUNS_STRIDE; grows up to RUNS_CAP)
    size_t runs_e_entries = 0;          // entries allocated behind runs_e
    int2* runs_n = nullptr;             // [vectors] (boundaries, ones)
    size_t runs_vecs = 0;               // vectors the two buffers above have room for
    unsigned* pack_buf = nullptr;       // bit-packed images of a call's FFS_DTYPE_U8 vectors / of list-only vectors that need the transforms
    size_t pack_bytes = 0;
    int* runs_flags = nullptr;          // [sub-batches] 1 = goes through the transforms, 2 = unusable list (k_runs_chunk_flags); then the
                                        // 8-byte boundary counter of the call
    int* runs_zero_flags = nullptr;     // [sub-batches] zeros: calls whose host-known list bounds rule the transforms out
    int* runs_flags_host = nullptr;     // pinned copy of runs_flags
    size_t runs_flags_n = 0;
    RunsBest* runs_best = nullptr;      // [candidates][tiles] (windows wider than one tile)
    size_t runs_best_n = 0;
    bool runs_prev_fft = false;         // the previous run-boundary call needed the transforms for some sub-batch
    unsigned* lvl_buf = nullptr;        // threshold planes of a call's multi-level (float) references
    size_t lvl_bytes = 0;
    hipEvent_t runs_ev = nullptr;
    int64_t runs_calls = 0, runs_fft_chunks = 0, runs_chunks = 0, runs_last_boundaries = 0;  // statistics (ffs_plan_runs_stats)
    // FFS_HOST_TIMING=1: host nanoseconds of the run-boundary calls by section, printed when the plan is destroyed
    bool host_timing = false;
    double ht_vec = 0, ht_cand = 0, ht_wait = 0, ht_decide = 0, ht_total = 0;
    // kernels whose dynamic-LDS limit has been raised on this plan's device (the attribute is per device)
    mutable std::vector<const void*> lds_configured;
    // optional per-kernel event timing
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;            // reusable events
    size_t ev_used = 0;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> ev_spans;  // (kernel id, (start, stop))
};

namespace {

template <class T>
int upload(T** dst, const std::vector<T>& src, int64_t* total) {
    HIP_TRY(hipMalloc((void**)dst, src.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    *total += (int64_t)(src.size() * sizeof(T));
    return FFS_OK;
}

int ensure_desc(ffs_plan* p, size_t bytes) {
    if (bytes <= p->dev_desc_bytes) return FFS_OK;
    if (p->upload_done) HIP_TRY(hipEventSynchronize(p->upload_done));
    if (p->dev_desc) HIP_TRY(hipFree(p->dev_desc));
    if (p->dev_desc2) HIP_TRY(hipFree(p->dev_desc2));  // (allocated again by ensure_copy, at the new size)
    p->dev_desc2 = nullptr;
    p->half_used[1] = false;
    if (p->host_desc) HIP_TRY(hipHostFree(p->host_desc));
    p->dev_desc = nullptr;
    p->host_desc = nullptr;
    p->dev_desc_bytes = p->host_desc_bytes = 0;
    const size_t cap = bytes + bytes / 2 + 4096;
    HIP_TRY(hipMalloc(&p->dev_desc, cap));
    HIP_TRY(hipHostMalloc(&p->host_desc, cap, hipHostMallocDefault));
    p->dev_desc_bytes = p->host_desc_bytes = cap;
    return FFS_OK;
}

// The copy stream: ONE per device for every plan of the process, created on first use (the calling thread's current device
// is `device`) and never destroyed -- a stream per plan made plan creation slow enough to stretch the GPU test suite from
// 40 s to 7.6 min (round 6, first attempt).  Plans only ever queue their own descriptor uploads on it, each behind an
// event of their own that has long completed; what they share is the order of those uploads.
hipStream_t shared_copy_stream(int device) {
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!streams[device] && hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        streams[device] = nullptr;
    }
    return streams[device];
}
// Copy stream, its event and the second descriptor block (see ffs_plan::dev_desc2); false: this call uploads on its own stream.
bool ensure_copy(ffs_plan* p) {
    if (!p->copy_stream) {
        p->copy_stream = shared_copy_stream(p->device);
        if (!p->copy_stream) return false;
    }
    if (!p->copy_ev && hipEventCreateWithFlags(&p->copy_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        p->copy_ev = nullptr;
        return false;
    }
    if (!p->dev_desc2 && hipMalloc(&p->dev_desc2, p->dev_desc_bytes) != hipSuccess) {
        (void)hipGetLastError();
        p->dev_desc2 = nullptr;
        return false;
    }
    return true;
}

// Raise a kernel's dynamic-LDS limit once per plan (tiles above 64 KB need it).
int ensure_lds(const ffs_plan* p, const void* fn, size_t bytes) {
    for (const void* f : p->lds_configured)
        if (f == fn) return FFS_OK;
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    p->lds_configured.push_back(fn);
    return FFS_OK;
}

// Transform workspace of a plan (and of its block-segmented sub-plan), allocated on first use.  All or nothing: a failed
// allocation frees what this call had obtained, so a later call tries again instead of launching on null buffers.
int ensure_workspace(ffs_plan* p) {
    if (p->direct_only || p->workspace_ready) return FFS_OK;
    const size_t work_bytes = (size_t)p->pairs_in_flight * p->max_slots * p->N * sizeof(cf);
    const size_t bn_bytes = (size_t)p->pairs_in_flight * (p->max_slots - 1) * 2 * (p->N2 / p->C) * sizeof(BlockNom);
    bool ok = hipMalloc((void**)&p->work, work_bytes) == hipSuccess;
    ok = ok && hipMalloc((void**)&p->bnom, bn_bytes) == hipSuccess;
    ok = ok && hipMalloc((void**)&p->xlist, (1 + (size_t)p->pairs_in_flight * p->max_cand) * sizeof(int)) == hipSuccess;
    ok = ok && hipMalloc((void**)&p->pool_entries, (size_t)kPoolCapacity * sizeof(PoolEntry)) == hipSuccess;
    int rc = FFS_OK;
    if (!ok) {
        (void)hipGetLastError();
        rc = fail(FFS_E_NOMEM, "transform workspace of %zu bytes could not be allocated", work_bytes + bn_bytes);
    } else if (p->seg) {
        rc = ensure_workspace(p->seg);
    }
    if (rc) {
        (void)hipFree(p->work);
        (void)hipFree(p->bnom);
        (void)hipFree(p->xlist);
        (void)hipFree(p->pool_entries);
        p->work = nullptr;
        p->bnom = nullptr;
        p->xlist = nullptr;
        p->pool_entries = nullptr;
        return rc;
    }
    p->workspace_ready = true;
    return FFS_OK;
}

// Buffers of the run-boundary path: boundary lists for `n_vec` vectors that arrive as bits (0: every vector of the call
// brings its own list), `n_best` tile records, per-sub-batch flags.  Counted in
// ffs_plan_workspace_bytes (the Python plan cache budgets HBM by it).
int ensure_runs(ffs_plan* p, size_t n_vec, size_t n_best, size_t n_chunks) {
    if (!p->runs_ev) HIP_TRY(hipEventCreateWithFlags(&p->runs_ev, hipEventDisableTiming));
    auto quiesce = [&]() -> int {
        if (p->has_last) HIP_TRY(hipEventSynchronize(p->last_done));  // the previous call may still be using them
        return FFS_OK;
    };
    int rc;
    if (n_vec > p->runs_vecs || n_vec * (size_t)p->runs_stride > p->runs_e_entries) {  // more vectors, or a longer stride
        if ((rc = quiesce())) return rc;
        (void)hipFree(p->runs_e);
        (void)hipFree(p->runs_n);
        p->workspace_bytes -= (int64_t)(p->runs_e_entries * sizeof(int2) + p->runs_vecs * sizeof(int2));
        p->runs_e = nullptr;
        p->runs_n = nullptr;
        p->runs_vecs = 0;
        p->runs_e_entries = 0;
        const size_t cap = (n_vec > p->runs_vecs ? n_vec : p->runs_vecs) + n_vec / 4 + 64;
        const size_t entries = cap * (size_t)p->runs_stride;
        if (hipMalloc((void**)&p->runs_e, entries * sizeof(int2)) != hipSuccess ||
            hipMalloc((void**)&p->runs_n, cap * sizeof(int2)) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(p->runs_e);
            p->runs_e = nullptr;
            return fail(FFS_E_NOMEM, "boundary lists for %zu vectors (%zu bytes) could not be allocated", cap, entries * sizeof(int2));
        }
        p->runs_vecs = cap;
        p->runs_e_entries = entries;
        p->workspace_bytes += (int64_t)(entries * sizeof(int2) + cap * sizeof(int2));
    }
    if (n_best > p->runs_best_n) {
        if ((rc = quiesce())) return rc;
        (void)hipFree(p->runs_best);
        p->workspace_bytes -= (int64_t)(p->runs_best_n * sizeof(RunsBest));
        p->runs_best = nullptr;
        p->runs_best_n = 0;
        HIP_TRY(hipMalloc((void**)&p->runs_best, n_best * sizeof(RunsBest)));
        p->runs_best_n = n_best;
        p->workspace_bytes += (int64_t)(n_best * sizeof(RunsBest));
    }
    if (n_chunks > p->runs_flags_n) {
        if ((rc = quiesce())) return rc;
        (void)hipFree(p->runs_flags);
        (void)hipFree(p->runs_zero_flags);
        if (p->runs_flags_host) (void)hipHostFree(p->runs_flags_host);
        p->runs_flags = p->runs_zero_flags = p->runs_flags_host = nullptr;
        p->runs_flags_n = 0;
        const size_t cap = n_chunks + 64;  // (two 8-byte statistics sit behind the flags, 8-byte aligned: boundaries, longest list)
        const size_t bytes = ((cap * sizeof(int) + 7) & ~(size_t)7) + 16;
        HIP_TRY(hipMalloc((void**)&p->runs_flags, bytes));
        HIP_TRY(hipMalloc((void**)&p->runs_zero_flags, bytes));
        HIP_TRY(hipMemset(p->runs_zero_flags, 0, bytes));
        HIP_TRY(hipHostMalloc((void**)&p->runs_flags_host, bytes, hipHostMallocDefault));
        p->runs_flags_n = cap;
    }
    return FFS_OK;
}

// Calls on one plan from different streams: order them (see ffs_plan::last_done).
int enter_stream(ffs_plan* p, hipStream_t st) {
    if (p->has_last && p->last_stream != st) HIP_TRY(hipStreamWaitEvent(st, p->last_done, 0));
    return FFS_OK;
}
int leave_stream(ffs_plan* p, hipStream_t st) {
    // (every call, whichever stream uploaded its descriptors: a later call's copy-stream upload into this block waits for it)
    HIP_TRY(hipEventRecord(p->half_done[p->cur_half], st));
    p->half_used[p->cur_half] = true;
    HIP_TRY(hipEventRecord(p->last_done, st));
    p->last_stream = st;
    p->has_last = true;
    return FFS_OK;
}

// Entry points without a plan work on whatever device owns the caller's buffer -- and leave the thread's current
// device as they found it (a torch process would otherwise see its current device change under it).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    int enter(const void* dev_ptr) {
        hipPointerAttribute_t attr;
        if (dev_ptr && hipPointerGetAttributes(&attr, dev_ptr) == hipSuccess) {
            HIP_TRY(hipGetDevice(&prev));
            if (attr.device != prev) {
                HIP_TRY(hipSetDevice(attr.device));
                switched = true;
            }
        } else {
            (void)hipGetLastError();  // not a tracked allocation: stay on the caller's current device
        }
        return FFS_OK;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

hipEvent_t prof_event(ffs_plan* p) {
    if (p->ev_used == p->ev_pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        p->ev_pool.push_back(e);
    }
    return p->ev_pool[p->ev_used++];
}

// RAII span: records start now and stop at scope exit when profiling is on
struct ProfSpan {
    ffs_plan* p;
    hipStream_t st;
    int id;
    hipEvent_t a = nullptr, b = nullptr;
    ProfSpan(ffs_plan* p_, hipStream_t st_, int id_) : p(p_), st(st_), id(id_) {
        if (!p->profiling) return;
        a = prof_event(p);
        b = prof_event(p);
        if (a && b) (void)hipEventRecord(a, st);
    }
    ~ProfSpan() {
        if (a && b) {
            (void)hipEventRecord(b, st);
            p->ev_spans.push_back({id, {a, b}});
        }
    }
};

// ---- kernel dispatch -------------------------------------------------------------------------
// Bit-packed inputs: how many grid rows (transforms) the prefetch blocks of pass A run ahead -- about the time of
// twelve 2^18-point transforms (measured plateau: 9-14 rows there), but at least four rows (2^21-point transforms:
// pass A 17.3 us/pair at one row ahead, 15.8 at four, 15.9 at twelve), 0 = off.
int bit_prefetch_rows(const ffs_plan* p) {
    if (p->pass_a_prefetch_bits <= 0) return 0;
    const long long rows = ((long long)p->pass_a_prefetch_bits << 18) / (long long)p->N;
    return (int)(rows < 4 ? 4 : (rows > 255 ? 255 : rows));
}

#ifndef FFS_PAIR_MAX_XF
#define FFS_PAIR_MAX_XF 64  // A/B builds: largest transform group that takes the paired transform
#endif
// row_sel = first | count << 16: only transforms [first, first + count) of every group (0 = all of them; see k_pass_a)
template <int L, int C, int DT>
int launch_pass_a_inst(const ffs_plan* p, const XformDesc* descs, int n_xf, int xf_per_pair, int slots_per_pair,
                       int ref_half, hipStream_t st, int row_sel) {
    const size_t lds = col_lds_bytes(L);
    int rc_lds;
    const int nt = p->N2 / C;
    // byte inputs: one prefetch block per 128-column line group and grid row (see the kernel)
    const int groups = p->N2 / 128;
    // bit-packed inputs: eight prefetch blocks per grid row, one per XCD (see the kernel)
    const int ahead = DT == 2 ? bit_prefetch_rows(p) : p->pass_a_prefetch;
    const int pf = (DT == 0 && ahead > 0 && nt % 8 == 0 && groups % 8 == 0) ? groups
                   : (DT == 2 && ahead > 0 && nt % 8 == 0) ? 8 : 0;
    // bit-packed inputs, reference slot and last candidate slot both half slots (one real vector each): one paired
    // column transform per group instead of two, in the same launch as the group's other transforms
    constexpr bool CAN_PAIR = DT == 2 && L % 3 != 0;
    const bool paired = CAN_PAIR && !row_sel && (ref_half & HALF_REF) && (ref_half & HALF_LAST) && xf_per_pair >= 2 &&
                        xf_per_pair <= FFS_PAIR_MAX_XF && xf_per_pair == slots_per_pair;
    const int grid_rows = row_sel ? (n_xf / xf_per_pair) * (row_sel >> 16) : n_xf;
    const int flags = ref_half | STORE_8B | (p->lab_flags & (31 << 10));
    if constexpr (CAN_PAIR) {
        if (paired) {
            const size_t lds_p = lds > (size_t)L * C * sizeof(cf) ? lds : (size_t)L * C * sizeof(cf);  // the whole column tile
            const int groups_y = n_xf / xf_per_pair;
            if (xf_per_pair == 2) {
                if ((rc_lds = ensure_lds(p, (const void*)k_pass_a<L, C, DT, 1>, lds_p))) return rc_lds;
                hipLaunchKernelGGL((k_pass_a<L, C, DT, 1>), dim3(nt + pf, groups_y), dim3((L / 16) * C), lds_p, st, descs, p->work,
                                   p->N2, (long long)p->N, p->tw1, p->tbA, p->tsA, p->twn1, p->log2CL, xf_per_pair,
                                   slots_per_pair, nt, ahead, (unsigned*)p->bnom, flags, 0);
            } else {
                if ((rc_lds = ensure_lds(p, (const void*)k_pass_a<L, C, DT, 2>, lds_p))) return rc_lds;
                hipLaunchKernelGGL((k_pass_a<L, C, DT, 2>), dim3(nt + pf, groups_y * (xf_per_pair - 1)), dim3((L / 16) * C), lds_p,
                                   st, descs, p->work, p->N2, (long long)p->N, p->tw1, p->tbA, p->tsA, p->twn1, p->log2CL,
                                   xf_per_pair, slots_per_pair, nt, ahead, (unsigned*)p->bnom, flags, 0);
            }
            HIP_TRY(hipGetLastError());
            return FFS_OK;
        }
    }
    if ((rc_lds = ensure_lds(p, (const void*)k_pass_a<L, C, DT>, lds))) return rc_lds;
    hipLaunchKernelGGL((k_pass_a<L, C, DT>), dim3(nt + pf, grid_rows), dim3((L / 16) * C), lds, st, descs, p->work, p->N2,
                       (long long)p->N, p->tw1, p->tbA, p->tsA, p->twn1, p->log2CL, xf_per_pair, slots_per_pair, nt, ahead,
                       (unsigned*)p->bnom, flags, row_sel);
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}

// columns of length 3*LI with three sub-transforms per thread (bit-packed inputs): tiles of C = 4096/LI columns
bool col3r_ok(const ffs_plan* p) { return p->tbR && (p->N1 == 192 || p->N1 == 384 || p->N1 == 768 || p->N1 == 512); }
int col3r_cols(const ffs_plan* p) { return p->N1 == 512 ? 16 : 4096 / (p->N1 / 3); }

template <int NS, int LI, int C>
int launch_pass_a3_inst(const ffs_plan* p, const XformDesc* descs, int n_xf, int xf_per_pair, int slots_per_pair,
                        int ref_half, hipStream_t st, int row_sel) {
    const size_t lds = (size_t)LI * C * sizeof(cf);
    int rc_lds;
    const int nt = p->N2 / C;
    const int ahead = nt % 8 == 0 ? bit_prefetch_rows(p) : 0;  // eight prefetch blocks per grid row, one per XCD
    const cf* tw = NS == 2 ? p->tw1h : p->tw1;
    // reference slot and last candidate slot both half slots (one real vector each): one paired column transform
    const bool paired = !row_sel && (ref_half & HALF_REF) && (ref_half & HALF_LAST) && xf_per_pair >= 2 &&
                        xf_per_pair <= FFS_PAIR_MAX_XF && xf_per_pair == slots_per_pair;
    const int flags = ref_half | (ahead << 16);
    const dim3 gx(nt + (ahead ? 8 : 0));
    const int groups_y = n_xf / xf_per_pair;
#define FFS_A3_LAUNCH(PM, GY)                                                                                              \
    do {                                                                                                                   \
        if ((rc_lds = ensure_lds(p, (const void*)k_pass_a3<NS, LI, C, PM>, lds))) return rc_lds;                          \
        hipLaunchKernelGGL((k_pass_a3<NS, LI, C, PM>), dim3(gx.x, (GY)), dim3(256), lds, st, descs, p->work, p->N2,        \
                           (long long)p->N, tw, p->tbR, p->tsR, p->thR, p->twn1, p->log2CL, xf_per_pair, slots_per_pair,  \
                           nt, flags, PM == 0 ? row_sel : 0);                                                              \
    } while (0)
    if (!paired)
        FFS_A3_LAUNCH(0, row_sel ? groups_y * (row_sel >> 16) : n_xf);
    else if (xf_per_pair == 2)
        FFS_A3_LAUNCH(1, groups_y);
    else
        FFS_A3_LAUNCH(2, groups_y * (xf_per_pair - 1));
#undef FFS_A3_LAUNCH
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}

template <int NS, int LI, int C>
int launch_pass_c3_inst(const ffs_plan* p, const CandDesc* cands, int first_cand, int n_cand, int n_packed, int n_slots,
                        int n_pairs, int half_last, hipStream_t st) {
    const size_t lds = (size_t)LI * C * sizeof(cf);
    int rc_lds;
    const cf* tw = NS == 2 ? p->tw1h : p->tw1;
    const int tiles = p->N2 / C;
    const int plain = n_packed - (half_last ? 1 : 0);  // slots with two candidates (or all of them without HALF_LAST)
    if (plain > 0) {
        if ((rc_lds = ensure_lds(p, (const void*)k_pass_c3<NS, LI, C, false>, lds))) return rc_lds;
        hipLaunchKernelGGL((k_pass_c3<NS, LI, C, false>), dim3(tiles, n_pairs * plain), dim3(256), lds, st, p->work, p->N2,
                           (long long)p->N, tw, cands, first_cand, n_cand, n_packed, n_slots, p->bnom, p->log2CL, p->twn1, plain);
    }
    if (half_last) {  // the single-candidate slot: two real output columns per complex column transform
        if ((rc_lds = ensure_lds(p, (const void*)k_pass_c3<NS, LI, C, true>, lds))) return rc_lds;
        hipLaunchKernelGGL((k_pass_c3<NS, LI, C, true>), dim3(tiles / 2, n_pairs), dim3(256), lds, st, p->work, p->N2,
                           (long long)p->N, tw, cands, first_cand, n_cand, n_packed, n_slots, p->bnom, p->log2CL, p->twn1, 1);
    }
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}
int launch_pass_c3(const ffs_plan* p, const CandDesc* cands, int first_cand, int n_cand, int n_packed, int n_slots,
                   int n_pairs, int half_last, hipStream_t st) {
    switch (p->N1) {
        case 192: return launch_pass_c3_inst<3, 64, 64>(p, cands, first_cand, n_cand, n_packed, n_slots, n_pairs, half_last, st);
        case 384: return launch_pass_c3_inst<3, 128, 32>(p, cands, first_cand, n_cand, n_packed, n_slots, n_pairs, half_last, st);
        case 768: return launch_pass_c3_inst<3, 256, 16>(p, cands, first_cand, n_cand, n_packed, n_slots, n_pairs, half_last, st);
        case 512: return launch_pass_c3_inst<2, 256, 16>(p, cands, first_cand, n_cand, n_packed, n_slots, n_pairs, half_last, st);
    }
    return fail(FFS_E_INVALID, "unsupported column length %d", p->N1);
}

template <int DT>
int launch_pass_a(const ffs_plan* p, const XformDesc* descs, int n_xf, int xf_per_pair, int slots_per_pair, int ref_half,
                  hipStream_t st, int row_sel = 0) {
    if (DT == 2 && col3r_ok(p)) {
        switch (p->N1) {
            case 192: return launch_pass_a3_inst<3, 64, 64>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
            case 384: return launch_pass_a3_inst<3, 128, 32>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
            case 768: return launch_pass_a3_inst<3, 256, 16>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
            case 512: return launch_pass_a3_inst<2, 256, 16>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        }
    }
    switch (p->N1) {
        case 48: return launch_pass_a_inst<48, 64, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 96: return launch_pass_a_inst<96, 32, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 192: return launch_pass_a_inst<192, 16, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 384: return launch_pass_a_inst<384, 16, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 768: return launch_pass_a_inst<768, 16, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 16: return launch_pass_a_inst<16, 256, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 32: return launch_pass_a_inst<32, 128, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 64: return launch_pass_a_inst<64, 64, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 128: return launch_pass_a_inst<128, 32, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 256: return launch_pass_a_inst<256, 16, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 512: return launch_pass_a_inst<512, 16, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 1024: return launch_pass_a_inst<1024, 16, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 2048: return launch_pass_a_inst<2048, 8, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
        case 4096: return launch_pass_a_inst<4096, 4, DT>(p, descs, n_xf, xf_per_pair, slots_per_pair, ref_half, st, row_sel);
    }
    return fail(FFS_E_INVALID, "unsupported column length %d", p->N1);
}

template <int L, bool SEP>
int launch_mid_inst(const ffs_plan* p, int n_pairs, int n_slots, int ref_half, hipStream_t st) {
    const size_t lds = row_lds_bytes(L);
    int rc_lds;
    if ((rc_lds = ensure_lds(p, (const void*)k_mid<L, SEP>, lds))) return rc_lds;
    constexpr int ROWS = 256 / (L / 16);
    dim3 grid(p->N1 / ROWS, n_pairs);
    hipLaunchKernelGGL((k_mid<L, SEP>), grid, dim3(256), lds, st, p->work, p->N1, p->log2CL, (long long)p->N, n_slots,
                       (float)(1.0 / (double)p->N), p->tw2, p->tbM, p->tsM, ref_half);
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}

// the reference slot may hold only its rows 0..N1/2 (see k_mid) when one block handles one row
bool ref_half_ok(const ffs_plan* p) { return p->N2 == 4096 && (p->N2 / 16) >= (1 << p->log2CL) && p->N1 % 2 == 0; }

int launch_mid(const ffs_plan* p, int n_pairs, int n_slots, int ref_half, hipStream_t st) {
    const bool sep = (p->N2 / 16) >= (1 << p->log2CL);
    if (p->N2 == 4096 && sep) {  // one row per block: candidate rows requested one item ahead
        const size_t lds = row_lds_bytes(4096);
        int rc_lds;
        if ((rc_lds = ensure_lds(p, (const void*)k_mid<4096, true, true>, lds))) return rc_lds;
        hipLaunchKernelGGL((k_mid<4096, true, true>), dim3(p->N1, n_pairs), dim3(256), lds, st, p->work, p->N1, p->log2CL,
                           (long long)p->N, n_slots, (float)(1.0 / (double)p->N), p->tw2, p->tbM, p->tsM, ref_half);
        HIP_TRY(hipGetLastError());
        return FFS_OK;
    }
#define FFS_MID(L) \
    case L: return sep ? launch_mid_inst<L, true>(p, n_pairs, n_slots, ref_half, st) : launch_mid_inst<L, false>(p, n_pairs, n_slots, ref_half, st)
    switch (p->N2) {
        FFS_MID(256);
        FFS_MID(512);
        FFS_MID(1024);
        FFS_MID(2048);
        FFS_MID(4096);
    }
#undef FFS_MID
    return fail(FFS_E_INVALID, "unsupported row length %d", p->N2);
}

int launch_mid_seg(const ffs_plan* sp, int n_pairs, int n_slots, int n_blocks, int ref_half, hipStream_t st) {
    int rc_lds;
    const size_t lds = row_lds_bytes(4096);
    const size_t ldsp = row_lds_bytes(4096) + 4096 * sizeof(cf);  // + conj(R_k)/N parked next to the row buffer
    const float inv_n = (float)(1.0 / (double)sp->N);
    const int flags = ref_half | PAIR_ROWS;
    const int mid_lab = sp->lab_flags & (DBG_NO_FFT | DBG_HOT_MEM | DBG_NO_STORE);
    if (n_slots - 1 == 1) {  // one packed slot (every FFTAligner.fit): one accumulator row, three blocks per CU
        if ((rc_lds = ensure_lds(sp, (const void*)k_mid_seg_one<4096, false, 1>, lds))) return rc_lds;
        hipLaunchKernelGGL((k_mid_seg_one<4096, false, 1>), dim3(sp->N1, n_pairs), dim3(256), lds, st, sp->work, sp->N1,
                           sp->log2CL, (long long)sp->N, n_slots, n_blocks, inv_n, sp->tw2, sp->tbM, sp->tsM, flags | mid_lab);
    } else if (n_slots - 1 <= 4) {  // single sweep: all (<= 4) candidate slots accumulate at once
        if ((rc_lds = ensure_lds(sp, (const void*)k_mid_seg_one<4096, true>, ldsp))) return rc_lds;
        hipLaunchKernelGGL((k_mid_seg_one<4096, true>), dim3(sp->N1, n_pairs), dim3(256), ldsp, st, sp->work, sp->N1,
                           sp->log2CL, (long long)sp->N, n_slots, n_blocks, inv_n, sp->tw2, sp->tbM, sp->tsM, flags | mid_lab);
    } else {  // nine or more candidates: two slots per sweep
        if ((rc_lds = ensure_lds(sp, (const void*)k_mid_seg_pipe<4096>, ldsp))) return rc_lds;
        hipLaunchKernelGGL((k_mid_seg_pipe<4096>), dim3(sp->N1, n_pairs), dim3(256), ldsp, st, sp->work, sp->N1, sp->log2CL,
                           (long long)sp->N, n_slots, n_blocks, inv_n, sp->tw2, sp->tbM, sp->tsM, flags);
    }
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}

struct PoolArgs {
    const NomList* noms;
    PoolHeader* header;
    PoolEntry* entries;
    PoolBest* best;
    int shares;  // sub-batches of the call that share the pool (each flagged candidate's quota is divided by it)
    int half_last = 0;  // HALF_LAST layout of the last candidate slot (see ffs_kernels.h)
};

template <int L, int C, int MODE>
int launch_pass_c_inst(const ffs_plan* p, const CandDesc* cands, int first_cand, int n_cand, int n_packed, int n_slots,
                       int n_pairs, float* out_a, float* out_b, const PoolArgs& pa, hipStream_t st) {
    const size_t lds = col_lds_bytes(L);
    int rc_lds;
    if ((rc_lds = ensure_lds(p, (const void*)k_pass_c<L, C, MODE>, lds))) return rc_lds;
    dim3 grid(p->N2 / C, MODE == 2 ? kCollectRows : n_pairs * n_packed);
    hipLaunchKernelGGL((k_pass_c<L, C, MODE>), grid, dim3((L / 16) * C), lds, st, p->work, p->N2, (long long)p->N, p->tw1,
                       cands, first_cand, n_cand, n_packed, n_slots, p->bnom, out_a, out_b, pa.noms, pa.header, pa.entries, p->log2CL,
                       p->twn1, p->xlist, pa.best, pa.shares, pa.half_last);
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}

template <int MODE>
int launch_pass_c(const ffs_plan* p, const CandDesc* cands, int first_cand, int n_cand, int n_packed, int n_slots,
                  int n_pairs, float* out_a, float* out_b, const PoolArgs& pa, hipStream_t st) {
#define FFS_PC(L, C) \
    case L: return launch_pass_c_inst<L, C, MODE>(p, cands, first_cand, n_cand, n_packed, n_slots, n_pairs, out_a, out_b, pa, st)
    switch (p->N1) {
        FFS_PC(48, 64);
        FFS_PC(96, 32);
        FFS_PC(192, 16);
        FFS_PC(384, 16);
        FFS_PC(768, 16);
        FFS_PC(16, 256);
        FFS_PC(32, 128);
        FFS_PC(64, 64);
        FFS_PC(128, 32);
        FFS_PC(256, 16);
        FFS_PC(512, 16);
        FFS_PC(1024, 16);
        FFS_PC(2048, 8);
        FFS_PC(4096, 4);
    }
#undef FFS_PC
    return fail(FFS_E_INVALID, "unsupported column length %d", p->N1);
}

size_t pruned_lds_bytes(int L) {
    const int C = tile_cols(L), LT = L / 16;
    return 1024 + (size_t)L * sizeof(cf) + (size_t)MAXBINS * LT * C * sizeof(cf);
}

template <int L, int C, bool EXH>
int launch_pass_c_pruned_inst(const ffs_plan* p, const CandDesc* cands, int first_cand, int n_cand, int n_packed,
                              int n_slots, int n_pairs, const BinList& bins, const PoolArgs& pa, hipStream_t st, int seg,
                              int seg_shift) {
    const size_t lds = pruned_lds_bytes(L);
    int rc_lds;
    if ((rc_lds = ensure_lds(p, (const void*)k_pass_c_pruned<L, C, EXH>, lds))) return rc_lds;
    dim3 grid(p->N2 / C, EXH ? kCollectRows : n_pairs * n_packed);
    hipLaunchKernelGGL((k_pass_c_pruned<L, C, EXH>), grid, dim3((L / 16) * C), lds, st, p->work, p->N2, (long long)p->N,
                       p->twn1, cands, first_cand, n_cand, n_packed, n_slots, p->bnom, bins, pa.noms, pa.header, pa.entries, p->log2CL,
                       p->xlist, seg, seg_shift, pa.best, pa.shares, pa.half_last);
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}

template <bool EXH>
int launch_pass_c_pruned(const ffs_plan* p, const CandDesc* cands, int first_cand, int n_cand, int n_packed, int n_slots,
                         int n_pairs, const BinList& bins, const PoolArgs& pa, hipStream_t st, int seg = 0, int seg_shift = 0) {
#define FFS_PCP(L, C) \
    case L: return launch_pass_c_pruned_inst<L, C, EXH>(p, cands, first_cand, n_cand, n_packed, n_slots, n_pairs, bins, pa, st, seg, seg_shift)
    switch (p->N1) {
        FFS_PCP(48, 64);
        FFS_PCP(96, 32);
        FFS_PCP(192, 16);
        FFS_PCP(384, 16);
        FFS_PCP(768, 16);
        FFS_PCP(16, 256);
        FFS_PCP(32, 128);
        FFS_PCP(64, 64);
        FFS_PCP(128, 32);
        FFS_PCP(256, 16);
        FFS_PCP(512, 16);
        FFS_PCP(1024, 16);
        FFS_PCP(2048, 8);
        FFS_PCP(4096, 4);
    }
#undef FFS_PCP
    return fail(FFS_E_INVALID, "unsupported column length %d", p->N1);
}

// Output bins m2 = m / N2 of the last pass that the lag window [d_lo, d_hi] of a candidate touches
// (m = d for d >= 0, m = d + N for d < 0), as signed offsets in (-N1/2, N1/2].
void add_bins(const ffs_plan* p, const CandDesc& cd, std::vector<int>* bins, bool* overflow) {
    if (cd.flags & CAND_NO_LAGS) return;
    auto add_range = [&](int64_t m_lo, int64_t m_hi) {
        for (int64_t m2 = m_lo / p->N2; m2 <= m_hi / p->N2; ++m2) {
            int b = (int)m2;
            if (b > p->N1 / 2) b -= p->N1;
            bool seen = false;
            for (int x : *bins) seen |= (x == b);
            if (!seen) {
                if ((int)bins->size() >= MAXBINS) {
                    *overflow = true;
                    return;
                }
                bins->push_back(b);
            }
        }
    };
    if (cd.d_hi >= 0) add_range(cd.d_lo > 0 ? cd.d_lo : 0, cd.d_hi);
    if (!*overflow && cd.d_lo < 0) add_range(p->N + cd.d_lo, p->N + (cd.d_hi < -1 ? cd.d_hi : -1));
}

// Python slice semantics of  x[:stop] = v  /  x[start:] = v  on a length-n array
int64_t py_clamp(int64_t i, int64_t n) {
    if (i < 0) i += n;
    if (i < 0) i = 0;
    if (i > n) i = n;
    return i;
}

double mapped(double x) { return 2.0 * x - 1.0; }  // aligners.py:55-57

struct VecView {
    const void* ptr;
    int64_t len;
    double lo, hi;
    int64_t lead = 0;  // positions [0, lead) of the transform input are zero padding (see XformDesc)
    int64_t off = 0;   // bit-packed vectors: bit index of sample 0 relative to ptr (see XformDesc)
};

// The reference's lag window for (R, S): allowed k in [kA, kB) of its length-n_ref convolve array
// (aligners.py:31-43), returned as inclusive lags [d_lo, d_hi] (d = n_ref-1-S-k); false if empty.
bool lag_window(int64_t R, int64_t S, int64_t n_ref, int64_t max_off, int64_t* d_lo, int64_t* d_hi) {
    int64_t kA = 0, kB = n_ref;
    if (max_off >= 0) {
        kA = py_clamp(n_ref - 1 - max_off - S, n_ref);
        kB = py_clamp(n_ref - 1 + max_off - S, n_ref);
    }
    if (kA >= kB) return false;
    *d_hi = n_ref - 1 - S - kA;
    *d_lo = n_ref - S - kB;
    return true;
}

// What the transform pipeline has to evaluate of that window: only lags with a non-empty overlap,
// d in (-S, R).  Every other lag of the window has c(d) = 0 exactly and is represented by the single
// virtual nominee (0, d_zero) -- d_zero = the largest such lag, i.e. the first such k (see CAND_HAS_ZERO
// in ffs_kernels.h).  Returns false when the reference's window is empty (everything -inf);
// *d_lo > *d_hi when the window holds only zero-overlap lags.
bool overlap_window(int64_t R, int64_t S, int64_t n_ref, int64_t max_off, int64_t* d_lo, int64_t* d_hi, bool* has_zero,
                    int64_t* d_zero) {
    int64_t wl = 0, wh = -1;
    *has_zero = false;
    *d_zero = 0;
    if (!lag_window(R, S, n_ref, max_off, &wl, &wh)) return false;
    if (wh >= R) {
        *has_zero = true;
        *d_zero = wh;
    } else if (wl <= -S) {
        *has_zero = true;
        *d_zero = -S;
    }
    *d_lo = wl > -S + 1 ? wl : -S + 1;
    *d_hi = wh < R - 1 ? wh : R - 1;
    return true;
}

// Samples that can meet the other vector at some lag of the window: s[i] is multiplied by r[i+d], which
// is zero padding unless i + d < R, so only i < R - d_lo of the candidate and j < S + d_hi of the
// reference ever contribute.  The transforms read just these prefixes.
void effective_lengths(int64_t R, int64_t S, int64_t d_lo, int64_t d_hi, int64_t* s_eff, int64_t* r_eff) {
    int64_t se = R - d_lo < S ? R - d_lo : S;
    int64_t re = S + d_hi < R ? S + d_hi : R;
    *s_eff = se < 1 ? 1 : se;
    *r_eff = re < 1 ? 1 : re;
}

int64_t next_pow2(int64_t x) {
    int64_t n = 2;
    while (n < x) n <<= 1;
    return n;
}

// Transform length a (R, S) pair needs under its lag window [d_lo, d_hi] (ffs_plan_length without the window arithmetic
// the caller has already done, and without reading the environment).
int64_t plan_length_core(int64_t R, int64_t S, int64_t n_ref, int64_t d_lo, int64_t d_hi, bool radix3_off) {
    int64_t s_eff, r_eff;
    effective_lengths(R, S, d_lo, d_hi, &s_eff, &r_eff);
    int64_t need = s_eff + d_hi + 1;
    if (r_eff - d_lo + 1 > need) need = r_eff - d_lo + 1;
    if (s_eff > need) need = s_eff;
    if (r_eff > need) need = r_eff;
    int64_t n = next_pow2(need);
    if (!radix3_off && n / 4 * 3 >= need && radix3_length(n / 4 * 3)) n = n / 4 * 3;
    return n < n_ref ? n : n_ref;
}

// Fill the candidate descriptor for (ref, sub); returns a negative code on error -- including FFS_E_TOO_LONG when the
// plan's transform length cannot hold the pair, on EVERY path (the run-boundary kernels would not need the transform,
// but whether a call is accepted must not depend on how dense its vectors are).  The fp32 tie margin, which only the
// transform path needs, is added by finish_cand_for_transforms.
int fill_cand(const ffs_plan* p, const VecView& ref, const VecView& sub, int64_t max_off, CandDesc* cd, int ref_dt, int cand_dt) {
    const int64_t R = ref.len, S = sub.len;
    if (R <= 0 || S <= 0)
        return fail(FFS_E_EMPTY, "cannot align empty speech data (reference length=%lld, subtitle length=%lld)",
                    (long long)R, (long long)S);
    const int64_t n_ref = ffs_fft_length(R, S);
    memset(cd, 0, sizeof *cd);
    cd->s = sub.ptr;
    cd->r = ref.ptr;
    cd->S = (int32_t)S;
    cd->R = (int32_t)R;
    cd->n_ref = (int32_t)n_ref;
    int64_t d_lo = 0, d_hi = -1, d_zero = 0;
    bool has_zero = false;
    if (!overlap_window(R, S, n_ref, max_off, &d_lo, &d_hi, &has_zero, &d_zero) || d_lo > d_hi) {
        cd->flags = CAND_NO_LAGS;
        cd->d_lo = 0;
        cd->d_hi = -1;
    } else {
        cd->d_lo = (int32_t)d_lo;
        cd->d_hi = (int32_t)d_hi;
        if (!p->direct_only) {
            const int64_t n_need = plan_length_core(R, S, n_ref, d_lo, d_hi, p->radix3_off);
            if (n_need > p->N)
                return fail(FFS_E_TOO_LONG, "R=%lld S=%lld needs transform length %lld > plan length %lld", (long long)R,
                            (long long)S, (long long)n_need, (long long)p->N);
        }
    }
    if (has_zero) {
        cd->flags |= CAND_HAS_ZERO;
        cd->d_zero = (int32_t)d_zero;
    }
    cd->flags |= (cand_dt << CAND_DTS_SHIFT) | (ref_dt << CAND_DTR_SHIFT);
    cd->s0 = mapped(sub.lo);
    cd->s1 = mapped(sub.hi);
    cd->r0 = mapped(ref.lo);
    cd->r1 = mapped(ref.hi);
    return FFS_OK;
}

int finish_cand_for_transforms(const ffs_plan* p, int64_t max_off, CandDesc* cd) {
    (void)max_off;
    const int64_t R = cd->R, S = cd->S;
    const double as = fmax(fabs(cd->s0), fabs(cd->s1)), ar = fmax(fabs(cd->r0), fabs(cd->r1));
    const double lg = (double)ilog2(p->N > 2 ? p->N : 2);
    // measured max fp32 error of the pipeline: ~0.02 (activity density 0.5) to ~0.2 (density 0.2) of
    // eps*log2(N)*sqrt(S*R)*|s||r| for N = 2^12..2^24; lags within 1.0x of it are nominated, and the
    // kernels widen this to 24 ulp of the maximum for DC-heavy signals (eff_margin)
    cd->margin = (float)(1.0 * 5.9604645e-08 * lg * sqrt((double)S * (double)R) * as * ar);
    return FFS_OK;
}

void fill_xform(XformDesc* x, const VecView* a, const VecView* b, const void* safe = nullptr) {
    memset(x, 0, sizeof *x);
    x->a = a->ptr;
    x->len_a = (int32_t)a->len;
    x->lead_a = (int32_t)a->lead;
    x->off_a = (int32_t)a->off;
    x->a0 = (float)mapped(a->lo);
    x->a1 = (float)mapped(a->hi);
    x->b = safe ? safe : a->ptr;  // absent second candidate: any valid address with length 0 (reads are clamped)
    if (b) {
        x->b = b->ptr;
        x->len_b = (int32_t)b->len;
        x->lead_b = (int32_t)b->lead;
        x->off_b = (int32_t)b->off;
        x->b0 = (float)mapped(b->lo);
        x->b1 = (float)mapped(b->hi);
    }
}

}  // namespace

extern "C" {

const char* ffs_last_error(void) { return g_err.c_str(); }
int ffs_version(void) { return 300; }

int64_t ffs_fft_length(int64_t ref_len, int64_t sub_len) {
    if (ref_len <= 0 || sub_len <= 0) return 0;
    // Between powers of two the reference's float expression (below) and the integer ceil(log2) agree -- log2 of
    // 2^k +- 1 differs from k by more than 1e-7 relative up to 2^24, a thousand times the rounding error of the quotient --
    // so only exact powers of two (where log(x)/log(2) can land a hair above the integer) take the float path.
    const int64_t x = ref_len + sub_len;
    if ((x & (x - 1)) != 0 && x < (int64_t(1) << 40)) return int64_t(1) << (64 - __builtin_clzll((unsigned long long)x));
    // aligners.py:67-68: int(2 ** math.ceil(math.log(R + S, 2))) -- math.log(x, 2) is
    // log(x)/log(2) in doubles, which is not exact at every power of two; keep the same quirk.
    const double bits = log((double)(ref_len + sub_len)) / log(2.0);
    return (int64_t)pow(2.0, ceil(bits));
}

int64_t ffs_plan_length(int64_t ref_len, int64_t sub_len, int64_t max_offset_samples) {
    const int64_t n_ref = ffs_fft_length(ref_len, sub_len);
    if (n_ref == 0) return n_ref;
    int64_t d_lo, d_hi, d_zero;
    bool has_zero;
    if (!overlap_window(ref_len, sub_len, n_ref, max_offset_samples, &d_lo, &d_hi, &has_zero, &d_zero) || d_lo > d_hi)
        return 2;  // nothing to evaluate
    // lags d in [d_lo, d_hi] of a length-n circular correlation equal the linear ones iff no product
    // wraps: S' + d_hi <= n and R' - d_lo <= n for the prefixes (S', R') that reach the window at all
    int64_t s_eff, r_eff;
    effective_lengths(ref_len, sub_len, d_lo, d_hi, &s_eff, &r_eff);
    int64_t need = s_eff + d_hi + 1;
    if (r_eff - d_lo + 1 > need) need = r_eff - d_lo + 1;
    if (s_eff > need) need = s_eff;
    if (r_eff > need) need = r_eff;
    int64_t n = next_pow2(need);
    // three quarters of that is enough for many inputs (a 2 h pair at 100 Hz needs 726 001 points:
    // 3 * 2^18 = 786 432 instead of 2^20), and the column passes handle one factor of three
    const char* e = getenv("FFS_DISABLE_RADIX3");  // (read per call: only the transform path asks for plan lengths)
    if (!(e && e[0] == '1') && n / 4 * 3 >= need && radix3_length(n / 4 * 3)) n = n / 4 * 3;
    return n < n_ref ? n : n_ref;
}

int ffs_plan_create(int device, int64_t n_fft, int pairs_in_flight, int max_cand, ffs_plan** out) {
    if (!out) return fail(FFS_E_INVALID, "out is null");
    *out = nullptr;
    const bool r3 = radix3_length(n_fft);
    if (n_fft < 2 || n_fft > kMaxFftN || ((n_fft & (n_fft - 1)) && !r3))
        return fail(FFS_E_INVALID, "n_fft must be a power of two in [2, 2^24] or 3*2^k in [12288, 3145728], got %lld",
                    (long long)n_fft);
    if (pairs_in_flight < 1 || max_cand < 1) return fail(FFS_E_INVALID, "pairs_in_flight and max_cand must be >= 1");
    HIP_TRY(hipSetDevice(device));
    ffs_plan* p = new ffs_plan();
    struct Guard {  // destroy the half-built plan on any early return
        ffs_plan* p;
        ~Guard() {
            if (p) ffs_plan_destroy(p);
        }
    } guard{p};
    p->device = device;
    {
        const char* e3 = getenv("FFS_DISABLE_RADIX3");
        p->radix3_off = e3 && e3[0] == '1';
    }
    p->N = n_fft;
    p->pairs_in_flight = pairs_in_flight;
    p->max_cand = max_cand;
    p->max_slots = 1 + (max_cand + 1) / 2;
    HIP_TRY(hipEventCreateWithFlags(&p->upload_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p->last_done, hipEventDisableTiming));
    for (int h = 0; h < 2; ++h) HIP_TRY(hipEventCreateWithFlags(&p->half_done[h], hipEventDisableTiming));
    {
        // Run-time knobs: five, all listed in INTEGRATION.md section 6, each selecting between code paths that
        // return identical records (every pairing has an "identical records" GPU test).
        auto on = [](const char* name) {
            const char* e = getenv(name);
            return e && e[0] == '1';
        };
        p->allow_pruned = !on("FFS_DISABLE_PRUNED_PASS_C");
        p->allow_seg = !on("FFS_DISABLE_SEGMENTED");
        p->allow_half_last = !on("FFS_DISABLE_HALF_LAST");
        if (const char* ea = getenv("FFS_ALGORITHM")) {  // auto (default) | fft | runs
            if (!strcmp(ea, "fft")) p->algo = FFS_ALGO_FFT;
            else if (!strcmp(ea, "runs")) p->algo = FFS_ALGO_RUNS;
        }
        // Measured break-even (profiles/r04_runs_experiments.json, `boundary_density` / `windowless` in the bench line):
        // k_runs_corr spends 1.2-1.6e-4 us*CU per boundary coincidence, the transform pipeline 6-9e-4 us*CU per point of
        // the plan length and packed transform slot -> eight coincidences per transform point and slot; a candidate's
        // share of the slots is (n_cand + 1) / (2 n_cand) (two candidates per complex transform + the reference's half
        // slot): 3.6 M for seven candidates on the window-shortened 2 h plan, 7.2 M on the windowless one.  Either path is
        // within ~10 % of the other around the threshold.  FFS_RUNS_BUDGET=<coincidences> overrides it.
        p->runs_budget = -1;
        if (const char* eb = getenv("FFS_RUNS_BUDGET")) p->runs_budget = atoll(eb);
        if (const char* es = getenv("FFS_RUNS_SPLIT")) p->runs_split = atoi(es);
        p->runs_stride = kRunsStride0;
        if (const char* es = getenv("FFS_RUNS_STRIDE")) {
            const int v = atoi(es);
            p->runs_stride = v < 64 ? 64 : (v > RUNS_CAP ? RUNS_CAP : v);
        }
        if (const char* et = getenv("FFS_HOST_TIMING")) p->host_timing = et[0] == '1';
        const char* e3 = getenv("FFS_PASS_A_PREFETCH");
        if (e3) p->pass_a_prefetch = p->pass_a_prefetch_bits = atoi(e3);
#ifdef FFS_LAB
        // section experiments of the lab build (timing only: these switches produce WRONG RESULTS)
        const char* e20 = getenv("FFS_PASS_A_DEBUG");
        if (e20) p->lab_flags |= (atoi(e20) & 31) << 10;
        const char* e15 = getenv("FFS_MID_DEBUG");
        if (e15) p->lab_flags |= ((atoi(e15) & 1) ? DBG_NO_FFT : 0) | ((atoi(e15) & 2) ? DBG_HOT_MEM : 0) | ((atoi(e15) & 4) ? DBG_NO_STORE : 0);
#endif
    }
    if (n_fft < kMinFftN) {
        p->direct_only = true;
        guard.p = nullptr;
        *out = p;
        return FFS_OK;
    }
    if (r3) {
        p->N2 = n_fft >= 48 * 4096 ? 4096 : (int)(n_fft / 48);
        p->N1 = (int)(n_fft / p->N2);
    } else {
        const int lg = ilog2(n_fft);
        const int lg2 = lg - 4 < 12 ? lg - 4 : 12;
        p->N2 = 1 << lg2;
        p->N1 = (int)(n_fft >> lg2);
    }
    p->C = tile_cols(p->N1);
    p->log2C = ilog2(p->C);
    p->log2CL = p->log2C < 6 ? 6 : p->log2C;
    if ((1 << p->log2CL) > p->N2) p->log2CL = ilog2(p->N2);
#ifdef FFS_LAB
    if (const char* ecl = getenv("FFS_LOG2CL")) {  // wider row chunks (timing experiments of the lab build)
        const int v = atoi(ecl);
        if (v >= p->log2C && (1 << v) <= p->N2 / 16) p->log2CL = v;
    }
#endif
    const int64_t N = n_fft;
    const int N1 = p->N1, N2 = p->N2, LT1 = N1 / 16, LT2 = N2 / 16;
    int rc;
    const int LI1 = r3 ? N1 / 3 : N1;  // power-of-two part of the column length
    if ((rc = upload(&p->tw1, make_stage_tables(LI1), &p->workspace_bytes))) return rc;
    if ((rc = upload(&p->tw2, make_stage_tables(N2), &p->workspace_bytes))) return rc;
    if (N1 == 512 && (rc = upload(&p->tw1h, make_stage_tables(256), &p->workspace_bytes))) return rc;  // sub-transforms of k_pass_a3/c3<2, 256>
    {
        // Power-of-two columns: thread u holds the outputs k1 = u + LT1*q.  3*2^k columns (k_pass_a's
        // radix-3 branch): store thread rg = r*KG + kg holds k1 = kg + LI*r + KG*j, KG = 2*(LI/16).
        const int KG = 2 * (LI1 / 16);
        const int nb = r3 ? 3 * KG : LT1, ostep = r3 ? KG : LT1;
        std::vector<cf> tb((size_t)nb * N2), ts((size_t)16 * N2);
        for (int u = 0; u < nb; ++u) {
            const int ob = r3 ? (u % KG) + LI1 * (u / KG) : u;
            for (int n2 = 0; n2 < N2; ++n2) tb[(size_t)u * N2 + n2] = wn(N, (int64_t)n2 * ob);
        }
        for (int q = 0; q < 16; ++q)
            for (int n2 = 0; n2 < N2; ++n2) ts[(size_t)q * N2 + n2] = wn(N, (int64_t)n2 * ostep * q);
        if ((rc = upload(&p->tbA, tb, &p->workspace_bytes))) return rc;
        if ((rc = upload(&p->tsA, ts, &p->workspace_bytes))) return rc;
    }
    {
        std::vector<cf> tb((size_t)N1 * LT2), ts((size_t)N1 * 16);
        for (int k1 = 0; k1 < N1; ++k1) {
            for (int u = 0; u < LT2; ++u) tb[(size_t)k1 * LT2 + u] = wn(N, (int64_t)k1 * u);
            for (int q = 0; q < 16; ++q) ts[(size_t)k1 * 16 + q] = wn(N, (int64_t)k1 * LT2 * q);
        }
        if ((rc = upload(&p->tbM, tb, &p->workspace_bytes))) return rc;
        if ((rc = upload(&p->tsM, ts, &p->workspace_bytes))) return rc;
    }
    if ((r3 && LI1 >= 64) || N1 == 512) {  // k_pass_a3: k1 = u + LTI*q + LI*r (three sub-transforms; two for N1 = 512)
        const int LIr = r3 ? LI1 : 256;
        const int LTI = LIr / 16;
        std::vector<cf> tb((size_t)LTI * N2), ts((size_t)4 * N2), th((size_t)2 * N2);
        for (int n2 = 0; n2 < N2; ++n2) {
            for (int u = 0; u < LTI; ++u) tb[(size_t)u * N2 + n2] = wn(N, (int64_t)n2 * u);
            for (int i = 0; i < 4; ++i) ts[(size_t)i * N2 + n2] = wn(N, (int64_t)n2 * LTI * (1 << i));
            for (int r = 1; r <= 2; ++r) th[(size_t)(r - 1) * N2 + n2] = wn(N, (int64_t)n2 * LIr * r);
        }
        if ((rc = upload(&p->tbR, tb, &p->workspace_bytes))) return rc;
        if ((rc = upload(&p->tsR, ts, &p->workspace_bytes))) return rc;
        if ((rc = upload(&p->thR, th, &p->workspace_bytes))) return rc;
    }
    {
        std::vector<cf> t((size_t)N1);
        for (int k = 0; k < N1; ++k) t[k] = wn(N1, k);
        if ((rc = upload(&p->twn1, t, &p->workspace_bytes))) return rc;
    }
    // The transform workspace ([pairs_in_flight][max_slots][N] complex fp32 -- 15 GiB for 512 seven-ratio 2 h pairs --,
    // block-nominee records, the overflow pool) is allocated by the first call that runs the transforms
    // (ensure_workspace): calls served by the run-boundary path never touch it.  workspace_bytes counts it from the start.
    p->workspace_bytes += (int64_t)((size_t)pairs_in_flight * p->max_slots * N * sizeof(cf));
    p->workspace_bytes += (int64_t)((size_t)pairs_in_flight * (p->max_slots - 1) * 2 * (N2 / p->C) * sizeof(BlockNom));
    p->workspace_bytes += (int64_t)kPoolCapacity * sizeof(PoolEntry);
    if (r3 && p->allow_seg && p->N2 == 4096 && n_fft / 3 >= 65536) {
        if ((rc = ffs_plan_create(device, n_fft / 3, pairs_in_flight * kSegBlocks, max_cand, &p->seg))) return rc;
        p->workspace_bytes += p->seg->workspace_bytes;
    }
    guard.p = nullptr;
    *out = p;
    return FFS_OK;
}

int ffs_plan_destroy(ffs_plan* p) {
    if (!p) return FFS_OK;
    if (p->host_timing && p->runs_calls)
        fprintf(stderr, "[ffs host timing] %lld run-boundary calls: total %.1f us/call = vectors+extract launch %.1f, candidate "
                "descriptors %.1f, wait for list lengths %.1f, decision+rest %.1f\n", (long long)p->runs_calls,
                p->ht_total / p->runs_calls / 1e3, p->ht_vec / p->runs_calls / 1e3, p->ht_cand / p->runs_calls / 1e3,
                p->ht_wait / p->runs_calls / 1e3, p->ht_decide / p->runs_calls / 1e3);
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(p->tw1);
    (void)hipFree(p->tw2);
    (void)hipFree(p->tbA);
    (void)hipFree(p->tsA);
    (void)hipFree(p->tw1h);
    (void)hipFree(p->tbR);
    (void)hipFree(p->tsR);
    (void)hipFree(p->thR);
    (void)hipFree(p->tbM);
    (void)hipFree(p->tsM);
    (void)hipFree(p->twn1);
    (void)hipFree(p->work);
    (void)hipFree(p->bnom);
    (void)hipFree(p->pool_entries);
    (void)hipFree(p->xlist);
    if (p->seg) (void)ffs_plan_destroy(p->seg);
    (void)hipFree(p->runs_e);
    (void)hipFree(p->runs_n);
    (void)hipFree(p->runs_best);
    (void)hipFree(p->runs_flags);
    (void)hipFree(p->runs_zero_flags);
    (void)hipFree(p->pack_buf);
    (void)hipFree(p->lvl_buf);
    if (p->runs_flags_host) (void)hipHostFree(p->runs_flags_host);
    if (p->runs_ev) (void)hipEventDestroy(p->runs_ev);
    if (p->upload_done) (void)hipEventSynchronize(p->upload_done);  // (the shared copy stream may still be reading the pinned block)
    if (p->copy_ev) (void)hipEventDestroy(p->copy_ev);
    for (int h = 0; h < 2; ++h)
        if (p->half_done[h]) (void)hipEventDestroy(p->half_done[h]);
    (void)hipFree(p->dev_desc);
    (void)hipFree(p->dev_desc2);
    if (p->host_desc) (void)hipHostFree(p->host_desc);
    if (p->upload_done) (void)hipEventDestroy(p->upload_done);
    if (p->last_done) (void)hipEventDestroy(p->last_done);
    for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
    delete p;
    return FFS_OK;
}

int64_t ffs_plan_workspace_bytes(const ffs_plan* p) { return p ? p->workspace_bytes : 0; }

// Leaves the plan's stream bookkeeping consistent on every exit once work has been queued (an error return after a
// kernel launch must still record the call's last event: the next call on another stream waits for it).
struct StreamLeave {
    ffs_plan* p;
    hipStream_t st;
    bool armed = false;
    ~StreamLeave() {
        if (armed) (void)leave_stream(p, st);
    }
};

int grow_pack_buf(ffs_plan* p, size_t bytes) {
    if (bytes <= p->pack_bytes) return FFS_OK;
    if (p->has_last) HIP_TRY(hipEventSynchronize(p->last_done));
    (void)hipFree(p->pack_buf);
    p->pack_buf = nullptr;
    p->pack_bytes = 0;
    HIP_TRY(hipMalloc((void**)&p->pack_buf, bytes + bytes / 4));
    p->pack_bytes = bytes + bytes / 4;
    return FFS_OK;
}

// vec_bound (may be null): FFS_DTYPE_RUNS vectors only -- a host-known upper bound of the list's length (0: unknown).
static int align_impl(ffs_plan* p, int n_pairs, int n_cand, int ref_dt, int dtype, const void* const* vec_ptr,
                      const int64_t* vec_len, const double* vec_lo, const double* vec_hi, const int32_t* vec_bound,
                      int64_t max_offset_samples, int64_t filter_max_offset, ffs_cand_result* cand_out_dev,
                      ffs_pair_result* pair_out_dev, void* hip_stream) {
    // `dtype` = the candidates' element type, `ref_dt` = the references'; `mixed` when they differ: the first pass
    // runs once per role (row_sel) and the exact re-evaluation goes through the type-generic instantiations (DT 4)
    auto known = [](int dt) {
        return dt == FFS_DTYPE_U8 || dt == FFS_DTYPE_F32 || dt == FFS_DTYPE_U1 || dt == FFS_DTYPE_F64 || dt == FFS_DTYPE_RUNS;
    };
    auto esz_of = [](int dt) -> size_t { return dt == FFS_DTYPE_U8 ? 1 : (dt == FFS_DTYPE_F32 ? 4 : (dt == FFS_DTYPE_F64 ? 8 : 0)); };
    auto amask_of = [](int dt) -> uintptr_t { return dt == FFS_DTYPE_U8 ? 0 : (dt == FFS_DTYPE_F64 || dt == FFS_DTYPE_RUNS ? 7 : 3); };
    if (!p) return fail(FFS_E_INVALID, "plan is null");
    if (n_pairs < 0 || n_cand < 1 || n_cand > p->max_cand)
        return fail(FFS_E_INVALID, "n_cand=%d outside [1, plan max_cand=%d]", n_cand, p->max_cand);
    if (!known(dtype) || !known(ref_dt)) return fail(FFS_E_INVALID, "unknown dtype %d / %d", ref_dt, dtype);
    if (!vec_ptr || !vec_len || !vec_lo || !vec_hi || !cand_out_dev || !pair_out_dev)
        return fail(FFS_E_INVALID, "null argument");
    if (n_pairs == 0) return FFS_OK;
    const int stride = 1 + n_cand;
    const size_t n_vec = (size_t)n_pairs * (size_t)stride;
    // ---- one contract for every path, checked before anything is queued: no empty vector (aligners.py:58-66), no null
    // or misaligned pointer, no vector of 2^30 samples or more (positions are 32-bit; 0x3fffffff is the run-boundary
    // kernels' "beyond everything").  The plan-length check (FFS_E_TOO_LONG whichever path would have served the pair) is
    // part of fill_cand, which every path runs for every candidate.
    for (int pi = 0; pi < n_pairs; ++pi) {
        const size_t b = (size_t)pi * stride;
        for (int v = 0; v < stride; ++v) {
            const int dt = v ? dtype : ref_dt;
            if (vec_len[b] <= 0 || vec_len[b + v] <= 0)
                return fail(FFS_E_EMPTY, "cannot align empty speech data (reference length=%lld, subtitle length=%lld)",
                            (long long)vec_len[b], (long long)vec_len[b + (v ? v : 1)]);
            if (!vec_ptr[b + v]) return fail(FFS_E_INVALID, "null device pointer for pair %d", pi);
            if ((uintptr_t)vec_ptr[b + v] & amask_of(dt))
                return fail(FFS_E_INVALID, "float / bit-packed vectors and boundary lists must be aligned to their element (pair %d)", pi);
            if (vec_len[b + v] >= (int64_t(1) << 30))
                return fail(FFS_E_TOO_LONG, "vector of %lld samples (pair %d): at most 2^30 - 1", (long long)vec_len[b + v], pi);
        }
    }
    hipStream_t st = (hipStream_t)hip_stream;
    const bool lists_in = dtype == FFS_DTYPE_RUNS || ref_dt == FFS_DTYPE_RUNS;
    auto runs_able = [](int dt) { return dt == FFS_DTYPE_U1 || dt == FFS_DTYPE_RUNS; };
    if (p->algo != FFS_ALGO_FFT && !p->direct_only && dtype == FFS_DTYPE_U8 && ref_dt == FFS_DTYPE_U8) {
        // 0/1 BYTES (the north star's literal input format): one pass packs every vector of the call to bits (bit =
        // byte != 0, exactly the two-level reading the byte kernels apply), then the call continues as FFS_DTYPE_U1 --
        // run-boundary path where the lists are short, transforms on an eighth of the input bytes otherwise.  Identical
        // records (tests/test_gpu_headline.py::test_byte_inputs_give_the_same_records; FFS_ALGO_FFT keeps the byte kernels).
        std::vector<size_t> off(n_vec + 1, 0);
        int64_t len_max = 1;
        for (size_t i = 0; i < n_vec; ++i) {
            off[i + 1] = off[i] + (((size_t)(vec_len[i] + 31) / 32 * 4 + 63) & ~(size_t)63);
            if (vec_len[i] > len_max) len_max = vec_len[i];
        }
        HIP_TRY(hipSetDevice(p->device));
        int rc0;
        if ((rc0 = enter_stream(p, st))) return rc0;
        if ((rc0 = grow_pack_buf(p, off[n_vec]))) return rc0;
        if ((rc0 = ensure_desc(p, n_vec * sizeof(PackVec) + 4096))) return rc0;
        HIP_TRY(hipEventSynchronize(p->upload_done));
        p->cur_half = 0;
        PackVec* hp = (PackVec*)p->host_desc;
        std::vector<const void*> packed(n_vec);
        for (size_t i = 0; i < n_vec; ++i) {
            unsigned* dst = (unsigned*)((char*)p->pack_buf + off[i]);
            hp[i] = PackVec{(const unsigned char*)vec_ptr[i], dst, (int32_t)vec_len[i], 0};
            packed[i] = dst;
        }
        HIP_TRY(hipMemcpyAsync(p->dev_desc, hp, n_vec * sizeof(PackVec), hipMemcpyHostToDevice, st));
        HIP_TRY(hipEventRecord(p->upload_done, st));
        const int chunks_per_vec = (int)(((len_max + 31) / 32 + 255) / 256);
        hipLaunchKernelGGL(k_pack_bytes_batch, dim3((unsigned)(n_vec * chunks_per_vec)), dim3(256), 0, st, (const PackVec*)p->dev_desc,
                           chunks_per_vec);
        HIP_TRY(hipGetLastError());
        if ((rc0 = leave_stream(p, st))) return rc0;
        return align_impl(p, n_pairs, n_cand, FFS_DTYPE_U1, FFS_DTYPE_U1, packed.data(), vec_len, vec_lo, vec_hi, nullptr,
                          max_offset_samples, filter_max_offset, cand_out_dev, pair_out_dev, hip_stream);
    }
    // Expands the call's list-only vectors (role `ref`/`cand` of type FFS_DTYPE_RUNS) to bits in the plan's pack buffer;
    // `out` = the call's pointer table with those vectors replaced.  Queued on `st`.
    auto expand_lists = [&](std::vector<const void*>* out) -> int {
        std::vector<size_t> off(n_vec + 1, 0);
        int64_t len_max = 1;
        for (size_t i = 0; i < n_vec; ++i) {
            const bool is_list = ((i % stride) ? dtype : ref_dt) == FFS_DTYPE_RUNS;
            off[i + 1] = off[i] + (is_list ? (((size_t)(vec_len[i] + 31) / 32 * 4 + 63) & ~(size_t)63) : 0);
            if (is_list && vec_len[i] > len_max) len_max = vec_len[i];
        }
        int rc0;
        if ((rc0 = grow_pack_buf(p, off[n_vec]))) return rc0;
        ExpandVec* d_ev = nullptr;
        std::vector<ExpandVec> hev(n_vec);
        out->assign(vec_ptr, vec_ptr + n_vec);
        for (size_t i = 0; i < n_vec; ++i) {
            const bool is_list = ((i % stride) ? dtype : ref_dt) == FFS_DTYPE_RUNS;
            unsigned* dst = is_list ? (unsigned*)((char*)p->pack_buf + off[i]) : nullptr;
            hev[i] = ExpandVec{(const int2*)((const char*)vec_ptr[i] + 16), (const int2*)vec_ptr[i], dst, (int32_t)vec_len[i], 0};
            if (is_list) (*out)[i] = dst;
        }
        // (rare path; the table is followed by one int the kernel raises when a block's header does not fit the block)
        const size_t tab = n_vec * sizeof(ExpandVec);
        int bad_list = 0;
        HIP_TRY(hipMallocAsync((void**)&d_ev, tab + sizeof(int), st));
        hipError_t he = hipMemcpyAsync(d_ev, hev.data(), tab, hipMemcpyHostToDevice, st);
        if (he == hipSuccess) he = hipMemsetAsync((char*)d_ev + tab, 0, sizeof(int), st);
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        if (he == hipSuccess) {
            const int chunks_per_vec = (int)(((len_max + 31) / 32 + 255) / 256);
            hipLaunchKernelGGL(k_runs_expand, dim3((unsigned)(n_vec * chunks_per_vec)), dim3(256), 0, st, (const ExpandVec*)d_ev, chunks_per_vec,
                               (int*)((char*)d_ev + tab));
            he = hipGetLastError();
        }
        if (he == hipSuccess) he = hipMemcpyAsync(&bad_list, (char*)d_ev + tab, sizeof(int), hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipStreamSynchronize(st);
        const hipError_t hf = hipFreeAsync(d_ev, st);  // (on every path: an error above must not leak the table)
        HIP_TRY(he);
        HIP_TRY(hf);
        if (bad_list)
            return fail(FFS_E_INVALID, "a boundary list of the call is truncated (more entries than its block holds) or was made for "
                        "another vector length: nothing can solve it -- pass the vector as bits");
        return FFS_OK;
    };
    if (lists_in && (p->algo == FFS_ALGO_FFT || p->direct_only || !runs_able(dtype) || !runs_able(ref_dt))) {
        // boundary lists where only the transforms (or the direct kernel of short plans) may run: as bits
        HIP_TRY(hipSetDevice(p->device));
        int rc0;
        if ((rc0 = enter_stream(p, st))) return rc0;
        std::vector<const void*> bits_ptr;
        if ((rc0 = expand_lists(&bits_ptr))) return rc0;
        if ((rc0 = leave_stream(p, st))) return rc0;
        return align_impl(p, n_pairs, n_cand, ref_dt == FFS_DTYPE_RUNS ? FFS_DTYPE_U1 : ref_dt,
                          dtype == FFS_DTYPE_RUNS ? FFS_DTYPE_U1 : dtype, bits_ptr.data(), vec_len, vec_lo, vec_hi, nullptr,
                          max_offset_samples, filter_max_offset, cand_out_dev, pair_out_dev, hip_stream);
    }
    {
        // The plan-owned boundary lists take 256 KiB per vector: a call with more than 65 536 vectors of bit-packed
        // two-level samples is solved as consecutive sub-calls (results land where one call would put them).
        // (a multi-level float reference brings three threshold planes of its own)
        const bool ml_split = (ref_dt == FFS_DTYPE_F64 || ref_dt == FFS_DTYPE_F32) && runs_able(dtype);
        const int64_t per_pair = stride + (ml_split ? 3 : 0);
        const int64_t max_pairs = (int64_t(1) << 16) / per_pair > 0 ? (int64_t(1) << 16) / per_pair : 1;
        if (p->algo != FFS_ALGO_FFT && !p->direct_only && runs_able(dtype) && (runs_able(ref_dt) || ml_split) && n_pairs > max_pairs) {
            for (int64_t p0 = 0; p0 < n_pairs; p0 += max_pairs) {
                const int np = (int)((n_pairs - p0) < max_pairs ? (n_pairs - p0) : max_pairs);
                const int rc_sub = align_impl(p, np, n_cand, ref_dt, dtype, vec_ptr + p0 * stride, vec_len + p0 * stride,
                                              vec_lo + p0 * stride, vec_hi + p0 * stride, vec_bound ? vec_bound + p0 * stride : nullptr,
                                              max_offset_samples, filter_max_offset, cand_out_dev + p0 * n_cand, pair_out_dev + p0,
                                              hip_stream);
                if (rc_sub) return rc_sub;
            }
            return FFS_OK;
        }
    }
    static_assert(sizeof(CandResult) == sizeof(ffs_cand_result), "ABI struct mismatch");
    static_assert(sizeof(PairResult) == sizeof(ffs_pair_result), "ABI struct mismatch");
    HIP_TRY(hipSetDevice(p->device));
    int rc;
    if ((rc = enter_stream(p, st))) return rc;
    StreamLeave leave{p, st};
    auto now_ns = [] { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double ht0 = p->host_timing ? now_ns() : 0.0;
    double ht1 = ht0, ht2 = ht0, ht3 = ht0;

    const int n_packed = (n_cand + 1) / 2;
    const int n_slots = 1 + n_packed;  // length-N buffers per pair, in either layout
    const int xf_per_pair = n_slots;
    const int sel_ref = 0 | (1 << 16), sel_cand = 1 | ((n_slots - 1) << 16);  // row_sel of the two first-pass launches when mixed
    const bool ref_half = !p->direct_only && ref_half_ok(p);
    // odd candidate count: the last packed transform carries one real candidate -> half of its rows suffice
    // (needs the one-row-per-block mid kernels, like ref_half; the plan that runs the kernels decides)
    auto half_flags_of = [&](const ffs_plan* q, bool rh) {
        const bool hl = (n_cand % 2 == 1) && q->allow_half_last && ref_half_ok(q) && q->N1 >= 4;
        return (rh ? HALF_REF : 0) | (hl ? HALF_LAST : 0);
    };
    const int hflags = p->direct_only ? 0 : half_flags_of(p, ref_half);
    const int slot_map = n_slots;  // see slot_stride()/cand_slot() in ffs_kernels.h
    const size_t n_cands = (size_t)n_pairs * n_cand;
    size_t n_xf = (size_t)n_pairs * xf_per_pair;
    const size_t n_xf_alloc = p->seg ? (size_t)n_pairs * kSegBlocks * n_slots : n_xf;  // block-segmented mode needs more
    const int n_chunks = (n_pairs + p->pairs_in_flight - 1) / p->pairs_in_flight;
    // Run-boundary path: two-level vectors on both sides, as bits (FFS_DTYPE_U1: their lists are extracted here) or as
    // boundary lists (FFS_DTYPE_RUNS); everything else, and every sub-batch whose lists turn out too long, goes through
    // the transforms.  Its buffers come first: without them (out of memory) an AUTO call on bits still has the transforms.
    // Multi-level float references (the `weighted` fused VAD's four levels, speech_transformers.py:290-293) against
    // two-level candidates: the reference's threshold planes are made on the device (k_levels_*), their lists extracted,
    // and the run-boundary kernel adds the levels up (ffs_runs.h, LevelInfo); whatever does not qualify takes the transforms.
    const bool ml = p->algo != FFS_ALGO_FFT && !p->direct_only && (ref_dt == FFS_DTYPE_F64 || ref_dt == FFS_DTYPE_F32) && runs_able(dtype);
    bool runs_ok = p->algo != FFS_ALGO_FFT && !p->direct_only && runs_able(dtype) && (runs_able(ref_dt) || ml);
    const bool need_extract = runs_ok && (ml || dtype == FFS_DTYPE_U1 || ref_dt == FFS_DTYPE_U1);
    const size_t n_rr = n_vec + (ml ? 3 * (size_t)n_pairs : 0);  // RunsRef entries: every vector, then three threshold planes per pair
    if (runs_ok && (rc = ensure_runs(p, need_extract ? n_rr : 0, 0, (size_t)n_chunks))) {
        if (lists_in || p->algo != FFS_ALGO_AUTO) return rc;
        runs_ok = false;  // FFS_ALGO_AUTO: "only the time differs" -- the transforms need no list buffers
    }
    // the types the transform kernels see (list-only vectors are expanded to bits before they run)
    const int t_dtype = dtype == FFS_DTYPE_RUNS ? FFS_DTYPE_U1 : dtype, t_ref_dt = ref_dt == FFS_DTYPE_RUNS ? FFS_DTYPE_U1 : ref_dt;
    const bool mixed = t_ref_dt != t_dtype;
    // descriptor block layout: [PoolHeader][CandDesc n_cands][RunsRef n_vec][XformDesc n_xf][NomList n_cands][RescoreAcc n_cands*KNOM]
    const size_t o_pool = 0;  // PoolHeader (uploaded: count = 0, capacity)
    const size_t o_cand = 64;
    const size_t o_rv = o_cand + n_cands * sizeof(CandDesc);  // RunsRef[n_rr] when runs_ok
    // multi-level tables behind it: LevelInfo | sample pointers | plane pointers | lengths | plane words, one per pair
    const size_t o_li = (o_rv + (runs_ok ? n_rr * sizeof(RunsRef) : 0) + 63) & ~(size_t)63;
    const bool ml_on = runs_ok && ml;
    const size_t o_mp = o_li + (ml_on ? (size_t)n_pairs * sizeof(LevelInfo) : 0);
    const size_t o_mq = o_mp + (ml_on ? (size_t)n_pairs * 8 : 0);
    const size_t o_ml = o_mq + (ml_on ? (size_t)n_pairs * 8 : 0);
    const size_t o_mw = o_ml + (ml_on ? (size_t)n_pairs * 4 : 0);
    const size_t o_xf = (o_mw + (ml_on ? (size_t)n_pairs * 4 : 0) + 63) & ~(size_t)63;
    const size_t host_bytes_max = o_xf + (n_xf_alloc > n_xf ? n_xf_alloc : n_xf) * sizeof(XformDesc);
    const size_t o_nom = (host_bytes_max + 255) & ~(size_t)255;
    const size_t o_acc = o_nom + n_cands * sizeof(NomList);
    const size_t o_pbest = o_acc + n_cands * KNOM * sizeof(RescoreAcc);  // zeroed together with acc
    const size_t total = o_pbest + n_cands * sizeof(PoolBest);
    if ((rc = ensure_desc(p, total))) return rc;
    HIP_TRY(hipEventSynchronize(p->upload_done));  // previous call's upload has left the pinned buffer
    char* hb = (char*)p->host_desc;
    {
        PoolHeader* ph = (PoolHeader*)(hb + o_pool);
        memset(hb + o_pool, 0, 64);
        ph->capacity = kPoolCapacity;
    }
    CandDesc* hc = (CandDesc*)(hb + o_cand);
    XformDesc* hx = (XformDesc*)(hb + o_xf);
    RunsRef* hrv = (RunsRef*)(hb + o_rv);
    const bool ml_early = runs_ok && ml;
    const bool use_copy = runs_ok && need_extract && !ml_early && !(p->algo == FFS_ALGO_AUTO && p->runs_prev_fft && !lists_in) && ensure_copy(p);
    p->cur_half = use_copy ? 1 - p->cur_half : 0;
    char* db = (char*)(p->cur_half ? p->dev_desc2 : p->dev_desc);
    bool probe_first = false;           // (run-boundary path) a density probe stands in for the extraction so far
    // sub-batch flags + two 8-byte statistics behind them (boundaries of the call, its longest plan-owned list): cleared by the
    // extraction kernel's first workgroup when one is launched, by a memset otherwise
    const size_t flag_bytes = (((size_t)n_chunks * sizeof(int) + 7) & ~(size_t)7) + 16;
    bool flags_cleared = false;
    // Small calls: ONE upload (vector table + candidate descriptors) and the extraction behind it, instead of the table, the
    // extraction and then the candidates -- the host takes longer to build ~1000 descriptors than the device to extract
    // 1000 lists, so the early launch only moved a 20 us hole behind the extraction and cost one more stream operation
    // (profiles/small_step_timeline.py).  Large calls keep the early launch: the device reads the vectors while the host
    // builds tens of thousands of descriptors.
    bool late_extract = false;
    const void* const* xptr = vec_ptr;  // the vectors as the transform path reads them (list-only vectors: their expansion)
    std::vector<const void*> expanded;
    if (runs_ok) {
        // Run-boundary path, step 1: what the kernels need of every vector.  The lists of vectors that arrive as bits are
        // extracted first -- the launch needs only this table, so the candidate descriptors are built while the device
        // reads the vectors.
        for (size_t i = 0; i < n_vec; ++i) {
            const int dt = (i % stride) ? dtype : ref_dt;
            if (dt == FFS_DTYPE_RUNS)  // caller-owned block: 16-byte header (n, ones, len, capacity), then the entries
                hrv[i] = RunsRef{(const int2*)((const char*)vec_ptr[i] + 16), (const int2*)vec_ptr[i], nullptr, (int32_t)vec_len[i], 0};
            else if (dt == FFS_DTYPE_U1)
                hrv[i] = RunsRef{p->runs_e + i * (size_t)p->runs_stride, p->runs_n + i, (const unsigned*)vec_ptr[i], (int32_t)vec_len[i], p->runs_stride};
            else  // a multi-level reference: its threshold planes follow the vectors
                hrv[i] = RunsRef{nullptr, p->runs_n + i, nullptr, (int32_t)vec_len[i], p->runs_stride};
        }
        if (ml_on) {
            std::vector<size_t> poff((size_t)n_pairs + 1, 0);
            int64_t len_max = 1;
            for (int pi = 0; pi < n_pairs; ++pi) {
                const size_t pw = (size_t)(vec_len[(size_t)pi * stride] + 31) / 32;
                poff[pi + 1] = poff[pi] + ((3 * pw * 4 + 63) & ~(size_t)63);
                if (vec_len[(size_t)pi * stride] > len_max) len_max = vec_len[(size_t)pi * stride];
            }
            if (poff[n_pairs] > p->lvl_bytes) {
                if (p->has_last) HIP_TRY(hipEventSynchronize(p->last_done));
                (void)hipFree(p->lvl_buf);
                p->lvl_buf = nullptr;
                p->lvl_bytes = 0;
                HIP_TRY(hipMalloc((void**)&p->lvl_buf, poff[n_pairs] + poff[n_pairs] / 4));
                p->lvl_bytes = poff[n_pairs] + poff[n_pairs] / 4;
            }
            const void** h_vp = (const void**)(hb + o_mp);
            unsigned** h_pp = (unsigned**)(hb + o_mq);
            int32_t* h_len = (int32_t*)(hb + o_ml);
            int32_t* h_pw = (int32_t*)(hb + o_mw);
            for (int pi = 0; pi < n_pairs; ++pi) {
                const size_t b = (size_t)pi * stride;
                const int32_t pw = (int32_t)((vec_len[b] + 31) / 32);
                unsigned* planes = (unsigned*)((char*)p->lvl_buf + poff[pi]);
                h_vp[pi] = vec_ptr[b];
                h_pp[pi] = planes;
                h_len[pi] = (int32_t)vec_len[b];
                h_pw[pi] = pw;
                for (int k = 0; k < 3; ++k) {
                    const size_t r = n_vec + 3 * (size_t)pi + k;
                    hrv[r] = RunsRef{p->runs_e + r * (size_t)p->runs_stride, p->runs_n + r, planes + (size_t)k * pw, (int32_t)vec_len[b], p->runs_stride};
                }
            }
            HIP_TRY(hipMemcpyAsync(db + o_rv, hb + o_rv, o_xf - o_rv, hipMemcpyHostToDevice, st));
            leave.armed = true;
            LevelInfo* d_li = (LevelInfo*)(db + o_li);
            const unsigned chunks = (unsigned)((len_max + 16383) / 16384);
            {
            ProfSpan lspan(p, st, FFS_K_LEVELS);
            if (ref_dt == FFS_DTYPE_F64) {
                hipLaunchKernelGGL((k_levels_sample<double>), dim3((unsigned)n_pairs), dim3(256), 0, st, (const double* const*)(db + o_mp),
                                   (const int*)(db + o_ml), d_li);
                hipLaunchKernelGGL((k_levels_bits<double>), dim3(chunks, (unsigned)n_pairs), dim3(256), 0, st, (const double* const*)(db + o_mp),
                                   (const int*)(db + o_ml), d_li, (unsigned* const*)(db + o_mq), (const int*)(db + o_mw));
            } else {
                hipLaunchKernelGGL((k_levels_sample<float>), dim3((unsigned)n_pairs), dim3(256), 0, st, (const float* const*)(db + o_mp),
                                   (const int*)(db + o_ml), d_li);
                hipLaunchKernelGGL((k_levels_bits<float>), dim3(chunks, (unsigned)n_pairs), dim3(256), 0, st, (const float* const*)(db + o_mp),
                                   (const int*)(db + o_ml), d_li, (unsigned* const*)(db + o_mq), (const int*)(db + o_mw));
            }
            }
            HIP_TRY(hipGetLastError());
            {
                ProfSpan span(p, st, FFS_K_RUNS_EXTRACT);
                runs_extract_launch((const RunsRef*)(db + o_rv), n_rr, false, st, p->runs_flags, (int)(flag_bytes / sizeof(int)));  // (full lists: see runs_corr_body, p_sh)
                flags_cleared = true;
            }
            HIP_TRY(hipGetLastError());
        } else if (need_extract && n_cands <= kSmallCallCands && !(p->algo == FFS_ALGO_AUTO && p->runs_prev_fft && !lists_in)) {
            late_extract = true;  // (uploaded and launched behind the candidate descriptors, below)
        } else if (need_extract) {
            if (use_copy) {
                if (p->half_used[p->cur_half]) HIP_TRY(hipStreamWaitEvent(p->copy_stream, p->half_done[p->cur_half], 0));
                HIP_TRY(hipMemcpyAsync(db + o_rv, hb + o_rv, n_vec * sizeof(RunsRef), hipMemcpyHostToDevice, p->copy_stream));
                HIP_TRY(hipEventRecord(p->copy_ev, p->copy_stream));
                HIP_TRY(hipStreamWaitEvent(st, p->copy_ev, 0));
            } else {
                HIP_TRY(hipMemcpyAsync(db + o_rv, hb + o_rv, n_vec * sizeof(RunsRef), hipMemcpyHostToDevice, st));
            }
            leave.armed = true;
            // A stream of dense calls (the previous one needed the transforms): sample the vectors first -- when the
            // estimate puts every sub-batch far over budget the lists are never extracted (decided below, once the
            // candidate descriptors exist); otherwise the extraction is launched then.
            probe_first = p->algo == FFS_ALGO_AUTO && p->runs_prev_fft && !lists_in;
            if (probe_first) {
                hipLaunchKernelGGL(k_runs_probe, dim3((unsigned)((n_vec + 3) / 4)), dim3(256), 0, st, (const RunsRef*)(db + o_rv), (int)n_vec);
            } else {
                ProfSpan span(p, st, FFS_K_RUNS_EXTRACT);
                runs_extract_launch((const RunsRef*)(db + o_rv), n_vec, false, st, p->runs_flags, (int)(flag_bytes / sizeof(int)), (int)stride, (int)n_vec);
                flags_cleared = true;
            }
            HIP_TRY(hipGetLastError());
        }
    }
    if (p->host_timing) ht1 = now_ns();
    int tiles_max = 1;
    auto fill_all_cands = [&](const void* const* ptrs, int rdt, int cdt) -> int {
        for (int pi = 0; pi < n_pairs; ++pi) {
            const size_t b = (size_t)pi * stride;
            const VecView ref{ptrs[b], vec_len[b], vec_lo[b], vec_hi[b]};
            for (int j = 0; j < n_cand; ++j) {
                const VecView sub{ptrs[b + 1 + j], vec_len[b + 1 + j], vec_lo[b + 1 + j], vec_hi[b + 1 + j]};
                CandDesc& cd = hc[(size_t)pi * n_cand + j];
                int rcf;
                if ((rcf = fill_cand(p, ref, sub, max_offset_samples, &cd, rdt, cdt))) return rcf;
                if (runs_ok && !(cd.flags & CAND_NO_LAGS)) {
                    const int t = (cd.d_hi - cd.d_lo + RUNS_T) / RUNS_T;
                    if (t > tiles_max) tiles_max = t;
                }
            }
        }
        return FFS_OK;
    };
    if ((rc = fill_all_cands(vec_ptr, t_ref_dt, t_dtype))) return rc;
    // ---- descriptors of the transform path (built only when some sub-batch needs it) -------------------------
    BinList bins;
    memset(&bins, 0, sizeof bins);
    bool pruned = false, seg = false;
    int seg_blocks = 0;
    int64_t seg_lo = 0;
    auto build_xforms = [&]() -> int {
    for (size_t i = 0; i < n_cands; ++i)
        if ((rc = finish_cand_for_transforms(p, max_offset_samples, &hc[i]))) return rc;
    std::vector<VecView> views((size_t)n_pairs * stride);  // per pair: reference, candidates (reachable prefixes)
    for (int pi = 0; pi < n_pairs; ++pi) {
        const size_t b = (size_t)pi * stride;
        VecView ref{xptr[b], vec_len[b], vec_lo[b], vec_hi[b]};
        std::vector<VecView> subs(n_cand);
        int64_t ref_used = 1;
        for (int j = 0; j < n_cand; ++j) {
            subs[j] = VecView{xptr[b + 1 + j], vec_len[b + 1 + j], vec_lo[b + 1 + j], vec_hi[b + 1 + j]};
            const CandDesc& cd = hc[(size_t)pi * n_cand + j];
            // the transforms only read the prefixes that can reach the lag window (the exact re-evaluation
            // keeps working on the whole vectors through the candidate descriptor)
            int64_t s_eff, r_eff;
            effective_lengths(ref.len, subs[j].len, cd.d_lo, cd.d_hi, &s_eff, &r_eff);
            if (!(cd.flags & CAND_NO_LAGS)) {
                subs[j].len = s_eff;
                if (r_eff > ref_used) ref_used = r_eff;
            }
        }
        ref.len = ref_used;
        views[b] = ref;
        for (int j = 0; j < n_cand; ++j) views[b + 1 + j] = subs[j];
        fill_xform(&hx[(size_t)pi * n_slots], &ref, nullptr);
        for (int k = 0; k < n_packed; ++k)
            fill_xform(&hx[(size_t)pi * n_slots + 1 + k], &subs[2 * k], (2 * k + 1 < n_cand) ? &subs[2 * k + 1] : nullptr);
    }
    // last-pass bins reachable by any lag window of this call (pruned pass C when there are few)
    std::vector<int> bin_set;
    bool bins_overflow = p->direct_only;
    for (size_t i = 0; i < n_cands && !bins_overflow; ++i) add_bins(p, hc[i], &bin_set, &bins_overflow);
    bins.n = (int)bin_set.size();
    for (int i = 0; i < bins.n; ++i) bins.b[i] = bin_set[i];
    pruned = p->allow_pruned && !bins_overflow && bins.n > 0 && bins.n * 2 <= p->N1;
    // Block-segmented mode (see k_mid_seg): every candidate is cut into n_blocks <= 3 blocks of B samples,
    // block k is correlated with the reference stretch [kB + D_lo, kB + D_lo + M) by a length-M transform
    // (lag d at output index d - D_lo, nothing wraps), the blocks' spectrum products are added in the
    // mid pass.  [D_lo, D_hi] = union of the call's lag windows.
    if (p->seg && p->allow_seg && p->allow_pruned && !p->direct_only && max_offset_samples >= 0 && p->seg->N2 == 4096 &&
        ref_half_ok(p->seg)) {
        const ffs_plan* sp = p->seg;
        int64_t d_lo = INT64_MAX, d_hi = INT64_MIN, s_max = 1;
        for (size_t i = 0; i < n_cands; ++i) {
            if (hc[i].flags & CAND_NO_LAGS) continue;
            if (hc[i].d_lo < d_lo) d_lo = hc[i].d_lo;
            if (hc[i].d_hi > d_hi) d_hi = hc[i].d_hi;
            const int64_t sl = views[(i / n_cand) * stride + 1 + i % n_cand].len;
            if (sl > s_max) s_max = sl;
        }
        if (d_lo <= d_hi) {
            const int64_t W = d_hi - d_lo + 1, B = sp->N - (W - 1);
            const int64_t nb = (W - 1) / sp->N2 + 1;
            if (B >= 1 && nb <= MAXBINS && nb * 2 <= sp->N1 && (s_max + B - 1) / B <= kSegBlocks) {
                seg = true;
                seg_blocks = (int)((s_max + B - 1) / B);
                seg_lo = d_lo;
                // byte / float vectors move the pointer to the block's first sample, bit-packed ones the bit offset
                const size_t esz = esz_of(t_dtype), esz_r = esz_of(t_ref_dt);
                n_xf = (size_t)n_pairs * seg_blocks * n_slots;
                for (int pi = 0; pi < n_pairs; ++pi) {
                    const VecView& ref = views[(size_t)pi * stride];
                    for (int k = 0; k < seg_blocks; ++k) {
                        XformDesc* x = &hx[((size_t)pi * seg_blocks + k) * n_slots];
                        VecView rb = ref;  // positions [lead, len) of the block input = ref[start + n]
                        const int64_t start = (int64_t)k * B + d_lo;
                        rb.lead = start < 0 ? -start : 0;
                        rb.len = ref.len - start < sp->N ? ref.len - start : sp->N;
                        if (rb.len <= rb.lead) {
                            rb.len = rb.lead = 0;  // nothing of the reference in this stretch
                        } else {
                            rb.ptr = (const char*)ref.ptr + start * (int64_t)esz_r;  // may point in front of the vector (lead)
                            if (!esz_r) rb.off = ref.off + start;
                        }
                        fill_xform(x, &rb, nullptr, ref.ptr);
                        std::vector<VecView> sb(n_cand);
                        for (int j = 0; j < n_cand; ++j) {
                            sb[j] = views[(size_t)pi * stride + 1 + j];
                            const int64_t left = sb[j].len - (int64_t)k * B;
                            sb[j].len = left <= 0 ? 0 : (left < B ? left : B);
                            if (sb[j].len > 0) {
                                sb[j].ptr = (const char*)sb[j].ptr + (int64_t)k * B * (int64_t)esz;
                                if (!esz) sb[j].off += (int64_t)k * B;
                            }
                        }
                        for (int kk = 0; kk < n_packed; ++kk)
                            fill_xform(x + 1 + kk, &sb[2 * kk], (2 * kk + 1 < n_cand) ? &sb[2 * kk + 1] : nullptr, ref.ptr);
                    }
                }
                memset(&bins, 0, sizeof bins);
                bins.n = (int)nb;
                for (int i = 0; i < bins.n; ++i) bins.b[i] = i;
                pruned = true;
            }
        }
    }
    return FFS_OK;
    };
    if (!runs_ok && (rc = build_xforms())) return rc;
    // one upload when the transforms run anyway; with the run-boundary path [header, candidates] (+ the vector table
    // when no extraction was launched ahead of it)
    if (runs_ok && late_extract && use_copy) {  // the whole block on the copy stream: it passes the previous call's kernels
        if (p->half_used[p->cur_half]) HIP_TRY(hipStreamWaitEvent(p->copy_stream, p->half_done[p->cur_half], 0));
        HIP_TRY(hipMemcpyAsync(db, hb, o_rv + n_rr * sizeof(RunsRef), hipMemcpyHostToDevice, p->copy_stream));
        HIP_TRY(hipEventRecord(p->upload_done, p->copy_stream));
        HIP_TRY(hipStreamWaitEvent(st, p->upload_done, 0));
    } else if (runs_ok && (!need_extract || late_extract)) {
        HIP_TRY(hipMemcpyAsync(db, hb, o_rv + n_rr * sizeof(RunsRef), hipMemcpyHostToDevice, st));
    } else if (runs_ok && use_copy) {  // (beside the extraction of this call)
        HIP_TRY(hipMemcpyAsync(db, hb, o_rv, hipMemcpyHostToDevice, p->copy_stream));
        HIP_TRY(hipEventRecord(p->upload_done, p->copy_stream));
        HIP_TRY(hipStreamWaitEvent(st, p->upload_done, 0));
    } else {
        HIP_TRY(hipMemcpyAsync(db, hb, runs_ok ? o_rv : o_xf + n_xf * sizeof(XformDesc), hipMemcpyHostToDevice, st));
    }
    leave.armed = true;
    if (!(runs_ok && use_copy))
        HIP_TRY(hipEventRecord(p->upload_done, st));  // (recorded again should a fallback upload more: the next call waits for the latest)
    if (late_extract) {
        ProfSpan span(p, st, FFS_K_RUNS_EXTRACT);
        runs_extract_launch((const RunsRef*)(db + o_rv), n_vec, false, st, p->runs_flags, (int)(flag_bytes / sizeof(int)), (int)stride, (int)n_vec);
        flags_cleared = true;
        HIP_TRY(hipGetLastError());
    }
    const CandDesc* dc = (const CandDesc*)(db + o_cand);
    const XformDesc* dx = (const XformDesc*)(db + o_xf);
    NomList* dn = (NomList*)(db + o_nom);
    RescoreAcc* da = (RescoreAcc*)(db + o_acc);
    PoolBest* dpb = (PoolBest*)(db + o_pbest);
    PoolArgs pa{dn, (PoolHeader*)(db + o_pool), nullptr, dpb,  // .entries: once the transform workspace exists
                (n_pairs + p->pairs_in_flight - 1) / p->pairs_in_flight};
    pa.half_last = (hflags & HALF_LAST) ? 1 : 0;
    CandResult* cres = (CandResult*)cand_out_dev;
    PairResult* pres = (PairResult*)pair_out_dev;
    const int dt = t_dtype;  // (the transform kernels' element type of the candidates)

    std::vector<char> chunk_fft((size_t)n_chunks, runs_ok ? 0 : 1);
    bool any_fft = !runs_ok;
    if (p->direct_only) {
        FFS_BY_RESCORE_DTYPE(mixed, dt, hipLaunchKernelGGL((k_direct<DT>), dim3((unsigned)n_cands), dim3(256), 0, st, dc, cres));
        HIP_TRY(hipGetLastError());
    } else {
        // ---- run-boundary path: exact correlation of every candidate whose lists are short enough, candidate and pair
        // records written by the kernel itself; which sub-batches need the transforms instead is decided on the device
        // (k_runs_chunk_flags) and read back as one int per sub-batch -- after everything else of the call is queued
        if (runs_ok) {
            if ((rc = ensure_runs(p, need_extract ? n_rr : 0, tiles_max > 1 ? n_cands * (size_t)tiles_max : 0, (size_t)n_chunks)))
                return rc;
            const long long budget = p->algo == FFS_ALGO_RUNS ? INT64_MAX / 4
                                     : p->runs_budget >= 0 ? p->runs_budget
                                                           : kRunsBudgetPerPoint * (long long)p->N * (n_cand + 1) / (2 * n_cand);
            // Lists that arrive with host-known length bounds (the rasteriser's: two entries per subtitle) settle the
            // question on the host: within budget even at the bounds -> no flags kernel, no copy back, no wait.  A bound of
            // RUNS_CAP or more (16-bit histogram cells, as in k_runs_chunk_flags) proves nothing: the device decides.
            bool proven = !need_extract && vec_bound != nullptr;
            for (int pi = 0; proven && pi < n_pairs; ++pi) {
                const size_t bq = (size_t)pi * stride;
                if (vec_bound[bq] <= 0) proven = false;
                for (int j = 0; proven && j < n_cand; ++j) {
                    const CandDesc& cd = hc[(size_t)pi * n_cand + j];
                    if (vec_bound[bq + 1 + j] <= 0 ||
                        (!(cd.flags & CAND_NO_LAGS) &&
                         runs_over_budget(vec_bound[bq + 1 + j], vec_bound[bq], RUNS_CAP, RUNS_CAP, (long long)cd.d_hi - cd.d_lo + 1, cd.R, budget)))
                        proven = false;
                }
            }
            unsigned long long* d_stats = (unsigned long long*)((char*)p->runs_flags + flag_bytes - 16);  // [boundaries, longest list]
            const int* d_flags = proven ? p->runs_zero_flags : p->runs_flags;
            bool skip_runs = false;  // the probe says every sub-batch is dense: transforms only, nothing extracted
            if (probe_first) {
                // flags from the ESTIMATED counts against the budget (the estimate's sampling error is a few per cent of a
                // count whose square enters: a sub-batch within that of the break-even costs the same either way)
                HIP_TRY(hipMemsetAsync(p->runs_flags, 0, flag_bytes, st));
                hipLaunchKernelGGL(k_runs_chunk_flags, dim3((unsigned)n_chunks, runs_flag_split((long long)p->pairs_in_flight * n_cand)), dim3(256), 0, st, dc, n_pairs, n_cand,
                                   p->pairs_in_flight, (const RunsRef*)(db + o_rv), budget, p->runs_flags, d_stats, 1);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(p->runs_flags_host, p->runs_flags, flag_bytes, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipEventRecord(p->runs_ev, st));
                if ((rc = build_xforms())) return rc;  // (while the probe runs: needed in either case of a dense stream)
                HIP_TRY(hipEventSynchronize(p->runs_ev));
                skip_runs = true;
                for (int ch = 0; ch < n_chunks; ++ch) skip_runs = skip_runs && p->runs_flags_host[ch] == 1;
                if (!skip_runs) {  // not (any more) a dense stream: the usual order from here on
                    ProfSpan span(p, st, FFS_K_RUNS_EXTRACT);
                    runs_extract_launch((const RunsRef*)(db + o_rv), n_vec, false, st, nullptr, 0, (int)stride, (int)n_vec);
                }
                HIP_TRY(hipGetLastError());
            }
            if (skip_runs) {
                for (int ch = 0; ch < n_chunks; ++ch) chunk_fft[ch] = 1;
                any_fft = true;
                p->runs_calls += 1;
                p->runs_chunks += n_chunks;
                p->runs_fft_chunks += n_chunks;
                p->runs_last_boundaries = -1;
                HIP_TRY(hipMemcpyAsync(db + o_cand, hb + o_cand, o_xf + n_xf * sizeof(XformDesc) - o_cand, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemsetAsync(da, 0, n_cands * KNOM * sizeof(RescoreAcc) + n_cands * sizeof(PoolBest), st));
                HIP_TRY(hipEventRecord(p->upload_done, st));
            } else {
            if (!proven) {
                // (flags + the statistics behind them: cleared by the extraction kernel of this call unless a probe has
                // written estimates into them since, or no extraction ran)
                if (!flags_cleared || probe_first) HIP_TRY(hipMemsetAsync(p->runs_flags, 0, flag_bytes, st));
                if (ml_on)
                    hipLaunchKernelGGL(k_runs_chunk_flags_ml, dim3((unsigned)n_chunks), dim3(256), 0, st, dc, n_pairs, n_cand,
                                       p->pairs_in_flight, (const RunsRef*)(db + o_rv), (const LevelInfo*)(db + o_li), (int)n_vec, budget,
                                       p->runs_flags, d_stats);
                else
                hipLaunchKernelGGL(k_runs_chunk_flags, dim3((unsigned)n_chunks, runs_flag_split((long long)p->pairs_in_flight * n_cand)), dim3(256), 0, st, dc, n_pairs, n_cand,
                                   p->pairs_in_flight, (const RunsRef*)(db + o_rv), budget, p->runs_flags, d_stats, 0);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(p->runs_flags_host, p->runs_flags, flag_bytes, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipEventRecord(p->runs_ev, st));
            }
            {
                ProfSpan span(p, st, FFS_K_RUNS_CORR);
                if (ml_on)
                    hipLaunchKernelGGL(k_runs_corr_ml, dim3((unsigned)n_cands, (unsigned)tiles_max), dim3(RUNS_THREADS), 0, st, dc, n_cand,
                                       (const RunsRef*)(db + o_rv), cres, p->runs_best, tiles_max, d_flags, p->pairs_in_flight,
                                       (const LevelInfo*)(db + o_li), (int)n_vec);
                else {
                    // One workgroup per candidate.  The kernel can also walk several candidates of a pair in one workgroup
                    // (FFS_RUNS_SPLIT = workgroups per pair; the reference's list is then staged once for them), measured in
                    // round 6: 7 x fewer stagings, but the 7 x longer workgroups leave the chip idle longer at the end of a
                    // launch -- 0.248 against 0.238 us per pair at 4096 pairs (profiles/r06_runs_experiments.json).
                    int split = n_cand;
                    if (p->runs_split > 0) split = p->runs_split < n_cand ? p->runs_split : n_cand;
                    hipLaunchKernelGGL(k_runs_corr, dim3((unsigned)((size_t)n_pairs * split), (unsigned)tiles_max), dim3(RUNS_THREADS), 0, st,
                                       dc, n_cand, (const RunsRef*)(db + o_rv), cres, p->runs_best, tiles_max, d_flags, p->pairs_in_flight,
                                       split);
                }
                if (tiles_max > 1)
                    hipLaunchKernelGGL(k_runs_pick, dim3((unsigned)((n_cands + 255) / 256)), dim3(256), 0, st, dc, (int)n_cands, n_cand,
                                       p->runs_best, tiles_max, cres, d_flags, p->pairs_in_flight);
            }
            // (pairs of sub-batches that go through the transforms are finished again behind those)
            hipLaunchKernelGGL(k_finalize_pairs, dim3((n_pairs + 255) / 256), dim3(256), 0, st, cres, pres, n_pairs, n_cand,
                               (long long)filter_max_offset);
            HIP_TRY(hipGetLastError());
            if (p->host_timing) ht2 = now_ns();
            p->runs_calls += 1;
            p->runs_chunks += n_chunks;
            p->runs_last_boundaries = -1;  // (not known on the host when nothing was copied back)
            if (!proven) {
                // a stream of dense calls: the transform descriptors are built while the device extracts and decides
                bool built = probe_first;  // (a probe has built them already)
                if (!built && p->runs_prev_fft && !lists_in) {
                    if ((rc = build_xforms())) return rc;
                    built = true;
                }
                HIP_TRY(hipEventSynchronize(p->runs_ev));  // k_runs_corr keeps the device busy meanwhile
                if (p->host_timing) ht3 = now_ns();
                bool bad = false;
                for (int ch = 0; ch < n_chunks; ++ch) {
                    const int f = p->runs_flags_host[ch];
                    if (f == 2) bad = true;
                    if (f) chunk_fft[ch] = 1, any_fft = true;
                }
                memcpy(&p->runs_last_boundaries, (char*)p->runs_flags_host + flag_bytes - 16, 8);
                {
                    // A plan-owned list filled its slot: the call is solved again with four times the room (at most twice:
                    // 4096 -> 16 384 -> 32 768 entries; what the first attempt wrote is overwritten in stream order), and the
                    // plan keeps the longer stride.  Which path solves what therefore does not depend on the stride.
                    long long longest = 0;
                    memcpy(&longest, (char*)p->runs_flags_host + flag_bytes - 8, 8);
                    if (need_extract && longest >= p->runs_stride && p->runs_stride < RUNS_CAP) {
                        p->runs_stride = p->runs_stride * 4 < RUNS_CAP ? p->runs_stride * 4 : RUNS_CAP;
                        p->runs_calls -= 1;
                        p->runs_chunks -= n_chunks;
                        leave.armed = false;
                        if ((rc = leave_stream(p, st))) return rc;
                        return align_impl(p, n_pairs, n_cand, ref_dt, dtype, vec_ptr, vec_len, vec_lo, vec_hi, vec_bound, max_offset_samples,
                                          filter_max_offset, cand_out_dev, pair_out_dev, hip_stream);
                    }
                }
                if (bad)
                    return fail(FFS_E_INVALID, "a boundary list of the call is truncated (more entries than its block holds): "
                                "nothing can solve it -- pass the vector as bits");
                p->runs_prev_fft = any_fft;
                if (any_fft) {
                    for (int ch = 0; ch < n_chunks; ++ch) p->runs_fft_chunks += chunk_fft[ch];
                    if (lists_in) {  // the transforms read bits: expand the list-only vectors, point the descriptors at them
                        if ((rc = expand_lists(&expanded))) return rc;
                        xptr = expanded.data();
                        int tm = tiles_max;
                        if ((rc = fill_all_cands(xptr, t_ref_dt, t_dtype))) return rc;
                        tiles_max = tm;
                    }
                    if (!built && (rc = build_xforms())) return rc;  // also completes the candidate descriptors (tie margins)
                    HIP_TRY(hipMemcpyAsync(db + o_cand, hb + o_cand, o_xf + n_xf * sizeof(XformDesc) - o_cand, hipMemcpyHostToDevice, st));
                    for (int ch = 0; ch < n_chunks; ++ch) {  // clean accumulators for the sub-batches that are solved again
                        if (!chunk_fft[ch]) continue;
                        const size_t c0 = (size_t)ch * p->pairs_in_flight * n_cand;
                        const size_t c1 = c0 + (size_t)p->pairs_in_flight * n_cand < n_cands ? c0 + (size_t)p->pairs_in_flight * n_cand : n_cands;
                        HIP_TRY(hipMemsetAsync(da + c0 * KNOM, 0, (c1 - c0) * KNOM * sizeof(RescoreAcc), st));
                        HIP_TRY(hipMemsetAsync(dpb + c0, 0, (c1 - c0) * sizeof(PoolBest), st));
                    }
                    HIP_TRY(hipEventRecord(p->upload_done, st));
                }
            }
            }  // !skip_runs
        } else {
            HIP_TRY(hipMemsetAsync(da, 0, n_cands * KNOM * sizeof(RescoreAcc) + n_cands * sizeof(PoolBest), st));
        }
        if (any_fft) {
            if ((rc = ensure_workspace(p))) return rc;
            pa.entries = p->pool_entries;
        }
        const int tiles = p->N2 / p->C;
        const bool full3 = !pruned && col3r_ok(p);  // full last pass over radix-3 columns: k_pass_c3's (wider) tiles
        for (int p0 = 0; seg && p0 < n_pairs; p0 += p->pairs_in_flight) {
            if (!chunk_fft[p0 / p->pairs_in_flight]) continue;
            // block-segmented pipeline on the length-M sub-plan: 3x shorter transforms, the blocks'
            // spectrum products are added in the mid pass, last pass over one third of the data
            ffs_plan* sp = p->seg;
            const int sflags = half_flags_of(sp, true);
            PoolArgs spa = pa;
            spa.half_last = (sflags & HALF_LAST) ? 1 : 0;
            const int np = (n_pairs - p0) < p->pairs_in_flight ? (n_pairs - p0) : p->pairs_in_flight;
            const int first_cand = p0 * n_cand;
            const int tiles_s = sp->N2 / sp->C;
            const XformDesc* dxs = dx + (size_t)p0 * seg_blocks * n_slots;
            {
                ProfSpan span(p, st, FFS_K_PASS_A);
                if (mixed) {  // the reference slot of every group in its own type, then the candidate slots in theirs
                    FFS_BY_DTYPE(t_ref_dt, rc = launch_pass_a<DT>(sp, dxs, np * seg_blocks * n_slots, n_slots, n_slots, sflags, st, sel_ref));
                    if (!rc) FFS_BY_DTYPE(t_dtype, rc = launch_pass_a<DT>(sp, dxs, np * seg_blocks * n_slots, n_slots, n_slots, sflags, st, sel_cand));
                } else {
                    FFS_BY_DTYPE(t_dtype, rc = launch_pass_a<DT>(sp, dxs, np * seg_blocks * n_slots, n_slots, n_slots, sflags, st));
                }
            }
            if (rc) return rc;
            {
                ProfSpan span(p, st, FFS_K_MID);
                rc = launch_mid_seg(sp, np, n_slots, seg_blocks, sflags, st);
            }
            if (rc) return rc;
            {
                ProfSpan span(p, st, FFS_K_PASS_C);
                rc = launch_pass_c_pruned<false>(sp, dc, first_cand, n_cand, n_packed, seg_blocks * n_slots, np, bins, spa, st, 1,
                                                 (int)seg_lo);
            }
            if (rc) return rc;
            HIP_TRY(hipMemsetAsync(sp->xlist, 0, sizeof(int), st));
            {
                ProfSpan span(p, st, FFS_K_NOMINEES);
                hipLaunchKernelGGL(k_nominees, dim3(np * n_cand), dim3(64), 0, st, sp->bnom, tiles_s, n_cand, n_packed, dc, dn,
                                   first_cand, sp->xlist);
            }
            HIP_TRY(hipGetLastError());
            if ((rc = launch_pass_c_pruned<true>(sp, dc, first_cand, n_cand, n_packed, seg_blocks * n_slots, np, bins, spa, st, 1,
                                                 (int)seg_lo)))
                return rc;
            {
                ProfSpan span(p, st, FFS_K_RESCORE);
                FFS_BY_RESCORE_DTYPE(mixed, t_dtype, hipLaunchKernelGGL((k_rescore<DT>), dim3(DT == 2 ? 4 : RSEG, np * n_cand), dim3(256), 0,
                                                                      st, dc, dn, da, first_cand));
            }
            HIP_TRY(hipGetLastError());
        }
        for (int p0 = 0; !seg && any_fft && p0 < n_pairs; p0 += p->pairs_in_flight) {
            if (!chunk_fft[p0 / p->pairs_in_flight]) continue;
            const int np = (n_pairs - p0) < p->pairs_in_flight ? (n_pairs - p0) : p->pairs_in_flight;
            const int first_cand = p0 * n_cand;
            {
                ProfSpan sp(p, st, FFS_K_PASS_A);
                if (mixed) {
                    FFS_BY_DTYPE(t_ref_dt, rc = launch_pass_a<DT>(p, dx + (size_t)p0 * xf_per_pair, np * xf_per_pair, xf_per_pair,
                                                                n_slots, hflags, st, sel_ref));
                    if (!rc) FFS_BY_DTYPE(t_dtype, rc = launch_pass_a<DT>(p, dx + (size_t)p0 * xf_per_pair, np * xf_per_pair,
                                                                        xf_per_pair, n_slots, hflags, st, sel_cand));
                } else {
                    FFS_BY_DTYPE(t_dtype, rc = launch_pass_a<DT>(p, dx + (size_t)p0 * xf_per_pair, np * xf_per_pair, xf_per_pair,
                                                               n_slots, hflags, st));
                }
            }
            if (rc) return rc;
            {
                ProfSpan sp(p, st, FFS_K_MID);
                rc = launch_mid(p, np, n_slots, hflags, st);
            }
            if (rc) return rc;
            {
                ProfSpan sp(p, st, FFS_K_PASS_C);
                rc = pruned ? launch_pass_c_pruned<false>(p, dc, first_cand, n_cand, n_packed, slot_map, np, bins, pa, st)
                     : full3 ? launch_pass_c3(p, dc, first_cand, n_cand, n_packed, slot_map, np, pa.half_last, st)
                             : launch_pass_c<0>(p, dc, first_cand, n_cand, n_packed, slot_map, np, nullptr, nullptr, pa, st);
            }
            if (rc) return rc;
            HIP_TRY(hipMemsetAsync(p->xlist, 0, sizeof(int), st));
            {
                ProfSpan sp(p, st, FFS_K_NOMINEES);
                hipLaunchKernelGGL(k_nominees, dim3(np * n_cand), dim3(64), 0, st, p->bnom, full3 ? p->N2 / col3r_cols(p) : tiles,
                                   n_cand, n_packed, dc, dn, first_cand, p->xlist);
            }
            HIP_TRY(hipGetLastError());
            // candidates whose nominee lists overflowed (listed by k_nominees): sweep their transforms again,
            // exhaustively; with an empty list the few blocks of this launch exit at once
            rc = pruned ? launch_pass_c_pruned<true>(p, dc, first_cand, n_cand, n_packed, slot_map, np, bins, pa, st)
                        : launch_pass_c<2>(p, dc, first_cand, n_cand, n_packed, slot_map, np, nullptr, nullptr, pa, st);
            if (rc) return rc;
            {
                ProfSpan sp(p, st, FFS_K_RESCORE);
                FFS_BY_RESCORE_DTYPE(mixed, t_dtype, hipLaunchKernelGGL((k_rescore<DT>), dim3(DT == 2 ? 4 : RSEG, np * n_cand), dim3(256), 0,
                                                                      st, dc, dn, da, first_cand));
            }
            HIP_TRY(hipGetLastError());
        }
        if (any_fft) {
            FFS_BY_RESCORE_DTYPE(mixed, t_dtype, hipLaunchKernelGGL((k_pool_rescore<DT>), dim3(2048), dim3(256), 0, st, dc, pa.header, pa.entries, dpb));
            hipLaunchKernelGGL(k_pool_pick, dim3(256), dim3(256), 0, st, pa.header, pa.entries, dpb);
        }
        // candidate and pair records of everything the transforms solved (the run-boundary kernels wrote their own)
        auto finalize_range = [&](size_t c0, size_t c1, int p0, int p1) {
            hipLaunchKernelGGL(k_finalize_cands, dim3((unsigned)((c1 - c0 + 255) / 256)), dim3(256), 0, st, dc + c0, dn + c0,
                               da + c0 * KNOM, cres + c0, (int)(c1 - c0), mixed ? 4 : t_dtype, pa.header, dpb + c0);
            hipLaunchKernelGGL(k_finalize_pairs, dim3((unsigned)((p1 - p0 + 255) / 256)), dim3(256), 0, st, cres + (size_t)p0 * n_cand,
                               pres + p0, p1 - p0, n_cand, (long long)filter_max_offset);
        };
        if (!runs_ok) {
            finalize_range(0, n_cands, 0, n_pairs);
        } else {
            for (int ch = 0; ch < n_chunks; ++ch) {
                if (!chunk_fft[ch]) continue;
                const int p0 = ch * p->pairs_in_flight, p1 = (p0 + p->pairs_in_flight) < n_pairs ? (p0 + p->pairs_in_flight) : n_pairs;
                finalize_range((size_t)p0 * n_cand, (size_t)p1 * n_cand, p0, p1);
            }
        }
        HIP_TRY(hipGetLastError());
    }
    if (p->direct_only) {
        hipLaunchKernelGGL(k_finalize_pairs, dim3((n_pairs + 255) / 256), dim3(256), 0, st, cres, pres, n_pairs, n_cand,
                           (long long)filter_max_offset);
        HIP_TRY(hipGetLastError());
    }
    leave.armed = false;
    rc = leave_stream(p, st);
    if (p->host_timing && runs_ok) {
        const double ht4 = now_ns();
        p->ht_vec += ht1 - ht0, p->ht_cand += ht2 - ht1, p->ht_wait += ht3 - ht2, p->ht_decide += ht4 - ht3, p->ht_total += ht4 - ht0;
    }
    return rc;
}

int ffs_align_batch(ffs_plan* p, int n_pairs, int n_cand, int dtype, const void* const* vec_ptr,
                    const int64_t* vec_len, const double* vec_lo, const double* vec_hi, int64_t max_offset_samples,
                    int64_t filter_max_offset, ffs_cand_result* cand_out_dev, ffs_pair_result* pair_out_dev,
                    void* hip_stream) {
    return align_impl(p, n_pairs, n_cand, dtype, dtype, vec_ptr, vec_len, vec_lo, vec_hi, nullptr, max_offset_samples,
                      filter_max_offset, cand_out_dev, pair_out_dev, hip_stream);
}

int ffs_align_batch_runs(ffs_plan* p, int n_pairs, int n_cand, const int32_t* vec_dtype, const void* const* vec_ptr,
                         const int64_t* vec_len, const double* vec_lo, const double* vec_hi, const int32_t* vec_max_boundaries,
                         int64_t max_offset_samples, int64_t filter_max_offset, ffs_cand_result* cand_out_dev,
                         ffs_pair_result* pair_out_dev, void* hip_stream) {
    if (!vec_dtype) return fail(FFS_E_INVALID, "null argument");
    if (n_pairs <= 0 || n_cand < 1) return n_pairs == 0 ? FFS_OK : fail(FFS_E_INVALID, "bad n_pairs / n_cand");
    // one type per ROLE: every reference of the call shares vec_dtype[0], every candidate vec_dtype[1]
    const int ref_dt = vec_dtype[0], cand_dt = vec_dtype[1];
    const size_t stride = 1 + (size_t)n_cand;
    for (size_t i = 0; i < (size_t)n_pairs * stride; ++i)
        if (vec_dtype[i] != (i % stride == 0 ? ref_dt : cand_dt))
            return fail(FFS_E_INVALID, "vector %zu has element type %d: within one call all references must share one type "
                        "and all candidates one type (split the call)", i, (int)vec_dtype[i]);
    return align_impl(p, n_pairs, n_cand, ref_dt, cand_dt, vec_ptr, vec_len, vec_lo, vec_hi, vec_max_boundaries, max_offset_samples,
                      filter_max_offset, cand_out_dev, pair_out_dev, hip_stream);
}

int ffs_align_batch_typed(ffs_plan* p, int n_pairs, int n_cand, const int32_t* vec_dtype, const void* const* vec_ptr,
                          const int64_t* vec_len, const double* vec_lo, const double* vec_hi, int64_t max_offset_samples,
                          int64_t filter_max_offset, ffs_cand_result* cand_out_dev, ffs_pair_result* pair_out_dev,
                          void* hip_stream) {
    return ffs_align_batch_runs(p, n_pairs, n_cand, vec_dtype, vec_ptr, vec_len, vec_lo, vec_hi, nullptr, max_offset_samples,
                                filter_max_offset, cand_out_dev, pair_out_dev, hip_stream);
}

int64_t ffs_runs_list_bytes(int64_t cap) { return cap < 0 ? 0 : 16 + 8 * cap; }

int ffs_runs_from_bits(const uint32_t* bits_dev, int64_t len, void* list_dev, int64_t cap, void* hip_stream) {
    if (!bits_dev || !list_dev) return fail(FFS_E_INVALID, "null argument");
    if (len <= 0 || len >= (int64_t(1) << 30)) return fail(len <= 0 ? FFS_E_EMPTY : FFS_E_TOO_LONG, "vector of %lld samples", (long long)len);
    if (cap < 1 || cap >= (int64_t(1) << 28)) return fail(FFS_E_INVALID, "list capacity %lld outside [1, 2^28)", (long long)cap);
    if (((uintptr_t)bits_dev & 3) || ((uintptr_t)list_dev & 7)) return fail(FFS_E_INVALID, "misaligned pointer");
    hipStream_t st = (hipStream_t)hip_stream;
    DeviceGuard guard;
    int rc_dev;
    if ((rc_dev = guard.enter(list_dev))) return rc_dev;
    hipLaunchKernelGGL(k_runs_extract_one, dim3(1), dim3(1024), 0, st, (const unsigned*)bits_dev, (int)len,
                       (int2*)((char*)list_dev + 16), (int2*)list_dev, (int)cap);
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}

int ffs_runs_to_bits(const void* list_dev, int64_t len, uint32_t* bits_out_dev, void* hip_stream) {
    if (!list_dev || !bits_out_dev) return fail(FFS_E_INVALID, "null argument");
    if (len <= 0 || len >= (int64_t(1) << 30)) return fail(len <= 0 ? FFS_E_EMPTY : FFS_E_TOO_LONG, "vector of %lld samples", (long long)len);
    if (((uintptr_t)bits_out_dev & 3) || ((uintptr_t)list_dev & 7)) return fail(FFS_E_INVALID, "misaligned pointer");
    hipStream_t st = (hipStream_t)hip_stream;
    DeviceGuard guard;
    int rc_dev;
    if ((rc_dev = guard.enter(list_dev))) return rc_dev;
    ExpandVec ev{(const int2*)((const char*)list_dev + 16), (const int2*)list_dev, (unsigned*)bits_out_dev, (int32_t)len, 0};
    ExpandVec* d_ev = nullptr;
    HIP_TRY(hipMallocAsync((void**)&d_ev, sizeof ev, st));
    {
        hipError_t hu = hipMemcpyAsync(d_ev, &ev, sizeof ev, hipMemcpyHostToDevice, st);
        if (hu == hipSuccess) hu = hipStreamSynchronize(st);  // (ev is a stack temporary)
        if (hu != hipSuccess) {
            (void)hipFreeAsync(d_ev, st);
            HIP_TRY(hu);
        }
    }
    const int chunks = (int)(((len + 31) / 32 + 255) / 256);
    // (asynchronous: a block whose header does not fit it is never followed past its capacity -- the bits are then garbage)
    hipLaunchKernelGGL(k_runs_expand, dim3((unsigned)chunks), dim3(256), 0, st, (const ExpandVec*)d_ev, chunks, (int*)nullptr);
    const hipError_t he = hipGetLastError();
    const hipError_t hf = hipFreeAsync(d_ev, st);
    HIP_TRY(he);
    HIP_TRY(hf);
    return FFS_OK;
}

int ffs_correlate_full(ffs_plan* p, int dtype, const void* ref_dev, int64_t ref_len, double ref_lo, double ref_hi,
                       const void* a_dev, int64_t a_len, double a_lo, double a_hi, const void* b_dev, int64_t b_len,
                       double b_lo, double b_hi, float* out_a_dev, float* out_b_dev, void* hip_stream) {
    if (!p || p->direct_only) return fail(FFS_E_INVALID, "plan has no FFT path (n_fft < %lld)", (long long)kMinFftN);
    if (dtype != FFS_DTYPE_U8 && dtype != FFS_DTYPE_F32 && dtype != FFS_DTYPE_U1 && dtype != FFS_DTYPE_F64)
        return fail(FFS_E_INVALID, "unknown dtype %d", dtype);
    if (!ref_dev || !a_dev || ref_len <= 0 || a_len <= 0) return fail(FFS_E_EMPTY, "empty reference or candidate");
    if (ref_len > p->N || a_len > p->N || (b_dev && b_len > p->N)) return fail(FFS_E_TOO_LONG, "vector longer than n_fft");
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(p->device));
    int rc;
    if ((rc = enter_stream(p, st))) return rc;
    if ((rc = ensure_workspace(p))) return rc;
    if ((rc = ensure_desc(p, 4096))) return rc;
    HIP_TRY(hipEventSynchronize(p->upload_done));
    p->cur_half = 0;
    XformDesc* hx = (XformDesc*)p->host_desc;
    VecView r{ref_dev, ref_len, ref_lo, ref_hi}, a{a_dev, a_len, a_lo, a_hi}, b{b_dev, b_len, b_lo, b_hi};
    fill_xform(&hx[0], &r, nullptr);
    fill_xform(&hx[1], &a, b_dev ? &b : nullptr);
    HIP_TRY(hipMemcpyAsync(p->dev_desc, hx, 2 * sizeof(XformDesc), hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(p->upload_done, st));
    const XformDesc* dx = (const XformDesc*)p->dev_desc;
    FFS_BY_DTYPE(dtype, rc = launch_pass_a<DT>(p, dx, 2, 2, 2, ref_half_ok(p) ? HALF_REF : 0, st));
    if (rc) return rc;
    if ((rc = launch_mid(p, 1, 2, ref_half_ok(p) ? HALF_REF : 0, st))) return rc;
    const PoolArgs none{nullptr, nullptr, nullptr, nullptr, 1};
    if ((rc = launch_pass_c<1>(p, nullptr, 0, 2, 1, 2, 1, out_a_dev, out_b_dev, none, st))) return rc;
    return leave_stream(p, st);
}

int64_t ffs_raster_length(const int64_t* end_us, int64_t n_subs, double ratio, double sample_rate) {
    double max_time = 0.0;  // speech_transformers.py:958-960
    for (int64_t i = 0; i < n_subs; ++i) {
        const double e = scaled_seconds(end_us[i], ratio);
        if (e > max_time) max_time = e;
    }
    return (int64_t)(max_time * sample_rate) + 2;       // :962
}

int ffs_raster_lengths(const int64_t* track_end_us_max, const double* ratio, int64_t n_vec, double sample_rate, int64_t* len_out) {
    if (n_vec < 0 || (n_vec > 0 && (!track_end_us_max || !ratio || !len_out))) return fail(FFS_E_INVALID, "bad argument");
    // the scaling is monotone (division, multiplication and the microsecond rounding all are), so a track's largest
    // scaled end is the scaled largest end
    for (int64_t v = 0; v < n_vec; ++v) len_out[v] = ffs_raster_length(&track_end_us_max[v], 1, ratio[v], sample_rate);
    return FFS_OK;
}

int64_t ffs_raster_intervals(const int64_t* start_us, const int64_t* end_us, const uint8_t* is_metadata, int64_t n_subs,
                             double ratio, double sample_rate, double start_seconds, int64_t out_len, int32_t* iv_out) {
    int64_t n = 0;
    for (int64_t i = 0; i < n_subs; ++i) {
        if (is_metadata && is_metadata[i]) continue;    // speech_transformers.py:966-967
        long long a, b;
        if (raster_interval(start_us[i], end_us[i], ratio, sample_rate, start_seconds, out_len, &a, &b)) {
            iv_out[2 * n] = (int32_t)a;
            iv_out[2 * n + 1] = (int32_t)b;
            ++n;
        }
    }
    return n;
}

static int rasterize_impl(const int64_t* start_us, const int64_t* end_us, const uint8_t* is_metadata, int64_t n_subs,
                          double ratio, double sample_rate, double start_seconds, void* out_dev, int64_t out_len,
                          bool bits, void* hip_stream) {
    if (n_subs < 0 || out_len < 0 || (n_subs > 0 && (!start_us || !end_us))) return fail(FFS_E_INVALID, "bad argument");
    if (out_len > 0 && !out_dev) return fail(FFS_E_INVALID, "null output");
    if (out_len >= (int64_t(1) << 31)) return fail(FFS_E_TOO_LONG, "raster longer than 2^31 samples");
    if (bits && ((uintptr_t)out_dev & 3)) return fail(FFS_E_INVALID, "bit-packed output must be 4-byte aligned");
    hipStream_t st = (hipStream_t)hip_stream;
    DeviceGuard guard;
    int rc_dev;
    if ((rc_dev = guard.enter(out_dev))) return rc_dev;
    if (out_len > 0) HIP_TRY(hipMemsetAsync(out_dev, 0, bits ? (size_t)((out_len + 31) / 32) * 4 : (size_t)out_len, st));
    // The interval list is a host temporary that an asynchronous copy reads later: park it in a
    // per-thread keep-alive list that is only emptied after a stream synchronisation, so that a run
    // of calls (seven ratios per file) costs one sync per 64 calls instead of one each.
    thread_local std::vector<std::vector<int32_t>> keep_alive;
    thread_local hipStream_t keep_stream = nullptr;
    if (keep_alive.size() >= 64 || (!keep_alive.empty() && keep_stream != st)) {
        HIP_TRY(hipStreamSynchronize(keep_stream));
        keep_alive.clear();
    }
    keep_stream = st;
    keep_alive.emplace_back((size_t)(2 * n_subs + 2));
    std::vector<int32_t>& iv = keep_alive.back();
    const int64_t n_iv = ffs_raster_intervals(start_us, end_us, is_metadata, n_subs, ratio, sample_rate, start_seconds,
                                              out_len, iv.data());
    if (n_iv == 0) {
        keep_alive.pop_back();
        return FFS_OK;
    }
    int2* d_iv = nullptr;
    HIP_TRY(hipMallocAsync((void**)&d_iv, (size_t)n_iv * sizeof(int2), st));
    HIP_TRY(hipMemcpyAsync(d_iv, iv.data(), (size_t)n_iv * sizeof(int2), hipMemcpyHostToDevice, st));
    const int blocks = (int)((n_iv + 3) / 4);
    if (bits)
        hipLaunchKernelGGL(k_fill_intervals_bits, dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, st, d_iv, (int)n_iv,
                           (unsigned*)out_dev);
    else
        hipLaunchKernelGGL(k_fill_intervals, dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, st, d_iv, (int)n_iv,
                           (unsigned char*)out_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipFreeAsync(d_iv, st));
    return FFS_OK;
}

int ffs_rasterize_subtitles(const int64_t* start_us, const int64_t* end_us, const uint8_t* is_metadata, int64_t n_subs,
                            double ratio, double sample_rate, double start_seconds, uint8_t* out_dev, int64_t out_len,
                            void* hip_stream) {
    return rasterize_impl(start_us, end_us, is_metadata, n_subs, ratio, sample_rate, start_seconds, out_dev, out_len, false,
                          hip_stream);
}

int ffs_rasterize_subtitles_bits(const int64_t* start_us, const int64_t* end_us, const uint8_t* is_metadata, int64_t n_subs,
                                 double ratio, double sample_rate, double start_seconds, uint32_t* out_dev,
                                 int64_t out_len, void* hip_stream) {
    return rasterize_impl(start_us, end_us, is_metadata, n_subs, ratio, sample_rate, start_seconds, out_dev, out_len, true,
                          hip_stream);
}

namespace {
// Pinned host staging for tables that an asynchronous copy reads after the entry point has returned.
struct PinnedStage {
    void* host = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    int device = -1;
    bool pending = false;
};
int stage_acquire(size_t bytes, PinnedStage** out) {
    thread_local PinnedStage stages[2];
    thread_local int next = 0;
    PinnedStage& sg = stages[next];
    next ^= 1;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (sg.pending) {
        (void)hipEventSynchronize(sg.done);  // the copy that read this buffer (two calls ago) has completed
        sg.pending = false;
    }
    if (sg.done && sg.device != dev) {  // events belong to a device
        (void)hipEventDestroy(sg.done);
        sg.done = nullptr;
    }
    if (!sg.done) {
        HIP_TRY(hipEventCreateWithFlags(&sg.done, hipEventDisableTiming));
        sg.device = dev;
    }
    if (sg.cap < bytes) {
        if (sg.host) (void)hipHostFree(sg.host);
        sg.host = nullptr;
        sg.cap = 0;
        const size_t cap = bytes + bytes / 2 + 4096;
        HIP_TRY(hipHostMalloc(&sg.host, cap, hipHostMallocDefault));
        sg.cap = cap;
    }
    *out = &sg;
    return FFS_OK;
}
}  // namespace

int ffs_rasterize_batch_bits(const int64_t* start_us, const int64_t* end_us, const uint8_t* is_metadata, int64_t n_subs_total,
                             const int64_t* vec_sub_first, const int64_t* vec_sub_count, const double* vec_ratio,
                             const int64_t* vec_out_word, const int64_t* vec_len, int64_t n_vec, double sample_rate,
                             double start_seconds, uint32_t* out_dev, int64_t out_words, void* hip_stream) {
    if (n_vec < 0 || n_subs_total < 0 || out_words < 0) return fail(FFS_E_INVALID, "bad argument");
    if (n_vec > 0 && (!vec_sub_first || !vec_sub_count || !vec_ratio || !vec_out_word || !vec_len))
        return fail(FFS_E_INVALID, "null vector table");
    if (n_subs_total > 0 && (!start_us || !end_us)) return fail(FFS_E_INVALID, "null subtitle arrays");
    if (out_words > 0 && (!out_dev || ((uintptr_t)out_dev & 3))) return fail(FFS_E_INVALID, "null or misaligned output");
    if (n_vec >= (int64_t(1) << 31)) return fail(FFS_E_INVALID, "too many vectors");
    std::vector<RasterVec> vecs((size_t)n_vec);
    int64_t longest = 0;
    for (int64_t v = 0; v < n_vec; ++v) {
        if (vec_sub_first[v] < 0 || vec_sub_count[v] < 0 || vec_sub_first[v] + vec_sub_count[v] > n_subs_total)
            return fail(FFS_E_INVALID, "vector %lld: subtitle range outside the arrays", (long long)v);
        if (vec_len[v] < 0 || vec_len[v] >= (int64_t(1) << 31)) return fail(FFS_E_TOO_LONG, "raster longer than 2^31 samples");
        if (vec_sub_count[v] >= (int64_t(1) << 31)) return fail(FFS_E_INVALID, "vector %lld: too many subtitles", (long long)v);
        if (vec_out_word[v] < 0 || vec_out_word[v] + (vec_len[v] + 31) / 32 > out_words)
            return fail(FFS_E_INVALID, "vector %lld: output range outside the buffer", (long long)v);
        vecs[(size_t)v] = RasterVec{(long long)vec_sub_first[v], (long long)vec_out_word[v], vec_ratio[v], (int)vec_sub_count[v],
                                    (int)vec_len[v]};
        if (vec_sub_count[v] > longest) longest = vec_sub_count[v];
    }
    if (out_words == 0) return FFS_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    DeviceGuard guard;
    int rc_dev;
    if ((rc_dev = guard.enter(out_dev))) return rc_dev;
    HIP_TRY(hipMemsetAsync(out_dev, 0, (size_t)out_words * 4, st));
    if (n_vec == 0 || longest == 0) return FFS_OK;
    // one staging allocation: start | end | vector table | metadata flags.  The tables are the caller's (and this
    // function's) pageable memory and must have been read before we return: they are copied into a pinned per-thread
    // staging buffer (two of them, alternating; a buffer is reused once the copy that read it has completed -- an event,
    // not a synchronisation of the caller's stream, which may hold a pipelined VAD sweep or the previous solve) and go
    // to the device in ONE asynchronous copy.
    const size_t nsub = (size_t)n_subs_total, off_end = nsub * 8, off_vec = 2 * nsub * 8;
    const size_t off_meta = off_vec + (size_t)n_vec * sizeof(RasterVec), total = off_meta + (is_metadata ? nsub : 0);
    PinnedStage* stage = nullptr;
    int rc = stage_acquire(total, &stage);
    if (rc) return rc;
    char* hs = (char*)stage->host;
    memcpy(hs, start_us, nsub * 8);
    memcpy(hs + off_end, end_us, nsub * 8);
    memcpy(hs + off_vec, vecs.data(), (size_t)n_vec * sizeof(RasterVec));
    if (is_metadata) memcpy(hs + off_meta, is_metadata, nsub);
    char* d = nullptr;
    HIP_TRY(hipMallocAsync((void**)&d, total, st));
    if (hipMemcpyAsync(d, hs, total, hipMemcpyHostToDevice, st) != hipSuccess) rc = fail(FFS_E_HIP, "copying the subtitle tables failed");
    if (rc == FFS_OK && hipEventRecord(stage->done, st) == hipSuccess) stage->pending = true;
    if (rc == FFS_OK) {
        const unsigned bx = (unsigned)((longest + 255) / 256);
        hipLaunchKernelGGL(k_rasterize_batch, dim3(bx < 64 ? bx : 64, (unsigned)(n_vec < 65535 ? n_vec : 65535)), dim3(256), 0, st,
                           (const long long*)d, (const long long*)(d + off_end),
                           is_metadata ? (const unsigned char*)(d + off_meta) : nullptr, (const RasterVec*)(d + off_vec),
                           (int)n_vec, sample_rate, start_seconds, out_dev);
        if (hipGetLastError() != hipSuccess) rc = fail(FFS_E_HIP, "k_rasterize_batch launch failed");
    }
    (void)hipFreeAsync(d, st);
    return rc;
}

int ffs_rasterize_batch_runs(const int64_t* start_us, const int64_t* end_us, const uint8_t* is_metadata, int64_t n_subs_total,
                             const int64_t* vec_sub_first, const int64_t* vec_sub_count, const double* vec_ratio,
                             const int64_t* vec_out_off, const int64_t* vec_cap, const int64_t* vec_len, int64_t n_vec,
                             double sample_rate, double start_seconds, void* out_dev, int64_t out_bytes, void* hip_stream) {
    if (n_vec < 0 || n_subs_total < 0 || out_bytes < 0) return fail(FFS_E_INVALID, "bad argument");
    if (n_vec > 0 && (!vec_sub_first || !vec_sub_count || !vec_ratio || !vec_out_off || !vec_cap || !vec_len))
        return fail(FFS_E_INVALID, "null vector table");
    if (n_subs_total > 0 && (!start_us || !end_us)) return fail(FFS_E_INVALID, "null subtitle arrays");
    if (n_vec > 0 && (!out_dev || ((uintptr_t)out_dev & 7))) return fail(FFS_E_INVALID, "null or misaligned output");
    if (n_vec >= (int64_t(1) << 31)) return fail(FFS_E_INVALID, "too many vectors");
    if (start_seconds > 0.0)
        return fail(FFS_E_INVALID, "start_seconds > 0 can make start samples negative (Python slices wrap them around): "
                    "use ffs_rasterize_batch_bits");
    if (n_vec == 0) return FFS_OK;
    // DEVICE-resident subtitle tables (the same tracks rasterised again and again: the steps of a golden-section search, a
    // subtitle file against many references): nothing of them is copied, only the vector table travels.  They must be
    // sorted by start time already (the host cannot look).
    bool tables_on_device = false;
    if (n_subs_total > 0) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, start_us) == hipSuccess && attr.type == hipMemoryTypeDevice) {
            tables_on_device = true;
            hipPointerAttribute_t a2;
            if (hipPointerGetAttributes(&a2, end_us) != hipSuccess || a2.type != hipMemoryTypeDevice)
                return fail(FFS_E_INVALID, "start_us is device memory, end_us is not: the subtitle tables must live on one side");
            if (is_metadata && (hipPointerGetAttributes(&a2, is_metadata) != hipSuccess || a2.type != hipMemoryTypeDevice))
                return fail(FFS_E_INVALID, "start_us is device memory, is_metadata is not: the subtitle tables must live on one side");
        } else {
            (void)hipGetLastError();
        }
    }
    // Tracks must reach the kernel sorted by start time.  Unsorted ones (rare) are sorted into extra rows behind the
    // caller's; vectors naming the same unsorted range share one copy.
    std::vector<int64_t> xs, xe;
    std::vector<uint8_t> xm;
    std::vector<RasterRunsVec> vecs((size_t)n_vec);
    int64_t last_first = -1, last_count = -1, last_new_first = -1;
    for (int64_t v = 0; v < n_vec; ++v) {
        const int64_t f = vec_sub_first[v], c = vec_sub_count[v];
        if (f < 0 || c < 0 || f + c > n_subs_total) return fail(FFS_E_INVALID, "vector %lld: subtitle range outside the arrays", (long long)v);
        if (c >= (int64_t(1) << 27)) return fail(FFS_E_INVALID, "vector %lld: too many subtitles", (long long)v);
        if (vec_len[v] < 0 || vec_len[v] >= (int64_t(1) << 30)) return fail(FFS_E_TOO_LONG, "raster longer than 2^30 - 1 samples");
        if (vec_cap[v] < 2 * c + 1 || vec_cap[v] >= (int64_t(1) << 28))
            return fail(FFS_E_INVALID, "vector %lld: list capacity %lld below 2 * %lld subtitles + 1", (long long)v, (long long)vec_cap[v], (long long)c);
        if (vec_out_off[v] < 0 || (vec_out_off[v] & 7) || vec_out_off[v] + ffs_runs_list_bytes(vec_cap[v]) > out_bytes)
            return fail(FFS_E_INVALID, "vector %lld: list block outside the buffer or misaligned", (long long)v);
        int64_t first = f;
        if (tables_on_device) {
            // sorted by contract
        } else if (f == last_first && c == last_count) {
            first = last_new_first;
        } else {
            // (branch-free per block of 64: the compiler vectorises the comparison; an early exit per element does not)
            bool sorted = true;
            for (int64_t i0 = f + 1; i0 < f + c && sorted; i0 += 64) {
                const int64_t i1 = i0 + 64 < f + c ? i0 + 64 : f + c;
                int ok = 1;
                for (int64_t i = i0; i < i1; ++i) ok &= (int)(start_us[i - 1] <= start_us[i]);
                sorted = ok != 0;
            }
            if (!sorted) {
                std::vector<int64_t> idx((size_t)c);
                for (int64_t i = 0; i < c; ++i) idx[(size_t)i] = f + i;
                std::stable_sort(idx.begin(), idx.end(), [&](int64_t x, int64_t y) { return start_us[x] < start_us[y]; });
                first = n_subs_total + (int64_t)xs.size();
                for (int64_t i : idx) {
                    xs.push_back(start_us[i]);
                    xe.push_back(end_us[i]);
                    xm.push_back(is_metadata ? is_metadata[i] : 0);
                }
            }
            last_first = f, last_count = c, last_new_first = first;
        }
        vecs[(size_t)v] = RasterRunsVec{(long long)first, (long long)vec_out_off[v], vec_ratio[v], (int32_t)c, (int32_t)vec_len[v],
                                        (int32_t)vec_cap[v], 0};
    }
    hipStream_t st = (hipStream_t)hip_stream;
    DeviceGuard guard;
    int rc_dev;
    if ((rc_dev = guard.enter(out_dev))) return rc_dev;
    // one staging allocation: start | end | vector table | metadata flags (see ffs_rasterize_batch_bits); with
    // device-resident tables only the vector table.  PINNED host tables (hipHostMalloc / hipHostRegister: the caller keeps
    // them alive and unchanged until the stream has passed this call) are copied to the device straight from where they
    // lie -- no staging copy on the host.
    bool tables_pinned = false;
    if (!tables_on_device && n_subs_total > 0 && xs.empty()) {
        hipPointerAttribute_t a0, a1, a2;
        tables_pinned = hipPointerGetAttributes(&a0, start_us) == hipSuccess && a0.type == hipMemoryTypeHost &&
                        hipPointerGetAttributes(&a1, end_us) == hipSuccess && a1.type == hipMemoryTypeHost &&
                        (!is_metadata || (hipPointerGetAttributes(&a2, is_metadata) == hipSuccess && a2.type == hipMemoryTypeHost));
        (void)hipGetLastError();
    }
    const bool stage_subs = !tables_on_device && !tables_pinned;
    const size_t nsub = tables_on_device ? 0 : (size_t)n_subs_total + xs.size(), off_end = nsub * 8, off_vec = 2 * nsub * 8;
    const bool with_meta = is_metadata != nullptr;
    const size_t off_meta = off_vec + (size_t)n_vec * sizeof(RasterRunsVec), total = off_meta + (with_meta && !tables_on_device ? nsub : 0);
    PinnedStage* stage = nullptr;
    int rc = stage_acquire(stage_subs ? total : (size_t)n_vec * sizeof(RasterRunsVec), &stage);
    if (rc) return rc;
    char* hs = (char*)stage->host;
    const size_t n0 = tables_on_device ? 0 : (size_t)n_subs_total;
    if (stage_subs) {
        if (n0) {
            memcpy(hs, start_us, n0 * 8);
            memcpy(hs + off_end, end_us, n0 * 8);
        }
        if (!xs.empty()) {
            memcpy(hs + n0 * 8, xs.data(), xs.size() * 8);
            memcpy(hs + off_end + n0 * 8, xe.data(), xe.size() * 8);
        }
        memcpy(hs + off_vec, vecs.data(), (size_t)n_vec * sizeof(RasterRunsVec));
        if (with_meta) {
            if (n0) memcpy(hs + off_meta, is_metadata, n0);
            if (!xm.empty()) memcpy(hs + off_meta + n0, xm.data(), xm.size());
        }
    } else {
        memcpy(hs, vecs.data(), (size_t)n_vec * sizeof(RasterRunsVec));
    }
    char* d = nullptr;
    HIP_TRY(hipMallocAsync((void**)&d, total, st));
    if (stage_subs) {
        if (hipMemcpyAsync(d, hs, total, hipMemcpyHostToDevice, st) != hipSuccess) rc = fail(FFS_E_HIP, "copying the subtitle tables failed");
    } else {
        bool ok = hipMemcpyAsync(d + off_vec, hs, (size_t)n_vec * sizeof(RasterRunsVec), hipMemcpyHostToDevice, st) == hipSuccess;
        if (tables_pinned) {
            ok = ok && hipMemcpyAsync(d, start_us, n0 * 8, hipMemcpyHostToDevice, st) == hipSuccess;
            ok = ok && hipMemcpyAsync(d + off_end, end_us, n0 * 8, hipMemcpyHostToDevice, st) == hipSuccess;
            if (with_meta) ok = ok && hipMemcpyAsync(d + off_meta, is_metadata, n0, hipMemcpyHostToDevice, st) == hipSuccess;
        }
        if (!ok) rc = fail(FFS_E_HIP, "copying the subtitle tables failed");
    }
    if (rc == FFS_OK && hipEventRecord(stage->done, st) == hipSuccess) stage->pending = true;
    if (rc == FFS_OK) {
        const long long* d_start = tables_on_device ? (const long long*)start_us : (const long long*)d;
        const long long* d_end = tables_on_device ? (const long long*)end_us : (const long long*)(d + off_end);
        const unsigned char* d_meta = !with_meta ? nullptr : tables_on_device ? (const unsigned char*)is_metadata : (const unsigned char*)(d + off_meta);
        hipLaunchKernelGGL(k_rasterize_runs, dim3((unsigned)n_vec), dim3(256), 0, st, d_start, d_end, d_meta,
                           (const RasterRunsVec*)(d + off_vec), sample_rate, start_seconds, (char*)out_dev);
        if (hipGetLastError() != hipSuccess) rc = fail(FFS_E_HIP, "k_rasterize_runs launch failed");
    }
    (void)hipFreeAsync(d, st);
    return rc;
}

int ffs_runs_from_bits_batch(const uint32_t* const* bits_dev, const int64_t* len, void* const* list_dev, const int64_t* cap,
                             int64_t n_vec, void* hip_stream) {
    if (n_vec < 0 || (n_vec > 0 && (!bits_dev || !len || !list_dev || !cap))) return fail(FFS_E_INVALID, "bad argument");
    if (n_vec == 0) return FFS_OK;
    if (n_vec >= (int64_t(1) << 31)) return fail(FFS_E_INVALID, "too many vectors");
    for (int64_t v = 0; v < n_vec; ++v) {
        if (!bits_dev[v] || !list_dev[v]) return fail(FFS_E_INVALID, "vector %lld: null pointer", (long long)v);
        if (len[v] <= 0 || len[v] >= (int64_t(1) << 30))
            return fail(len[v] <= 0 ? FFS_E_EMPTY : FFS_E_TOO_LONG, "vector %lld of %lld samples", (long long)v, (long long)len[v]);
        if (cap[v] < 1 || cap[v] >= (int64_t(1) << 28)) return fail(FFS_E_INVALID, "vector %lld: list capacity %lld outside [1, 2^28)", (long long)v, (long long)cap[v]);
        if (((uintptr_t)bits_dev[v] & 3) || ((uintptr_t)list_dev[v] & 7)) return fail(FFS_E_INVALID, "vector %lld: misaligned pointer", (long long)v);
    }
    hipStream_t st = (hipStream_t)hip_stream;
    DeviceGuard guard;
    int rc_dev;
    if ((rc_dev = guard.enter(list_dev[0]))) return rc_dev;
    PinnedStage* stage = nullptr;
    int rc = stage_acquire((size_t)n_vec * sizeof(RunsRef), &stage);
    if (rc) return rc;
    RunsRef* hr = (RunsRef*)stage->host;
    for (int64_t v = 0; v < n_vec; ++v)
        hr[v] = RunsRef{(const int2*)((const char*)list_dev[v] + 16), (const int2*)list_dev[v], (const unsigned*)bits_dev[v], (int32_t)len[v],
                        (int32_t)cap[v]};
    RunsRef* d = nullptr;
    HIP_TRY(hipMallocAsync((void**)&d, (size_t)n_vec * sizeof(RunsRef), st));
    if (hipMemcpyAsync(d, hr, (size_t)n_vec * sizeof(RunsRef), hipMemcpyHostToDevice, st) != hipSuccess) rc = fail(FFS_E_HIP, "copying the vector table failed");
    if (rc == FFS_OK && hipEventRecord(stage->done, st) == hipSuccess) stage->pending = true;
    if (rc == FFS_OK) {
        runs_extract_launch((const RunsRef*)d, (size_t)n_vec, true, st);
        if (hipGetLastError() != hipSuccess) rc = fail(FFS_E_HIP, "k_runs_extract_lists launch failed");
    }
    (void)hipFreeAsync(d, st);
    return rc;
}

// Host-only.  One activity vector as the reference hands it over (float64): are its samples two-level, and if so
// which levels, and the samples as bits -- ONE pass over the array instead of numpy's five temporaries.  The AVX2
// bodies are chosen at run time (the library is built without -march).
namespace {
struct MinMax {
    double lo, hi;
    bool nan;
};
// bits of up to 32 samples: bit k = (x[k] == hi); *ok is cleared when a sample equals neither level
uint32_t word_scalar(const double* x, int cnt, double lo, double hi, bool* ok) {
    uint32_t bits = 0, good = 1;
    for (int k = 0; k < cnt; ++k) {
        const uint32_t is_hi = x[k] == hi;
        good &= is_hi | (uint32_t)(x[k] == lo);
        bits |= is_hi << k;
    }
    if (!good) *ok = false;
    return bits;
}
#if FFS_HOST_AVX2
__attribute__((target("avx2"))) bool pack_avx2(const double* x, int64_t n_full_words, double lo, double hi, uint32_t* words) {
    const __m256d vlo = _mm256_set1_pd(lo), vhi = _mm256_set1_pd(hi);
    int all_ok = 0xf;
    for (int64_t w = 0; w < n_full_words; ++w) {
        const double* p = x + 32 * w;
        uint32_t bits = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const __m256d v = _mm256_loadu_pd(p + 4 * g);
            const __m256d eh = _mm256_cmp_pd(v, vhi, _CMP_EQ_OQ), el = _mm256_cmp_pd(v, vlo, _CMP_EQ_OQ);
            bits |= (uint32_t)_mm256_movemask_pd(eh) << (4 * g);
            all_ok &= _mm256_movemask_pd(_mm256_or_pd(eh, el));
        }
        words[w] = bits;
    }
    return all_ok == 0xf;
}
#endif
}  // namespace

int ffs_two_level_pack(const double* x, int64_t n, double* lo_out, double* hi_out, uint32_t* words) {
    if (n <= 0 || !x || !lo_out || !hi_out || !words) return fail(FFS_E_INVALID, "bad argument");
#if FFS_HOST_AVX2
    static const bool avx2 = __builtin_cpu_supports("avx2");
#else
    const bool avx2 = false;
#endif
    (void)avx2;
    // ONE pass over the samples (round 6; before: a min/max pass, then the packing pass -- 2 x 5.8 MB per 2 h vector, and
    // the drop-in's host time is this memory traffic): the first two distinct values are the candidate levels, the packing
    // pass itself verifies that every sample equals one of them.
    MinMax m{x[0], x[0], false};
    {
        int64_t i = 1;
        while (i < n && x[i] == x[0]) ++i;  // (a constant vector: this IS the one pass)
        if (i < n) {
            m.lo = x[i] < x[0] ? x[i] : x[0];
            m.hi = x[i] < x[0] ? x[0] : x[i];
        }
        m.nan = !(x[0] == x[0]) || (i < n && !(x[i] == x[i]));
    }
    *lo_out = m.lo;
    *hi_out = m.hi;
    if (m.nan || !std::isfinite(m.lo) || !std::isfinite(m.hi)) return 0;
    const int64_t n_words = (n + 31) / 32, n_full = n / 32;
    (void)n_full;
    if (m.hi == m.lo) {
        memset(words, 0, (size_t)n_words * 4);
        return 1;
    }
    bool ok = true;
    int64_t w = 0;
#if FFS_HOST_AVX2
    if (avx2) {
        ok = pack_avx2(x, n_full, m.lo, m.hi, words);
        w = n_full;
    }
#endif
    for (; ok && w < n_words; ++w) {
        const int64_t i0 = w * 32;
        words[w] = word_scalar(x + i0, (int)(n - i0 < 32 ? n - i0 : 32), m.lo, m.hi, &ok);
    }
    return ok ? 1 : 0;
}

int ffs_pack_bits(const void* src_dev, int src_dtype, int64_t n, double threshold, uint32_t* dst_dev, void* hip_stream) {
    if (n < 0 || (src_dtype != FFS_DTYPE_U8 && src_dtype != FFS_DTYPE_F32)) return fail(FFS_E_INVALID, "bad argument");
    if (n == 0) return FFS_OK;
    if (!src_dev || !dst_dev || ((uintptr_t)dst_dev & 3)) return fail(FFS_E_INVALID, "null or misaligned buffer");
    DeviceGuard guard;
    int rc;
    if ((rc = guard.enter(dst_dev))) return rc;
    const long long n_words = (n + 31) / 32;
    long long blocks = (n_words + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (src_dtype == FFS_DTYPE_U8)
        hipLaunchKernelGGL((k_pack_bits<0>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, src_dev, (long long)n,
                           (float)threshold, dst_dev, n_words);
    else
        hipLaunchKernelGGL((k_pack_bits<1>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, src_dev, (long long)n,
                           (float)threshold, dst_dev, n_words);
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}

int ffs_scatter_segments(const float* seg_labels_dev, const int64_t* seg_src_off, const int64_t* seg_dst_start,
                         const int64_t* seg_len, int n_segments, float* out_dev, int64_t out_len, void* hip_stream) {
    if (n_segments < 0 || out_len < 0 || (n_segments > 0 && (!seg_src_off || !seg_dst_start || !seg_len)))
        return fail(FFS_E_INVALID, "bad argument");
    if (out_len == 0) return FFS_OK;
    if (!out_dev || (n_segments > 0 && !seg_labels_dev)) return fail(FFS_E_INVALID, "null device pointer");
    hipStream_t st = (hipStream_t)hip_stream;
    DeviceGuard guard;
    int rc;
    if ((rc = guard.enter(out_dev))) return rc;
    HIP_TRY(hipMemsetAsync(out_dev, 0, (size_t)out_len * sizeof(float), st));
    for (int s0 = 0; s0 < n_segments; s0 += SCATTER_MAX) {
        ScatterSegs segs;
        memset(&segs, 0, sizeof segs);
        long long longest = 0;
        for (int i = s0; i < n_segments && i < s0 + SCATTER_MAX; ++i) {
            if (seg_len[i] < 0 || seg_src_off[i] < 0) return fail(FFS_E_INVALID, "negative segment length / source offset");
            // Python slice assignment sparse[lo:hi] = labels[:hi-lo] with lo = int(start*sample_rate) >= 0
            if (seg_dst_start[i] < 0 || seg_dst_start[i] >= out_len || seg_len[i] == 0) continue;
            segs.src_off[segs.n] = seg_src_off[i];
            segs.dst_start[segs.n] = seg_dst_start[i];
            segs.len[segs.n] = seg_len[i];
            if (seg_len[i] > longest) longest = seg_len[i];
            ++segs.n;
        }
        if (!segs.n) continue;
        long long bx = (longest + 255) / 256;
        if (bx > 1024) bx = 1024;
        hipLaunchKernelGGL(k_scatter_segments, dim3((unsigned)bx, segs.n), dim3(256), 0, st, seg_labels_dev, segs, out_dev,
                           (long long)out_len);
    }
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}

// ---- RCCL gather of the per-pair results (the only collective on the path) ---------------------
// RCCL is resolved at run time (dlopen), so single-GPU users of libffsalign.so do not load it; in a
// torch process the already-loaded librccl.so.1 is picked up, so both share one RCCL.
namespace {
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, ffs_comm_id, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    static std::string load_error;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;  // the copy this process already has
        for (int i = 0; !api.handle && i < 3; ++i) api.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) {
            api.GetUniqueId = (int (*)(void*))dlsym(api.handle, "ncclGetUniqueId");
            api.CommInitRank = (int (*)(void**, int, ffs_comm_id, int))dlsym(api.handle, "ncclCommInitRank");
            api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.handle, "ncclAllGather");
            api.CommDestroy = (int (*)(void*))dlsym(api.handle, "ncclCommDestroy");
            api.GetErrorString = (const char* (*)(int))dlsym(api.handle, "ncclGetErrorString");
        } else {
            const char* e = dlerror();  // may be null
            load_error = e ? e : "no loader message";
        }
    });
    if (!api.handle || !api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) {
        fail(FFS_E_RCCL, "librccl.so.1 could not be loaded or lacks the nccl* entry points: %s",
             load_error.empty() ? "symbol missing" : load_error.c_str());
        return nullptr;
    }
    return &api;
}
int rccl_fail(RcclApi* a, const char* what, int code) {
    return fail(FFS_E_RCCL, "%s failed: %s", what, (a && a->GetErrorString) ? a->GetErrorString(code) : "?");
}
}  // namespace

struct ffs_comm {
    void* comm = nullptr;
    int device = 0, rank = 0, world = 1;
};

int ffs_comm_unique_id(ffs_comm_id* id_out) {
    if (!id_out) return fail(FFS_E_INVALID, "id_out is null");
    RcclApi* a = rccl_api();
    if (!a) return FFS_E_RCCL;  // message set by rccl_api()
    const int rc = a->GetUniqueId(id_out);
    return rc ? rccl_fail(a, "ncclGetUniqueId", rc) : FFS_OK;
}

int ffs_comm_create(int device, int rank, int world_size, const ffs_comm_id* id, ffs_comm** out) {
    if (!out || !id || world_size < 1 || rank < 0 || rank >= world_size) return fail(FFS_E_INVALID, "bad argument");
    *out = nullptr;
    RcclApi* a = rccl_api();
    if (!a) return FFS_E_RCCL;  // message set by rccl_api()
    HIP_TRY(hipSetDevice(device));
    ffs_comm* c = new ffs_comm();
    c->device = device;
    c->rank = rank;
    c->world = world_size;
    const int rc = a->CommInitRank(&c->comm, world_size, *id, rank);
    if (rc) {
        delete c;
        return rccl_fail(a, "ncclCommInitRank", rc);
    }
    *out = c;
    return FFS_OK;
}

int ffs_gather_results(ffs_comm* c, const ffs_pair_result* send_dev, int64_t n_local, ffs_pair_result* recv_dev,
                       void* hip_stream) {
    if (!c || n_local < 0) return fail(FFS_E_INVALID, "bad argument");
    if (n_local == 0) return FFS_OK;
    if (!send_dev || !recv_dev) return fail(FFS_E_INVALID, "null device pointer");
    RcclApi* a = rccl_api();
    if (!a) return fail(FFS_E_RCCL, "librccl.so.1 could not be loaded");
    HIP_TRY(hipSetDevice(c->device));
    const int rc = a->AllGather(send_dev, recv_dev, (size_t)n_local * sizeof(ffs_pair_result), /*ncclUint8*/ 1, c->comm,
                                (hipStream_t)hip_stream);
    return rc ? rccl_fail(a, "ncclAllGather", rc) : FFS_OK;
}

int ffs_comm_destroy(ffs_comm* c) {
    if (!c) return FFS_OK;
    RcclApi* a = rccl_api();
    if (a && c->comm) {
        (void)hipSetDevice(c->device);
        (void)a->CommDestroy(c->comm);
    }
    delete c;
    return FFS_OK;
}

int ffs_plan_set_algorithm(ffs_plan* p, int algorithm) {
    if (!p) return fail(FFS_E_INVALID, "plan is null");
    if (algorithm != FFS_ALGO_AUTO && algorithm != FFS_ALGO_FFT && algorithm != FFS_ALGO_RUNS)
        return fail(FFS_E_INVALID, "unknown algorithm %d", algorithm);
    p->algo = algorithm;
    return FFS_OK;
}

int ffs_plan_runs_stats(ffs_plan* p, int64_t* calls, int64_t* sub_batches, int64_t* sub_batches_through_transforms,
                        int64_t* boundaries_last_call) {
    if (!p) return fail(FFS_E_INVALID, "plan is null");
    if (calls) *calls = p->runs_calls;
    if (sub_batches) *sub_batches = p->runs_chunks;
    if (sub_batches_through_transforms) *sub_batches_through_transforms = p->runs_fft_chunks;
    if (boundaries_last_call) *boundaries_last_call = p->runs_last_boundaries;
    return FFS_OK;
}

int ffs_plan_profile(ffs_plan* p, int enable) {
    if (!p) return fail(FFS_E_INVALID, "plan is null");
    p->profiling = enable != 0;
    return FFS_OK;
}

int ffs_plan_profile_read(ffs_plan* p, double* ms_total, int64_t* launches) {
    if (!p || !ms_total || !launches) return fail(FFS_E_INVALID, "null argument");
    for (auto& sp : p->ev_spans) {
        HIP_TRY(hipEventSynchronize(sp.second.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, sp.second.first, sp.second.second));
        ms_total[sp.first] += (double)ms;
        launches[sp.first] += 1;
    }
    p->ev_spans.clear();
    p->ev_used = 0;
    return FFS_OK;
}

static int vad_energy_impl(const int16_t* pcm_dev, int64_t n_samples, int frame_len, double energy_threshold_db,
                           float non_speech_label, float* labels_dev, uint8_t* bits_dev, void* hip_stream) {
    if (n_samples < 0 || frame_len < 1) return fail(FFS_E_INVALID, "bad n_samples/frame_len");
    if (n_samples == 0) return FFS_OK;
    if (!pcm_dev || (!labels_dev && !bits_dev)) return fail(FFS_E_INVALID, "null device pointer");
    DeviceGuard guard;
    int rc_dev;
    if ((rc_dev = guard.enter(labels_dev ? (const void*)labels_dev : (const void*)bits_dev))) return rc_dev;
    const long long n_frames = (n_samples + frame_len - 1) / frame_len;
    const double thr_lin = pow(10.0, energy_threshold_db / 10.0);
    long long blocks = ((n_frames + VAD_FPT - 1) / VAD_FPT + 3) / 4;  // one wave per VAD_FPT frames, four waves per block
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(k_vad_energy, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, pcm_dev,
                       (long long)n_samples, frame_len, n_frames, thr_lin, non_speech_label, labels_dev, bits_dev);
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}

int ffs_vad_energy(const int16_t* pcm_dev, int64_t n_samples, int frame_len, double energy_threshold_db,
                   float non_speech_label, float* labels_dev, void* hip_stream) {
    if (n_samples > 0 && !labels_dev) return fail(FFS_E_INVALID, "null device pointer");
    return vad_energy_impl(pcm_dev, n_samples, frame_len, energy_threshold_db, non_speech_label, labels_dev, nullptr,
                           hip_stream);
}

int ffs_vad_energy_bits(const int16_t* pcm_dev, int64_t n_samples, int frame_len, double energy_threshold_db,
                        uint8_t* bits_dev, void* hip_stream) {
    if (n_samples > 0 && !bits_dev) return fail(FFS_E_INVALID, "null device pointer");
    return vad_energy_impl(pcm_dev, n_samples, frame_len, energy_threshold_db, 0.0f, nullptr, bits_dev, hip_stream);
}

int ffs_vad_tokenize(const float* valid_dev, int64_t n_frames, int64_t chunk_frames, int min_length, int max_length,
                     int max_continuous_silence, float non_speech_label, float* labels_dev, void* hip_stream) {
    if (n_frames < 0 || chunk_frames < 1 || max_length < 1) return fail(FFS_E_INVALID, "bad argument");
    if (n_frames == 0) return FFS_OK;
    if (!valid_dev || !labels_dev || valid_dev == labels_dev) return fail(FFS_E_INVALID, "null or aliased buffers");
    DeviceGuard guard;
    int rc_dev;
    if ((rc_dev = guard.enter(labels_dev))) return rc_dev;
    const long long chunks = (n_frames + chunk_frames - 1) / chunk_frames;
    const long long longest = chunk_frames < n_frames ? chunk_frames : n_frames;
    if (longest <= TOK_SCAN_MAX && max_length >= min_length && min_length >= 0) {
        // one workgroup per chunk (validity, island starts and markers as bit words in LDS: ffs_kernels.h)
        hipLaunchKernelGGL(k_vad_tokenize_scan, dim3((unsigned)chunks), dim3(TOK_THREADS), 0, (hipStream_t)hip_stream, valid_dev,
                           (long long)n_frames, (long long)chunk_frames, min_length, max_length, max_continuous_silence,
                           non_speech_label, labels_dev);
    } else {
        hipLaunchKernelGGL(k_vad_tokenize, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, (hipStream_t)hip_stream,
                           valid_dev, (long long)n_frames, (long long)chunk_frames, min_length, max_length,
                           max_continuous_silence, non_speech_label, labels_dev);
    }
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}

int ffs_speech_bounds(const float* frames_dev, int64_t n_frames, int64_t* bounds_dev, void* hip_stream) {
    if (!bounds_dev || (n_frames > 0 && !frames_dev) || n_frames < 0) return fail(FFS_E_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)hip_stream;
    DeviceGuard guard;
    int rc_dev;
    if ((rc_dev = guard.enter(bounds_dev))) return rc_dev;
    hipLaunchKernelGGL(k_bounds_init, dim3(1), dim3(1), 0, st, (long long*)bounds_dev);
    if (n_frames > 0) {
        long long blocks = (n_frames + 255) / 256;
        if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL(k_speech_bounds, dim3((unsigned)blocks), dim3(256), 0, st, frames_dev, (long long)n_frames,
                           (long long*)bounds_dev);
    }
    hipLaunchKernelGGL(k_bounds_fix, dim3(1), dim3(1), 0, st, (long long*)bounds_dev);
    HIP_TRY(hipGetLastError());
    return FFS_OK;
}

}  // extern "C"
