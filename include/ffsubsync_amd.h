/*
 * ffsubsync_amd.h -- C ABI of libffsalign.so, the MI355X (gfx950) implementation of
 * ffsubsync's alignment hot path.
 *
 * The reference (smacke/ffsubsync) is pure Python and has no FFI of its own for this path;
 * each entry point below names the reference interface (file:line under
 * /root/reference/ffsubsync/) whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success or a negative FFS_E_* code; nothing throws across
 *     the ABI; ffs_last_error() returns a thread-local message for the last failure.
 *   - "_dev" pointers are device (HBM) addresses owned by the caller (e.g. torch tensors'
 *     data_ptr()); all other pointers are host memory.  `hip_stream` is a hipStream_t
 *     (0 = the null stream).  Calls are asynchronous with respect to the host unless stated;
 *     results land in caller-owned device buffers in stream order.  ffs_align_batch* may copy
 *     their own descriptors to the device on an internal copy stream (one per device); the kernels
 *     that read the caller's vectors and write its results run on `hip_stream` only, after the
 *     stream has waited for those copies.
 *   - a plan owns its twiddle tables and workspace in HBM and may be used by one host thread
 *     at a time; successive calls on different streams are ordered by the library (a call waits
 *     for the plan's previous call before it touches the workspace).
 *   - entry points without a plan run on the device that owns their output buffer.
 */
#ifndef FFSUBSYNC_AMD_H
#define FFSUBSYNC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFS_OK 0
#define FFS_E_INVALID (-1) /* bad argument */
#define FFS_E_HIP (-2)     /* a HIP runtime call failed */
#define FFS_E_NOMEM (-3)
#define FFS_E_TOO_LONG (-4) /* R+S exceeds the plan's transform length */
#define FFS_E_EMPTY (-5)    /* empty reference or candidate (aligners.py:58-66) */
#define FFS_E_RCCL (-6)     /* librccl could not be loaded or an RCCL call failed */

/* element types of the activity vectors */
#define FFS_DTYPE_U8 0  /* two-level signal: byte==0 -> lo, byte!=0 -> hi  */
#define FFS_DTYPE_F32 1 /* arbitrary float samples (lo/hi = bounds, used for the tie margin) */
#define FFS_DTYPE_F64 3 /* as FFS_DTYPE_F32 with double samples: the transforms still nominate in fp32, the winning
                           lags are re-evaluated in fp64 from the caller's own samples (the reference's arithmetic,
                           aligners.py:55-57, without any input rounding) */
#define FFS_DTYPE_U1 2  /* two-level signal, one bit per sample: sample i = bit (i & 31) of the 32-bit
                           little-endian word i >> 5 (numpy.packbits(..., bitorder="little")); 0 -> lo, 1 -> hi.
                           The native format of the 0/1 activity vectors: an eighth of the HBM and PCIe bytes of
                           FFS_DTYPE_U8.  Pointers 4-byte aligned; the buffer must cover whole 32-bit words. */
#define FFS_DTYPE_RUNS 5 /* two-level signal as its BOUNDARY LIST (what speech_transformers.py:957-980 is handed: the
                           subtitles' intervals).  The pointer is a device block `ffs_runs_list` (8-byte aligned):
                             int32 n, ones, len, cap;           -- boundaries, samples at the upper level, samples, entries
                             struct { int32 pos, ones_before; } e[cap];
                           e[k].pos (k < n, ascending, n even) = a sample where the value changes -- even k: a run of
                           the upper level starts, odd k: one past its last sample (a run that reaches the end closes at
                           `len`); e[k].ones_before = upper-level samples in front of it; e[n] = (INT32_MAX, ones).
                           Producers: ffs_rasterize_batch_runs, ffs_runs_from_bits.  A list with n >= cap is truncated
                           and unusable.  ffs_runs_list_bytes(cap) = 16 + 8 * cap. */

/* result flags */
#define FFS_FLAG_EMPTY_WINDOW 1 /* every lag masked: score=-inf, offset=N-1-S (aligners.py:45-48) */
#define FFS_FLAG_AMBIGUOUS 2    /* more near-ties than the nominee list holds; best of list returned */
#define FFS_FLAG_FILTERED 4     /* |offset| > filter_max_offset: dropped by MaxScoreAligner.transform */
#define FFS_FLAG_DIRECT 8       /* solved by the exact direct-correlation kernel (short inputs) */

typedef struct ffs_plan ffs_plan;

/* One FFTAligner solve: FFTAligner.best_score_ / best_offset_ (aligners.py:45-48). */
typedef struct ffs_cand_result {
    double score;    /* correlation at `offset`, re-evaluated exactly (integer counts for
                        two-level inputs, fp64 dot product for float inputs) */
    int64_t offset;  /* samples; subtitle must be shifted by +offset/sample_rate seconds */
    float score_f32; /* the fp32 FFT's value at that lag (diagnostic) */
    int32_t flags;
} ffs_cand_result;

/* One MaxScoreAligner solve: the winning ((score, offset), candidate) (aligners.py:154-167). */
typedef struct ffs_pair_result {
    double score;
    int64_t offset;
    int32_t best_cand; /* index into the pair's candidate list; -1 = none survived the filter */
    int32_t flags;
} ffs_pair_result;

/* Smallest transform length the reference would use for these lengths:
 * 2**ceil(log2(R+S)) (aligners.py:67-68).  Returns 0 if either length is <= 0. */
int64_t ffs_fft_length(int64_t ref_len, int64_t sub_len);

/* Smallest supported transform length (2^k, or 3*2^k in [12288, 3145728]) a plan needs for one
 * (reference, candidate) solve.
 * Only lags with a non-empty overlap, d in (-S, R), ever need a transform: every other lag the reference
 * looks at has c(d) = 0 exactly and is handled as one virtual nominee (score 0, the largest such lag).
 * Without a lag window the R+S-1 overlap lags need a circular correlation of length >= R+S (3*2^19 =
 * 1 572 864 instead of the reference's 2^21 for 2 h @ 100 Hz inputs).  With
 * max_offset_samples >= 0 only lags inside the reference's window [d_lo, d_hi] are ever looked at:
 * only the prefixes S' = min(S, R - d_lo) and R' = min(R, S + d_hi) of the two vectors can meet at
 * such a lag (the rest is multiplied by zero padding), and a circular correlation of any length
 * n >= max(S' + d_hi, R' - d_lo) + 1 over them reproduces those lags exactly (no product wraps), so
 * a shorter transform gives bit-identical results:
 * 3*2^18 = 786 432 instead of 2^21 for 2 h @ 100 Hz inputs with the default +-6000 window.
 * (FFS_DISABLE_RADIX3=1 in the environment restricts the answer to powers of two.) */
int64_t ffs_plan_length(int64_t ref_len, int64_t sub_len, int64_t max_offset_samples);

/* Create a plan for transform length n_fft (a power of two in [2, 2^24], or 3*2^k in
 * [12288, 3145728]) on `device`.
 * pairs_in_flight: how many (reference, candidates) problems share one sweep of the
 * A/mid/C kernels (sizes the workspace: pairs_in_flight * (1+ceil(max_cand/2)) * n_fft * 8 B; twice
 * that for n_fft = 3*2^k >= 3*2^16, which also carries the block-segmented pipeline used under
 * narrow lag windows).
 * Replaces: the per-call np.fft plan + temporaries of FFTAligner.fit (aligners.py:67-74). */
int ffs_plan_create(int device, int64_t n_fft, int pairs_in_flight, int max_cand, ffs_plan** out);
int ffs_plan_destroy(ffs_plan* plan);
int64_t ffs_plan_workspace_bytes(const ffs_plan* plan);

/* Batched MaxScoreAligner(FFTAligner).fit(...).transform() over n_pairs independent problems,
 * each one reference vector and n_cand candidate vectors (aligners.py:50-80, 131-167).
 *
 * Vectors are listed pair-major: index p*(1+n_cand) is pair p's reference, the next n_cand
 * entries its candidates.  vec_ptr[i] is a DEVICE pointer to vec_len[i] elements of `dtype`
 * (vec_len counts samples for every dtype);
 * vec_lo/vec_hi give the two sample values of a FFS_DTYPE_U8 / FFS_DTYPE_U1 vector *before* the reference's
 * 2*x-1 map (aligners.py:55-57), e.g. (0, 1) for a 0/1 vector or (0, 1/ratio) for a subtitle
 * track rasterised at a framerate ratio > 1 (speech_transformers.py:977).
 *
 * max_offset_samples: FFTAligner(max_offset_samples) lag window, -1 = None; the window is the
 *   reference's, including Python negative-slice semantics (aligners.py:31-43).
 * filter_max_offset: MaxScoreAligner.max_offset_samples used to drop candidates
 *   (aligners.py:156-159), -1 = None.
 * cand_out_dev[n_pairs*n_cand], pair_out_dev[n_pairs]: device buffers, written in stream order.
 * Every (R, S) must satisfy ffs_plan_length(R, S, max_offset_samples) <= plan n_fft.
 * Host arrays may be freed after return. */
int ffs_align_batch(ffs_plan* plan, int n_pairs, int n_cand, int dtype,
                    const void* const* vec_ptr, const int64_t* vec_len,
                    const double* vec_lo, const double* vec_hi,
                    int64_t max_offset_samples, int64_t filter_max_offset,
                    ffs_cand_result* cand_out_dev, ffs_pair_result* pair_out_dev,
                    void* hip_stream);

/* The same solve with one element type PER VECTOR: vec_dtype[i] is the FFS_DTYPE_* of vec_ptr[i] (pair-major like the
 * other arrays).  Within one call every reference must share one type and every candidate one type; the two may
 * differ.  The case this exists for: a multi-level float reference -- the `weighted` fused VAD emits
 * {0, 0.4, 0.6, 1} (speech_transformers.py:290-293), non_speech_label may be != 0 -- against two-level subtitle
 * rasters (speech_transformers.py:957-980), which then stay bit-packed (FFS_DTYPE_U1): the first pass reads each role
 * in its own format, the fp32 transforms nominate, and the nominees are re-evaluated in fp64 from the caller's own
 * samples (sum_i s'[i] * (2 r[i+d] - 1) with s' the candidate's fp64 level values).  vec_lo / vec_hi of a float vector
 * are bounds of its samples (used for the tie margin).  With all types equal this is ffs_align_batch.
 * Replaces: aligners.py:50-80, 131-167 for float-valued inputs. */
int ffs_align_batch_typed(ffs_plan* plan, int n_pairs, int n_cand, const int32_t* vec_dtype,
                          const void* const* vec_ptr, const int64_t* vec_len,
                          const double* vec_lo, const double* vec_hi,
                          int64_t max_offset_samples, int64_t filter_max_offset,
                          ffs_cand_result* cand_out_dev, ffs_pair_result* pair_out_dev,
                          void* hip_stream);

/* ffs_align_batch_typed plus, for FFS_DTYPE_RUNS vectors, vec_max_boundaries[i] (may be NULL; other types: ignored) = a
 * HOST-KNOWN upper bound of list i's length (0 = unknown) -- the rasteriser's lists have at most two entries per
 * subtitle.  When every vector of the call is a list with a bound and no candidate can exceed the coincidence budget
 * even at the bounds, the call queues its kernels and returns without waiting for anything from the device; otherwise it
 * waits once for one int per sub-batch (which sub-batches need the transforms).  A call whose lists break their stated
 * bounds has undefined results.
 * Replaces: aligners.py:50-80, 131-167 fed straight from speech_transformers.py:957-980 (no raster in between). */
int ffs_align_batch_runs(ffs_plan* plan, int n_pairs, int n_cand, const int32_t* vec_dtype,
                         const void* const* vec_ptr, const int64_t* vec_len,
                         const double* vec_lo, const double* vec_hi, const int32_t* vec_max_boundaries,
                         int64_t max_offset_samples, int64_t filter_max_offset,
                         ffs_cand_result* cand_out_dev, ffs_pair_result* pair_out_dev,
                         void* hip_stream);

/* Bytes of an `ffs_runs_list` block with room for `cap` entries (sentinel included): 16 + 8 * cap. */
int64_t ffs_runs_list_bytes(int64_t cap);
/* Boundary list of a bit-packed vector (FFS_DTYPE_U1, `len` samples at bits_dev) into the caller's block list_dev of
 * capacity `cap` entries (8-byte aligned, ffs_runs_list_bytes(cap) bytes): one pass over the bits.  A vector with cap
 * boundaries or more leaves n >= cap in the header (truncated).  Convert once -- VAD labels, a deserialised reference
 * (speech_transformers.py:993-1005) -- and every later solve skips the pass over the bits. */
int ffs_runs_from_bits(const uint32_t* bits_dev, int64_t len, void* list_dev, int64_t cap, void* hip_stream);
/* The same for n_vec vectors in one launch (host tables of device pointers / lengths / capacities). */
int ffs_runs_from_bits_batch(const uint32_t* const* bits_dev, const int64_t* len, void* const* list_dev, const int64_t* cap,
                             int64_t n_vec, void* hip_stream);
/* The inverse (parity tests; the transform path uses the same kernel for list-only vectors): `len` samples of the
 * list's vector as FFS_DTYPE_U1 words at bits_out_dev (ceil(len/32) words). */
int ffs_runs_to_bits(const void* list_dev, int64_t len, uint32_t* bits_out_dev, void* hip_stream);

/* How ffs_align_batch / ffs_align_batch_typed evaluate the correlation of aligners.py:50-80.  Results are identical
 * either way (same exact scores, same tie rule); only the time differs.
 *   FFS_ALGO_AUTO (default): two-level vectors given as bits or as boundary lists (FFS_DTYPE_U1 / FFS_DTYPE_RUNS on both
 *     sides) first go through the run-boundary path -- the exact integer correlation of the two run-length-coded vectors
 *     over every lag of the window, no transform (csrc/ffs_runs.h) -- and the call waits once (an event, not the stream,
 *     after all of its kernels are queued) for one int per sub-batch: whether it needs the transforms; sub-batches (pairs_in_flight pairs) holding a vector with 32 768 boundaries or more, or a candidate
 *     whose expected number of boundary coincidences inside its lag window (boundaries of the candidate x boundaries of
 *     the reference x window lags / reference length) exceeds the budget -- by default twelve per point of the plan's
 *     transform length and packed transform slot the candidate occupies, (n_cand + 1) / (2 n_cand) of one: the measured
 *     break-even -- are solved by the transforms instead.  A FLOAT reference (FFS_DTYPE_F32 / F64) with at most four
 *     distinct sample values whose steps are small integer multiples of one quantum -- the `weighted` fused VAD's
 *     {l, .4 + .6 l, .6 + .4 l, 1}, speech_transformers.py:290-293 -- against two-level candidates takes the same path:
 *     its threshold vectors are made on the device and their coincidences added with integer multiplicities; any other
 *     float vector (more levels, no common quantum, noise) and every other element type go through the transforms.
 *   FFS_ALGO_FFT: transforms only (the path of rounds 1-3).
 *   FFS_ALGO_RUNS: like AUTO without the coincidence budget (truncated boundary lists still fall back).
 *   Boundary lists of vectors that arrive as bits live in the plan, 4 096 entries per vector to start with (FFS_RUNS_STRIDE);
 *   a call that meets a longer list is solved again with four times the room (at most twice, up to 32 768 entries) and
 *   the plan keeps the longer stride -- which path solves what does not depend on it.
 * Environment: FFS_ALGORITHM=auto|fft|runs presets new plans, FFS_RUNS_BUDGET=<coincidences> the budget. */
#define FFS_ALGO_AUTO 0
#define FFS_ALGO_FFT 1
#define FFS_ALGO_RUNS 2
int ffs_plan_set_algorithm(ffs_plan* plan, int algorithm);
/* Since plan creation: calls that tried the run-boundary path, their sub-batches, and how many of those went through
 * the transforms after all; boundaries_last_call = boundary-list entries of all vectors of the most recent such call
 * (what k_runs_extract wrote: 8 bytes each).  Any pointer may be NULL. */
int ffs_plan_runs_stats(ffs_plan* plan, int64_t* calls, int64_t* sub_batches, int64_t* sub_batches_through_transforms,
                        int64_t* boundaries_last_call);

/* Full correlation of one reference with one or two candidates (b_dev may be NULL):
 *   out_x_dev[m] = sum_i x'[i] * ref'[(i + m) mod n_fft],  m in [0, n_fft)
 * i.e. the reference's `convolve` array (aligners.py:74) with convolve[k] = out[(N-1-S-k) mod N].
 * Used by the parity tests to compare the raw fp32 correlation against np.fft. */
int ffs_correlate_full(ffs_plan* plan, int dtype,
                       const void* ref_dev, int64_t ref_len, double ref_lo, double ref_hi,
                       const void* a_dev, int64_t a_len, double a_lo, double a_hi,
                       const void* b_dev, int64_t b_len, double b_lo, double b_hi,
                       float* out_a_dev, float* out_b_dev, void* hip_stream);

/* Frame-energy voice-activity sweep over s16le mono PCM resident in HBM.
 * labels_dev[f] = (10*log10(mean(x^2) over frame f) >= energy_threshold_db) ? 1.0f
 *                                                                           : non_speech_label
 * for f in [0, ceil(n_samples/frame_len)); the last frame may be short.  Replaces the
 * per-frame Python loop of the detector closures (speech_transformers.py:133-150, 169-181)
 * with the AudioEnergyValidator rule (threshold 50 dB in the reference, :124). */
int ffs_vad_energy(const int16_t* pcm_dev, int64_t n_samples, int frame_len,
                   double energy_threshold_db, float non_speech_label,
                   float* labels_dev, void* hip_stream);

/* The same sweep with the labels written bit-packed, straight into the form the aligner reads
 * (FFS_DTYPE_U1): bit (f & 7) of bits_dev[f >> 3] = 1 iff frame f is speech, for
 * f in [0, ceil(n_samples/frame_len)); the unused high bits of the last byte are written as 0.
 * ceil(n_frames/8) bytes are written, nothing beyond them -- a caller that sweeps a file chunk by
 * chunk (the reference's 100 s buffers, speech_transformers.py:683-685, are 10 000 frames = 1250
 * bytes) points each call at byte (first_frame / 8) of one zero-initialised word buffer.  Replaces
 * the label vector of speech_transformers.py:133-150 plus the host-side 0/1 conversion in front of
 * aligners.py:55-57. */
int ffs_vad_energy_bits(const int16_t* pcm_dev, int64_t n_samples, int frame_len,
                        double energy_threshold_db, uint8_t* bits_dev, void* hip_stream);

/* Token smoothing of a frame-validity sweep, as the reference's auditok detector applies it
 * (speech_transformers.py:125-131, 140-150): auditok 0.1.5's StreamTokenizer state machine
 * (min_length, max_length, max_continuous_silence in frames; default mode) run independently on
 * every chunk of `chunk_frames` frames (the reference re-opens the tokenizer per 100 s buffer), then
 * the reference's rasterisation: marker[start] = 1, marker[end+1] = non_speech_label - 1 (assigned in
 * token order), out = clip(cumsum(marker)[:-1], 0, 1).
 * valid_dev[f] != 0  <=>  frame f passed the energy test.  PARITY UNPINNED: auditok is not available
 * to check against; restated from its published source (oracle/vad_oracle.py, tokenize()).
 * Chunks of up to 28672 frames with max_length >= min_length >= 0 run as one workgroup per chunk (k_vad_tokenize_scan:
 * validity and island starts as bit words, markers written island by island; model:
 * oracle/vad_oracle.py::tokenize_chunk_words); anything else as one thread per chunk walking the state machine.
 * Identical outputs. */
int ffs_vad_tokenize(const float* valid_dev, int64_t n_frames, int64_t chunk_frames, int min_length,
                     int max_length, int max_continuous_silence, float non_speech_label,
                     float* labels_dev, void* hip_stream);

/* ComputeSpeechFrameBoundariesMixin.fit_boundaries (speech_transformers.py:310-317):
 * bounds_dev[0] = first index with frames[i] > 0.5, bounds_dev[1] = last such index;
 * both -1 when there is none. */
int ffs_speech_bounds(const float* frames_dev, int64_t n_frames, int64_t* bounds_dev,
                      void* hip_stream);

/* ---- subtitle rasteriser (the step immediately before the aligner) ----------
 * SubtitleScaler.fit (subtitle_transformers.py:35-47) followed by SubtitleSpeechTransformer.fit
 * (speech_transformers.py:957-980) for one framerate ratio, from subtitle start/end times given as
 * integer microseconds (datetime.timedelta's resolution) in HOST arrays:
 *   scaled = timedelta(seconds = total_seconds * ratio)         (microsecond rounding, half-even)
 *   start  = int(round((scaled_start - start_seconds) * sample_rate))
 *   end    = start + int(round((scaled_end - scaled_start) * sample_rate))
 *   out[start:end] = 1   (Python slice semantics, union over subtitles; metadata lines skipped)
 * ffs_raster_length = int(max scaled end * sample_rate) + 2 over ALL subtitles (metadata too).
 * The sample value min(1/ratio, 1) (speech_transformers.py:977) is passed to ffs_align_batch as the
 * vector's `hi` level; out_dev holds 0/1 bytes. */
int64_t ffs_raster_length(const int64_t* end_us, int64_t n_subs, double ratio, double sample_rate);
/* Host-only, many vectors at once: len_out[v] = ffs_raster_length of a track whose largest end time is
 * track_end_us_max[v], scaled by ratio[v] (the scaling is monotone, so the largest end decides). */
int ffs_raster_lengths(const int64_t* track_end_us_max, const double* ratio, int64_t n_vec, double sample_rate,
                       int64_t* len_out);
/* Host-only: the clamped [start, end) sample intervals ffs_rasterize_subtitles fills, written as
 * pairs into iv_out[2*n_subs]; returns how many intervals were produced. */
int64_t ffs_raster_intervals(const int64_t* start_us, const int64_t* end_us, const uint8_t* is_metadata,
                             int64_t n_subs, double ratio, double sample_rate, double start_seconds,
                             int64_t out_len, int32_t* iv_out);
int ffs_rasterize_subtitles(const int64_t* start_us, const int64_t* end_us, const uint8_t* is_metadata,
                            int64_t n_subs, double ratio, double sample_rate, double start_seconds,
                            uint8_t* out_dev, int64_t out_len, void* hip_stream);

/* Bit-packed (FFS_DTYPE_U1) variant: out_dev holds ceil(out_len/32) 32-bit words (zeroed, then filled). */
int ffs_rasterize_subtitles_bits(const int64_t* start_us, const int64_t* end_us, const uint8_t* is_metadata,
                                 int64_t n_subs, double ratio, double sample_rate, double start_seconds,
                                 uint32_t* out_dev, int64_t out_len, void* hip_stream);

/* Batched form, interval arithmetic on the device: n_vec bit-packed rasters from one call -- a file's seven framerate
 * ratios, or every vector of a batch of files, written straight into the buffer ffs_align_batch reads.  The subtitle
 * tracks are concatenated in start_us / end_us / is_metadata (host arrays of n_subs_total entries; is_metadata may be
 * NULL); vector v rasterises the subtitles [vec_sub_first[v], vec_sub_first[v] + vec_sub_count[v]) -- several vectors
 * may name the same track -- with times scaled by vec_ratio[v], as vec_len[v] samples (ffs_raster_length) whose bit 0
 * is bit 0 of word out_dev[vec_out_word[v]].  out_dev[0, out_words) is zeroed first.  Same results, bit for bit, as
 * one ffs_rasterize_subtitles_bits call per vector (the arithmetic is IEEE fp64 on both sides).  The host arrays may
 * be freed after return (they are staged in pinned memory; the caller's stream is NOT synchronised); the upload and
 * the rasterisation are enqueued on hip_stream. */
int ffs_rasterize_batch_bits(const int64_t* start_us, const int64_t* end_us, const uint8_t* is_metadata,
                             int64_t n_subs_total, const int64_t* vec_sub_first, const int64_t* vec_sub_count,
                             const double* vec_ratio, const int64_t* vec_out_word, const int64_t* vec_len, int64_t n_vec,
                             double sample_rate, double start_seconds, uint32_t* out_dev, int64_t out_words,
                             void* hip_stream);

/* The same tracks as BOUNDARY LISTS (FFS_DTYPE_RUNS): no bitmap is written -- the union of a track's scaled sample
 * intervals, overlapping and touching subtitles merged, IS the list (samples[start:end] = ... for every subtitle,
 * speech_transformers.py:966-977, then read back as runs).  Vector v's block starts at byte vec_out_off[v] of out_dev
 * (multiple of 8) and has room for vec_cap[v] >= 2 * vec_sub_count[v] + 1 entries (ffs_runs_list_bytes).  Expanded to
 * bits (ffs_runs_to_bits) a list equals ffs_rasterize_batch_bits' raster bit for bit.  Subtitles need not be sorted (a
 * track that is not sorted by start time is sorted in a staging copy); requires start_seconds <= 0 (a positive one can
 * make start samples negative, which Python's slice semantics wrap around: use the bit rasteriser then).
 * start_us / end_us / is_metadata may be DEVICE pointers (all of them): tracks that are rasterised again and again -- the
 * steps of a golden-section search, one subtitle file against many references -- are uploaded once, nothing of them is
 * copied per call; device-resident tracks must already be sorted by start time.  PINNED host tables (hipHostMalloc /
 * hipHostRegister, sorted) are copied to the device straight from the caller's memory, without a staging copy: the caller
 * keeps them alive and unchanged until the stream has passed the call (pageable tables may be freed after return).
 * Replaces: subtitle_transformers.py:35-47 + speech_transformers.py:957-980 for the device-resident pipeline. */
int ffs_rasterize_batch_runs(const int64_t* start_us, const int64_t* end_us, const uint8_t* is_metadata,
                             int64_t n_subs_total, const int64_t* vec_sub_first, const int64_t* vec_sub_count,
                             const double* vec_ratio, const int64_t* vec_out_off, const int64_t* vec_cap,
                             const int64_t* vec_len, int64_t n_vec, double sample_rate, double start_seconds,
                             void* out_dev, int64_t out_bytes, void* hip_stream);

/* Host-only helper of the drop-in classes: the float64 vectors FFTAligner.fit receives (aligners.py:51-57) are
 * two-level activity vectors in practice.  Returns 1 and writes lo = min, hi = max and the samples as bits
 * (bit i = (x[i] == hi); ceil(n/32) words; all zero when hi == lo) when every sample equals one of the two levels
 * and both are finite; returns 0 (words unspecified) otherwise -- such vectors go to the device as floats. */
int ffs_two_level_pack(const double* x, int64_t n, double* lo_out, double* hi_out, uint32_t* words);

/* Two-level vector -> FFS_DTYPE_U1 on the device.  src_dtype FFS_DTYPE_U8: bit = (byte != 0);
 * FFS_DTYPE_F32: bit = (x > threshold), e.g. VAD labels against 0.5 or (lo+hi)/2.  Writes
 * ceil(n/32) words; unused high bits of the last word are 0. */
int ffs_pack_bits(const void* src_dev, int src_dtype, int64_t n, double threshold, uint32_t* dst_dev,
                  void* hip_stream);

/* Sparse reference assembly of MultiSegmentVideoSpeechTransformer.fit (speech_transformers.py:871-890):
 * out = zeros(out_len); out[dst_start[i] : dst_start[i]+len[i]] = labels[src_off[i] : src_off[i]+len[i]]
 * for every sampled window i (clipped at out_len, Python slice semantics), in one device pass.
 * seg_* are HOST arrays of n_segments entries; seg_labels_dev holds the windows' VAD labels. */
int ffs_scatter_segments(const float* seg_labels_dev, const int64_t* seg_src_off, const int64_t* seg_dst_start,
                         const int64_t* seg_len, int n_segments, float* out_dev, int64_t out_len,
                         void* hip_stream);

/* ---- multi-GPU: RCCL all-gather of the per-pair results over xGMI --------------------
 * Problems are sharded by pair, one process per GPU; nothing is exchanged during the solves.  The
 * single collective of the path gathers every rank's n_local 24-byte ffs_pair_result records:
 *   recv_dev[r * n_local + i] = rank r's send_dev[i]      (ncclAllGather, latency-bound)
 * Bootstrap as with NCCL: rank 0 calls ffs_comm_unique_id and publishes the 128 bytes out of band
 * (torch.distributed's store, MPI, a file); every rank then calls ffs_comm_create.  librccl is
 * resolved with dlopen at the first call (the copy already loaded in the process, e.g. torch's).
 * Replaces: nothing in the reference (single process); the per-file call being sharded is
 * ffsubsync.py:230-235. */
typedef struct ffs_comm ffs_comm;
typedef struct ffs_comm_id {
    char internal[128];
} ffs_comm_id;
int ffs_comm_unique_id(ffs_comm_id* id_out);
int ffs_comm_create(int device, int rank, int world_size, const ffs_comm_id* id, ffs_comm** out);
int ffs_gather_results(ffs_comm* comm, const ffs_pair_result* send_dev, int64_t n_local,
                       ffs_pair_result* recv_dev, void* hip_stream);
int ffs_comm_destroy(ffs_comm* comm);

/* Per-kernel timing with HIP events recorded on the caller's stream around every launch of the
 * hot kernels (used by bench.py for the roofline figures).  Kernel ids: */
#define FFS_K_PASS_A 0   /* load/map/pad + column FFT + twiddle                    */
#define FFS_K_MID 1      /* row FFT * conj(ref spectrum), row FFT, twiddle (in place) */
#define FFS_K_PASS_C 2   /* column FFT + lag-window mask + block argmax nominees    */
#define FFS_K_NOMINEES 3 /* nominee gather per candidate                            */
#define FFS_K_RESCORE 4  /* exact re-evaluation of nominee lags                     */
#define FFS_K_RUNS_EXTRACT 5 /* run-boundary path: boundary lists of every vector (reads the bit-packed vectors) */
#define FFS_K_RUNS_CORR 6    /* run-boundary path: exact correlation over the lag window + argmax           */
#define FFS_K_LEVELS 7       /* multi-level float references: level detection + threshold planes (reads the float vectors) */
#define FFS_K_COUNT 8
int ffs_plan_profile(ffs_plan* plan, int enable);
/* Synchronises the recorded events, adds their durations to ms_total[FFS_K_COUNT] /
 * launches[FFS_K_COUNT] (caller-zeroed or accumulating) and clears the recording. */
int ffs_plan_profile_read(ffs_plan* plan, double* ms_total, int64_t* launches);

/* Thread-local description of the last error returned on this thread ("" if none). */
const char* ffs_last_error(void);

/* Library/ABI version (major*10000 + minor*100 + patch). */
int ffs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FFSUBSYNC_AMD_H */
