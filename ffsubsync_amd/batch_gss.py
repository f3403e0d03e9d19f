"""Batched golden-section search over the framerate ratio.

``MaxScoreAligner.fit_gss`` (ffsubsync/aligners.py:111-129) runs ``gss`` (golden_section_search.py:
15-74) per input file: 17 *dependent* rescale -> rasterise -> align evaluations on [0.9, 1.1].  Across
many files the k-th evaluations are independent, so they are issued as one device batch per step
while every file keeps exactly the reference's own evaluation sequence: same bracketing points,
same strict ``yc < yd`` branch, and only the evaluation flagged ``is_last_iter`` is recorded.
"""
import math
from typing import Callable, List, Sequence, Tuple

import numpy as np

from .aligners import MAX_FRAMERATE_RATIO, MIN_FRAMERATE_RATIO
from .golden_section_search import _INV_PHI, _INV_PHI_SQ


def gss_batch(evaluate: Callable[[np.ndarray, bool], np.ndarray], n_problems: int, a: float = MIN_FRAMERATE_RATIO,
              b: float = MAX_FRAMERATE_RATIO, tol: float = 1e-4):
    """Run ``n_problems`` independent golden-section searches in lock step.

    ``evaluate(x, is_last_iter)`` receives one abscissa per problem (float64 array of length
    n_problems) and returns the objective values (to be *minimised*, e.g. minus the alignment score).
    Returns (lo, hi, trace) with the final brackets and the list of (x_array, is_last) evaluated.
    The per-problem sequences are identical to ``gss(f_i, a, b, tol)`` run one at a time.
    """
    lo = np.full(n_problems, min(a, b), dtype=np.float64)
    hi = np.full(n_problems, max(a, b), dtype=np.float64)
    width0 = float(hi[0] - lo[0]) if n_problems else 0.0
    trace: List[Tuple[np.ndarray, bool]] = []
    if n_problems == 0 or width0 <= tol:
        return lo, hi, trace
    steps = int(math.ceil(math.log(tol / width0) / math.log(_INV_PHI)))
    width = np.full(n_problems, width0)

    def ev(x, last):
        trace.append((x.copy(), last))
        return np.asarray(evaluate(x, last), dtype=np.float64)

    x_left = lo + _INV_PHI_SQ * width
    x_right = lo + _INV_PHI * width
    y_left = ev(x_left, steps == 1)
    y_right = ev(x_right, steps == 1)
    for it in range(steps - 1):
        final = it == steps - 2
        width = width * _INV_PHI
        go_left = y_left < y_right  # strict, per problem (golden_section_search.py:56)
        # left branch: hi <- x_right, x_right <- x_left, new x_left; right branch: mirrored
        new_hi = np.where(go_left, x_right, hi)
        new_lo = np.where(go_left, lo, x_left)
        keep_x = np.where(go_left, x_left, x_right)
        keep_y = np.where(go_left, y_left, y_right)
        fresh = np.where(go_left, new_lo + _INV_PHI_SQ * width, new_lo + _INV_PHI * width)
        y_fresh = ev(fresh, final)
        lo, hi = new_lo, new_hi
        x_left = np.where(go_left, fresh, keep_x)
        y_left = np.where(go_left, y_fresh, keep_y)
        x_right = np.where(go_left, keep_x, fresh)
        y_right = np.where(go_left, keep_y, y_fresh)
    out_lo = np.where(y_left < y_right, lo, x_left)
    out_hi = np.where(y_left < y_right, x_right, hi)
    return out_lo, out_hi, trace


def fit_gss_batch(refs: Sequence, subtitle_records: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray]],
                  max_offset_samples=None, sample_rate: int = 100, start_seconds: float = 0, stats: dict = None):
    """``MaxScoreAligner(FFTAligner(max_offset_samples)).fit_gss`` for many files at once.

    refs[i]: reference activity vector of file i (array or DeviceRaster); subtitle_records[i]:
    (start_us, end_us, is_metadata) of its subtitles.  Returns a list of ((score, offset), ratio) -- the evaluation each
    file's search flagged as last, which is what the reference records in ``_scores`` (aligners.py:124-125).

    Every search step is ONE rasteriser call (every file's track at that file's current ratio, interval arithmetic on the
    device) and ONE solve over all files, on one plan held across the ~17 steps.  With ``start_seconds <= 0`` nothing is
    ever a bitmap (round 5): the subtitle tables are uploaded once (``TrackSet.to_device``), every step writes the tracks'
    BOUNDARY LISTS (``ffs_rasterize_batch_runs``), the references are converted to lists once (``ffs_runs_from_bits_batch``)
    and the solves take lists on both sides with host-known bounds -- no pass over any vector, no wait for the device
    besides the step's own scores.  Otherwise (or when a reference is too dense for a list) the step rasterises bits
    (``ffs_rasterize_batch_bits``) against the bit-packed references.  Multi-level (float) references take the per-file
    path (``_fit_gss_batch_per_file``).  ``stats``, if given, receives {"steps", "files", "plan_length", "lists"}."""
    from . import _native
    from .aligners import _Vec
    from .batch import TrackSet

    n = len(refs)
    ref_vecs = [_Vec(r) for r in refs]
    if n == 0 or not all(v.two_level and len(v) > 0 for v in ref_vecs) or any(len(t[0]) == 0 for t in subtitle_records):
        return _fit_gss_batch_per_file(refs, subtitle_records, max_offset_samples, sample_rate, start_seconds)
    torch = _native.require_gpu()
    # references: bit-packed, one buffer, uploaded once
    words = [None if v.raster is not None else (v.packed if v.packed is not None else np.packbits(v.bits, bitorder="little"))
             for v in ref_vecs]
    offs = np.zeros(n, dtype=np.int64)
    total = 0
    for i, wv in enumerate(words):
        if wv is not None:
            offs[i] = total
            total += (wv.size + 63) // 64 * 64
    host = np.zeros(max(total, 64), dtype=np.uint8)
    for wv, o in zip(words, offs):
        if wv is not None:
            host[o:o + wv.size] = wv
    ref_dev = torch.from_numpy(host).cuda()
    keep = [v.raster.packed_words() if v.raster is not None else None for v in ref_vecs]  # HBM-resident references, in place
    ref_ptr = np.array([k.data_ptr() if k is not None else ref_dev.data_ptr() + int(o) for k, o in zip(keep, offs)], dtype=np.uint64)
    ref_len = np.array([len(v) for v in ref_vecs], dtype=np.int64)
    tracks = TrackSet(list(subtitle_records))
    # one plan for the whole search: long enough for the longest candidate (ratio 1.1)
    longest = _native.raster_lengths(tracks.end_max, np.full(n, MAX_FRAMERATE_RATIO), sample_rate)
    n_fft = max(_native.plan_length(int(r), int(s), max_offset_samples) for r, s in zip(ref_len, longest))
    if n_fft > _native.MAX_FFT_LENGTH:
        raise ValueError("inputs too long for the device transform (N=%d > 2^24)" % n_fft)
    plan = _native.get_plan(n_fft, pairs_in_flight=min(max(n, 1), 512), max_cand=8)
    cand_out = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
    pair_out = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
    lo = np.zeros(2 * n, dtype=np.float64)
    hi = np.ones(2 * n, dtype=np.float64)
    lo[0::2] = [v.lo for v in ref_vecs]
    hi[0::2] = [v.hi for v in ref_vecs]
    ptrs = np.zeros(2 * n, dtype=np.uint64)
    lens = np.zeros(2 * n, dtype=np.int64)
    ptrs[0::2], lens[0::2] = ref_ptr, ref_len
    recorded = [None] * n
    which = np.arange(n)
    steps = [0]
    # boundary lists on both sides when the rasteriser may write them (start_seconds <= 0) and every reference fits one
    use_lists = start_seconds <= 0
    ref_lists = None
    bounds = np.zeros(2 * n, dtype=np.int32)
    if use_lists:
        cap = 32768
        block = (_native.runs_list_bytes(cap) + 63) // 64 * 64
        ref_lists = torch.empty(n * block, dtype=torch.uint8, device="cuda")
        list_ptr = ref_lists.data_ptr() + (np.arange(n, dtype=np.uint64) * np.uint64(block))
        _native.runs_from_bits_batch(ref_ptr, ref_len, list_ptr, np.full(n, cap, dtype=np.int64))
        n_ref = ref_lists.view(torch.int32).reshape(n, block // 4)[:, 0].cpu().numpy()  # (one small read-back per search)
        if (n_ref >= cap).any():
            use_lists, ref_lists = False, None
        else:
            tracks.to_device()
            ptrs[0::2] = list_ptr
            bounds[0::2] = np.maximum(n_ref, 2)
    dtypes = (_native.FFS_DTYPE_RUNS, _native.FFS_DTYPE_RUNS) if use_lists else _native.FFS_DTYPE_U1

    timers = stats.setdefault("timers_us", {"rasterize": 0.0, "solve": 0.0, "read_back": 0.0}) if stats is not None and stats.get("time_steps") else None
    import time as _time

    def evaluate(ratios, is_last):
        t0 = _time.perf_counter()
        if use_lists:
            data, c_offs, c_lens, c_bounds = tracks.rasterize_runs(which, ratios, sample_rate, start_seconds)
            bounds[1::2] = c_bounds
        else:
            data, c_offs, c_lens = tracks.rasterize(which, ratios, sample_rate, start_seconds)
        ptrs[1::2] = data.data_ptr() + c_offs.astype(np.uint64)
        lens[1::2] = c_lens
        hi[1::2] = np.minimum(1.0 / np.asarray(ratios, dtype=np.float64), 1.0)  # speech_transformers.py:977
        t1 = _time.perf_counter()
        plan.align_batch(n, 1, dtypes, ptrs, lens, lo, hi, max_offset_samples, None, cand_out, pair_out,
                         vec_max_boundaries=bounds if use_lists else None)
        t2 = _time.perf_counter()
        cres = cand_out.cpu().numpy().view(_native.CAND_RESULT_DTYPE)[:n]
        if timers is not None:
            t3 = _time.perf_counter()
            timers["rasterize"] += 1e6 * (t1 - t0)
            timers["solve"] += 1e6 * (t2 - t1)
            timers["read_back"] += 1e6 * (t3 - t2)
        steps[0] += 1
        if (cres["flags"] & _native.FLAG_AMBIGUOUS).any():  # degenerate input (e.g. a silent reference): the careful path
            raise _Ambiguous()
        if is_last:
            for i, ratio in enumerate(ratios):
                recorded[i] = ((np.float64(cres[i]["score"]), int(cres[i]["offset"])), float(ratio))
        return -cres["score"]

    try:
        gss_batch(evaluate, n)
    except _Ambiguous:
        return _fit_gss_batch_per_file(refs, subtitle_records, max_offset_samples, sample_rate, start_seconds)
    if stats is not None:
        stats.update({"steps": steps[0], "files": n, "plan_length": int(n_fft), "lists": bool(use_lists)})
    del ref_dev, keep, ref_lists
    return recorded


class _Ambiguous(Exception):
    pass


def _fit_gss_batch_per_file(refs, subtitle_records, max_offset_samples=None, sample_rate: int = 100, start_seconds: float = 0):
    """The same search with one rasterisation per file and step and the drop-in's ``solve_pairs`` (any element type,
    ambiguity handling): what ``fit_gss_batch`` falls back to for multi-level references and degenerate inputs."""
    from .aligners import _Vec, solve_pairs
    from .subtitle_raster import rasterize_candidates

    ref_vecs = [_Vec(r) for r in refs]
    recorded = [None] * len(refs)

    def evaluate(ratios, is_last):
        pairs = []
        for rv, (s, e, m), ratio in zip(ref_vecs, subtitle_records, ratios):
            raster = rasterize_candidates(s, e, m, [float(ratio)], sample_rate, start_seconds)[0]
            pairs.append((rv, [_Vec(raster)]))
        cres, _ = solve_pairs(pairs, max_offset_samples)
        if is_last:
            for i, ratio in enumerate(ratios):
                recorded[i] = ((np.float64(cres[i, 0]["score"]), int(cres[i, 0]["offset"])), float(ratio))
        return -cres[:, 0]["score"]

    gss_batch(evaluate, len(refs))
    return recorded
