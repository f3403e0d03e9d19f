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
                  max_offset_samples=None, sample_rate: int = 100, start_seconds: float = 0):
    """``MaxScoreAligner(FFTAligner(max_offset_samples)).fit_gss`` for many files at once.

    refs[i]: reference activity vector of file i (array or DeviceRaster); subtitle_records[i]:
    (start_us, end_us, is_metadata) of its subtitles.  Every step rasterises each file's track at
    that file's current ratio on the device and solves all files in one ``ffs_align_batch`` call.
    Returns a list of ((score, offset), ratio) -- the evaluation each file's search flagged as last,
    which is what the reference records in ``_scores`` (aligners.py:124-125)."""
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
