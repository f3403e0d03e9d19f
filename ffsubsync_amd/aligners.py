"""MI355X-native drop-in for ``ffsubsync.aligners`` (FFTAligner / MaxScoreAligner).

Same classes, constructor arguments, fitted attributes, return values and exceptions as the
reference (ffsubsync/aligners.py:20-167); the arithmetic -- +-1 map, zero-pad, three length-N
transforms, lag-window mask, argmax, max over candidates -- runs in ``libffsalign.so`` on the
GPU.  Install behind an unmodified ffsubsync with :func:`ffsubsync_amd.install`.
"""
import logging
import warnings
from typing import Any, List, Optional, Sequence, Tuple, Type, Union

import numpy as np

from . import _native
from .golden_section_search import gss
from .sklearn_shim import Pipeline, TransformerMixin  # the reference's own classes when ffsubsync is importable

logger: logging.Logger = logging.getLogger(__name__)

MIN_FRAMERATE_RATIO = 0.9  # aligners.py:16
MAX_FRAMERATE_RATIO = 1.1  # aligners.py:17


class FailedToFindAlignmentException(Exception):
    """aligners.py:20"""


def _as_array(x: Any) -> np.ndarray:
    """aligners.py:51-57 input handling: '0110' strings become ints; everything -> float64."""
    if isinstance(x, str):
        x = [int(ch) for ch in x]
    return np.asarray(x).astype(float).ravel()


class _Vec:
    """One activity vector as the native library wants it: two-level samples + (lo, hi) -- 0/1 bytes on
    the host, or a ``DeviceRaster`` already in HBM -- or arbitrary floats."""

    __slots__ = ("n", "two_level", "lo", "hi", "bits", "_values", "dev", "raster")

    def __init__(self, x: Any) -> None:
        self.dev = None
        self.raster = None
        self._values = None
        if hasattr(x, "bits") and hasattr(x, "lo") and hasattr(x, "hi") and hasattr(x.bits, "data_ptr"):
            # a DeviceRaster: nothing to convert or upload
            self.raster, self.n = x, len(x)
            self.two_level, self.lo, self.hi, self.bits = True, float(x.lo), float(x.hi), None
            return
        values = _as_array(x)
        self._values = values
        self.n = values.size
        if self.n == 0:
            self.two_level, self.lo, self.hi, self.bits = True, 0.0, 1.0, np.zeros(0, np.uint8)
            return
        lo, hi = float(values.min()), float(values.max())
        is_hi = values == hi
        self.two_level = bool(np.all(is_hi | (values == lo))) and np.isfinite(lo) and np.isfinite(hi)
        self.lo, self.hi = lo, hi
        self.bits = is_hi.astype(np.uint8) if (self.two_level and hi != lo) else np.zeros(self.n, np.uint8)

    def __len__(self) -> int:
        return self.n

    def host_values(self) -> np.ndarray:
        if self._values is None:
            self._values = np.asarray(self.raster, dtype=float)
        return self._values


def solve_pairs(pairs: Sequence[Tuple[_Vec, List[_Vec]]], max_offset_samples: Optional[int],
                filter_max_offset: Optional[int] = None, full_length: bool = False):
    """Solve a list of (reference, [candidates]) problems, all with the same candidate count, in one
    native batch.  Returns (cand_results, pair_results) as numpy structured arrays."""
    torch = _native.require_gpu()
    n_pairs = len(pairs)
    n_cand = len(pairs[0][1])
    vecs: List[_Vec] = []
    for ref, subs in pairs:
        if len(subs) != n_cand:
            raise ValueError("all pairs in one batch need the same number of candidates")
        vecs.append(ref)
        vecs.extend(subs)
    n_fft = 2
    for ref, subs in pairs:
        for s in subs:
            if len(ref) == 0 or len(s) == 0:
                # aligners.py:58-66
                raise FailedToFindAlignmentException(
                    "cannot align empty speech data "
                    "(reference length=%d, subtitle length=%d); "
                    "the reference or subtitles may contain no detectable speech" % (len(ref), len(s))
                )
            # full_length: always the reference's own N; otherwise the (possibly shorter) alias-free
            # length for the lag window -- results are identical either way
            n_fft = max(n_fft, _native.fft_length(len(ref), len(s)) if full_length
                        else _native.plan_length(len(ref), len(s), max_offset_samples))
    if n_fft > _native.MAX_FFT_LENGTH:
        raise ValueError("inputs too long for the device transform (N=%d > 2^24)" % n_fft)
    all_two_level = all(v.two_level for v in vecs)
    lens = np.array([len(v) for v in vecs], dtype=np.int64)
    keep_alive = []  # device tensors the descriptors point into
    if all_two_level:
        # bit-packed (FFS_DTYPE_U1): host vectors are packed here (an eighth of the PCIe bytes), rasters
        # that live in HBM as bytes are packed on the device, bit-packed rasters are used in place
        dtype = _native.FFS_DTYPE_U1
        chunks = [None if v.raster is not None else np.packbits(v.bits, bitorder="little") for v in vecs]
        for v in vecs:
            if v.raster is not None:
                v.dev = v.raster.packed_words()
                keep_alive.append(v.dev)
    else:
        # float inputs (fused / weighted VAD levels) go over as float64: the transforms nominate in fp32, the
        # winning lags are re-evaluated in fp64 from these very samples (no input rounding)
        dtype = _native.FFS_DTYPE_F64
        chunks = [np.ascontiguousarray(v.host_values(), dtype=np.float64).view(np.uint8) for v in vecs]
    # one H2D copy: host vectors packed back to back at 64-byte aligned offsets
    offs = np.zeros(len(chunks), dtype=np.int64)
    total = 0
    for i, c in enumerate(chunks):
        if c is None:
            continue
        offs[i] = total
        total += (c.size + 63) // 64 * 64
    host = np.zeros(max(total, 64), dtype=np.uint8)
    for c, o in zip(chunks, offs):
        if c is not None:
            host[o:o + c.size] = c
    dev = torch.from_numpy(host).cuda()
    ptrs = np.array([v.dev.data_ptr() if c is None else dev.data_ptr() + int(o)
                     for v, c, o in zip(vecs, chunks, offs)], dtype=np.uint64)
    lo = np.array([v.lo for v in vecs], dtype=np.float64)
    hi = np.array([v.hi for v in vecs], dtype=np.float64)
    plan = _native.get_plan(n_fft, pairs_in_flight=1 if n_pairs == 1 else 2, max_cand=max(8, n_cand))
    cand_out = torch.empty(n_pairs * n_cand * 24, dtype=torch.uint8, device=dev.device)
    pair_out = torch.empty(n_pairs * 24, dtype=torch.uint8, device=dev.device)
    plan.align_batch(n_pairs, n_cand, dtype, ptrs, lens, lo, hi, max_offset_samples, filter_max_offset,
                     cand_out, pair_out)
    cres = cand_out.cpu().numpy().view(_native.CAND_RESULT_DTYPE).reshape(n_pairs, n_cand).copy()
    pres = pair_out.cpu().numpy().view(_native.PAIR_RESULT_DTYPE).copy()
    del dev, keep_alive
    # A candidate with more tied maxima than its share of the exhaustive pool keeps FFS_FLAG_AMBIGUOUS.  Its
    # quota depends on how many candidates of the call overflowed, so solve such a pair again on its own
    # (the whole pool to itself); if it is still ambiguous the answer is the best of a truncated list: say so.
    amb = np.nonzero((cres["flags"] & _native.FLAG_AMBIGUOUS).any(axis=1))[0]
    if amb.size and n_pairs > 1:
        for p in amb:
            c1, p1 = solve_pairs([pairs[int(p)]], max_offset_samples, filter_max_offset, full_length)
            cres[p], pres[p] = c1[0], p1[0]
    elif amb.size:
        warnings.warn("alignment has more exactly tied best offsets than the device can enumerate "
                      "(degenerate input such as a silent reference); the reported offset is one of them",
                      RuntimeWarning)
    return cres, pres


class FFTAligner(TransformerMixin):
    """aligners.py:24-86.  ``fit(refstring, substring, get_score=False)`` finds the offset (in
    samples) by which ``substring`` must be shifted to best match ``refstring``."""

    def __init__(self, max_offset_samples: Optional[int] = None) -> None:
        self.max_offset_samples: Optional[int] = max_offset_samples
        self.best_offset_: Optional[int] = None
        self.best_score_: Optional[float] = None
        self.get_score_: bool = False

    def _solve_many(self, refstring: Any, substrings: Sequence[Any]) -> List[Tuple[float, int]]:
        """All candidates against one reference in a single device batch; leaves the fitted
        attributes as the reference's sequential loop would (those of the last candidate)."""
        ref = _Vec(refstring)
        subs = [_Vec(s) for s in substrings]
        cres, _ = solve_pairs([(ref, subs)], self.max_offset_samples)
        out = [(np.float64(r["score"]), int(r["offset"])) for r in cres[0]]
        self.best_score_, self.best_offset_ = out[-1]
        return out

    def fit(self, refstring, substring, get_score: bool = False) -> "FFTAligner":
        self._solve_many(refstring, [substring])
        self.get_score_ = get_score
        return self

    def transform(self, *_) -> Union[int, Tuple[float, int]]:
        if self.get_score_:
            return self.best_score_, self.best_offset_
        return self.best_offset_


class MaxScoreAligner(TransformerMixin):
    """aligners.py:89-167.  Runs the base aligner over candidate substrings / pipelines (one per
    framerate ratio) and keeps the best-scoring one."""

    def __init__(
        self,
        base_aligner: Union[FFTAligner, Type[FFTAligner]],
        srtin: Optional[str] = None,
        sample_rate=None,
        max_offset_seconds=None,
    ) -> None:
        self.srtin: Optional[str] = srtin
        if sample_rate is None or max_offset_seconds is None:
            self.max_offset_samples: Optional[int] = None
        else:
            self.max_offset_samples = abs(int(max_offset_seconds * sample_rate))
        if isinstance(base_aligner, type):
            self.base_aligner: FFTAligner = base_aligner(max_offset_samples=self.max_offset_samples)
        else:
            self.base_aligner = base_aligner
        self.max_offset_seconds: Optional[int] = max_offset_seconds
        self._scores: List[Tuple[Tuple[float, int], Pipeline]] = []

    def fit_gss(self, refstring, subpipe_maker):
        """aligners.py:111-129 -- golden-section search over the framerate ratio; only the final
        evaluation is recorded."""

        def opt_func(framerate_ratio, is_last_iter):
            subpipe = subpipe_maker(framerate_ratio)
            substring = subpipe.fit_transform(self.srtin)
            score = self.base_aligner.fit_transform(refstring, substring, get_score=True)
            logger.info("got score %.0f (offset %d) for ratio %.3f", score[0], score[1], framerate_ratio)
            if is_last_iter:
                self._scores.append((score, subpipe))
            return -score[0]

        gss(opt_func, MIN_FRAMERATE_RATIO, MAX_FRAMERATE_RATIO)
        return self

    def fit(self, refstring, subpipes: Union[Pipeline, List[Pipeline]]) -> "MaxScoreAligner":
        if not isinstance(subpipes, list):
            subpipes = [subpipes]
        batched = hasattr(self.base_aligner, "_solve_many")
        run: List[Tuple[Any, Any]] = []  # consecutive non-callable candidates -> one device batch

        def flush():
            if not run:
                return
            if batched:
                results = self.base_aligner._solve_many(refstring, [s for _, s in run])
                self.base_aligner.get_score_ = True
            else:
                results = [self.base_aligner.fit_transform(refstring, s, get_score=True) for _, s in run]
            for (pipe, _), res in zip(run, results):
                self._scores.append((res, pipe))
            run.clear()

        for subpipe in subpipes:
            if callable(subpipe):
                flush()
                self.fit_gss(refstring, subpipe)
                continue
            elif hasattr(subpipe, "transform"):
                substring = subpipe.transform(self.srtin)
            else:
                substring = subpipe
            run.append((subpipe, substring))
        flush()
        return self

    def transform(self, *_) -> Tuple[Tuple[float, float], Pipeline]:
        scores = self._scores
        if self.max_offset_samples is not None:
            scores = [s for s in scores if abs(s[0][1]) <= self.max_offset_samples]
        if len(scores) == 0:
            raise FailedToFindAlignmentException(
                "Synchronization failed; consider passing "
                "--max-offset-seconds with a number larger than "
                "{}".format(self.max_offset_seconds)
            )
        (score, offset), subpipe = max(scores, key=lambda x: x[0][0])
        return (score, offset), subpipe
