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
from .sklearn_shim import Pipeline, TransformerMixin, reference_module  # the reference's own classes when importable

logger: logging.Logger = logging.getLogger(__name__)

MIN_FRAMERATE_RATIO = 0.9  # aligners.py:16
MAX_FRAMERATE_RATIO = 1.1  # aligners.py:17


_ref_aligners = reference_module("aligners")

if _ref_aligners is not None:
    FailedToFindAlignmentException = _ref_aligners.FailedToFindAlignmentException  # `except` clauses in the caller
else:

    class FailedToFindAlignmentException(Exception):
        """aligners.py:20"""


def _as_array(x: Any) -> np.ndarray:
    """aligners.py:51-57 input handling: '0110' strings become ints; everything -> float64."""
    if isinstance(x, str):
        x = [int(ch) for ch in x]
    return np.ascontiguousarray(x, dtype=np.float64).ravel()  # (no copy of an array that already is float64)


class _Vec:
    """One activity vector as the native library wants it: two-level samples + (lo, hi) -- 0/1 bytes on
    the host, or a ``DeviceRaster`` already in HBM -- or arbitrary floats."""

    __slots__ = ("n", "two_level", "lo", "hi", "bits", "packed", "_values", "dev", "raster")

    def __init__(self, x: Any) -> None:
        self.dev = None
        self.raster = None
        self._values = None
        self.packed = None  # the samples as little-endian bits (host vectors whose levels the library found)
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
        found = _native.two_level_pack(values)  # two passes in C instead of five numpy temporaries
        if found is not None:
            self.two_level, self.bits = True, None
            self.lo, self.hi, self.packed = found
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


_pack_pool = None


def _vecs(arrays: Sequence[Any]) -> List[_Vec]:
    """``[_Vec(x) for x in arrays]`` with the host passes (level detection + bit packing of float64 vectors, 5.8 MB each
    for 2 h) spread over a small thread pool: ``ffs_two_level_pack`` runs outside the GIL and is memory-bound, so eight
    2 h vectors take about the time of two."""
    global _pack_pool
    big = [i for i, x in enumerate(arrays) if not hasattr(x, "bits") and not isinstance(x, str) and np.size(x) >= 1 << 16]
    if len(big) < 2:
        return [_Vec(x) for x in arrays]
    if _pack_pool is None:
        import os
        from concurrent.futures import ThreadPoolExecutor

        _pack_pool = ThreadPoolExecutor(max_workers=max(2, min(8, (os.cpu_count() or 2) // 2)), thread_name_prefix="ffs-pack")
    out: List[Optional[_Vec]] = [None] * len(arrays)
    futs = {i: _pack_pool.submit(_Vec, arrays[i]) for i in big}
    for i, x in enumerate(arrays):
        if i not in futs:
            out[i] = _Vec(x)
    for i, f in futs.items():
        out[i] = f.result()
    return out  # type: ignore[return-value]


def solve_host_batch(problems: Sequence[Tuple[Any, Sequence[Any]]], max_offset_samples: Optional[int],
                     filter_max_offset: Optional[int] = None):
    """Many files' numpy vectors at once -- ``problems`` = [(reference, [candidates...]), ...], every problem with the same
    number of candidates, the arrays exactly what the reference's pipelines hand ``MaxScoreAligner.fit`` (ffsubsync.py:230-235):
    threaded level detection + bit packing, ONE host-to-device copy, ONE ``ffs_align_batch``.  Returns
    (cand_results, pair_results) like ``solve_pairs``; pair_results[i]["best_cand"] indexes problem i's candidate list."""
    flat: List[Any] = []
    for ref, cands in problems:
        flat.append(ref)
        flat.extend(cands)
    vecs = _vecs(flat)
    pairs, k = [], 0
    for ref, cands in problems:
        pairs.append((vecs[k], vecs[k + 1: k + 1 + len(cands)]))
        k += 1 + len(cands)
    return solve_pairs(pairs, max_offset_samples, filter_max_offset)


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
    # One element type per ROLE (ffs_align_batch_typed): references and candidates are judged separately, so the
    # two-level subtitle rasters stay bit-packed (an eighth of a byte per sample over PCIe and in HBM) when only the
    # reference is float-valued -- e.g. the {0, .4, .6, 1} output of the weighted fused VAD, speech_transformers.py:290-293.
    stride = 1 + n_cand
    role_two_level = [all(v.two_level for v in vecs[0::stride]),
                      all(v.two_level for i, v in enumerate(vecs) if i % stride)]
    lens = np.array([len(v) for v in vecs], dtype=np.int64)
    # A role whose vectors all carry their boundary list (DeviceRaster.runs: straight from the subtitle intervals, or
    # extracted once when the reference vector was uploaded) goes over as lists (FFS_DTYPE_RUNS) with the lists'
    # host-known length bounds: no pass over the bits, and the call does not wait for the device before it returns.
    role_lists = [all(v.raster is not None and getattr(v.raster, "runs", None) is not None for v in vecs[0::stride]),
                  all(v.raster is not None and getattr(v.raster, "runs", None) is not None
                      for i, v in enumerate(vecs) if i % stride)]
    bounds = np.zeros(len(vecs), dtype=np.int32)
    keep_alive = []  # device tensors the descriptors point into
    chunks = []
    for i, v in enumerate(vecs):
        if role_lists[1 if i % stride else 0]:
            v.dev = v.raster.runs
            bounds[i] = v.raster.runs_bound
            keep_alive.append(v.dev)
            chunks.append(None)
        elif role_two_level[1 if i % stride else 0]:
            # bit-packed (FFS_DTYPE_U1): host vectors are packed here, rasters that live in HBM as bytes are packed on
            # the device, bit-packed rasters are used in place
            if v.raster is not None:
                v.dev = v.raster.packed_words()
                keep_alive.append(v.dev)
                chunks.append(None)
            else:
                chunks.append(v.packed if v.packed is not None else np.packbits(v.bits, bitorder="little"))
        else:
            # float inputs (fused / weighted VAD levels) go over as float64: the transforms nominate in fp32, the
            # winning lags are re-evaluated in fp64 from these very samples (no input rounding)
            chunks.append(np.ascontiguousarray(v.host_values(), dtype=np.float64).view(np.uint8))
    role_dtype = [_native.FFS_DTYPE_RUNS if l else (_native.FFS_DTYPE_U1 if t else _native.FFS_DTYPE_F64)
                  for l, t in zip(role_lists, role_two_level)]
    dtype = role_dtype[0] if role_dtype[0] == role_dtype[1] and not any(role_lists) else (role_dtype[0], role_dtype[1])
    # one H2D copy: host vectors packed back to back at 64-byte aligned offsets
    offs = np.zeros(len(chunks), dtype=np.int64)
    total = 0
    for i, c in enumerate(chunks):
        if c is None:
            continue
        offs[i] = total
        total += (c.size + 63) // 64 * 64
    dev = None
    if total:  # (nothing to upload when every vector already lives in HBM)
        host = np.zeros(total, dtype=np.uint8)
        for c, o in zip(chunks, offs):
            if c is not None:
                host[o:o + c.size] = c
        dev = torch.from_numpy(host).cuda()
    ptrs = np.array([v.dev.data_ptr() if c is None else dev.data_ptr() + int(o)
                     for v, c, o in zip(vecs, chunks, offs)], dtype=np.uint64)
    lo = np.array([v.lo for v in vecs], dtype=np.float64)
    hi = np.array([v.hi for v in vecs], dtype=np.float64)
    plan = _native.get_plan(n_fft, pairs_in_flight=1 if n_pairs == 1 else 2, max_cand=max(8, n_cand))
    n_cbytes = n_pairs * n_cand * 24
    results = torch.empty(n_cbytes + n_pairs * 24, dtype=torch.uint8, device="cuda")  # both record arrays: one copy back
    cand_out, pair_out = results[:n_cbytes], results[n_cbytes:]
    plan.align_batch(n_pairs, n_cand, dtype, ptrs, lens, lo, hi, max_offset_samples, filter_max_offset,
                     cand_out, pair_out, vec_max_boundaries=bounds if any(role_lists) else None)
    raw = results.cpu().numpy()
    cres = raw[:n_cbytes].view(_native.CAND_RESULT_DTYPE).reshape(n_pairs, n_cand).copy()
    pres = raw[n_cbytes:].view(_native.PAIR_RESULT_DTYPE).copy()
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
        vecs = _vecs([refstring] + list(substrings))
        ref, subs = vecs[0], vecs[1:]
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
    """aligners.py:89-167: the constructor contract of SURVEY 8b, ``fit`` re-done for the device -- all candidate
    substrings / pipelines of a call (one per framerate ratio) go through ONE batched solve --, the golden-section wrapper
    and the pick of the best candidate.  ONE implementation whether or not ffsubsync is importable (the same code runs in
    the CPU tests against the unmodified ``try_sync``, tests/test_install_real_seam.py, and on the GPU box, where
    ffsubsync is absent); only ``TransformerMixin`` and the exception class are the reference's own when it is there."""

    def __init__(self, base_aligner, srtin: Optional[str] = None, sample_rate=None, max_offset_seconds=None) -> None:
        have_window = sample_rate is not None and max_offset_seconds is not None
        self.max_offset_samples: Optional[int] = abs(int(max_offset_seconds * sample_rate)) if have_window else None
        # a class is instantiated with the window; an instance keeps whatever window it was built with (:103-108)
        self.base_aligner = (base_aligner(max_offset_samples=self.max_offset_samples)
                             if isinstance(base_aligner, type) else base_aligner)
        self.srtin, self.max_offset_seconds = srtin, max_offset_seconds
        self._scores: List[Tuple[Tuple[float, int], Pipeline]] = []  # append-only across fits (:109)

    def fit_gss(self, refstring, subpipe_maker):
        def negated_score(ratio, is_last_iter):
            pipe = subpipe_maker(ratio)
            result = self.base_aligner.fit_transform(refstring, pipe.fit_transform(self.srtin), get_score=True)
            logger.info("got score %.0f (offset %d) for ratio %.3f", result[0], result[1], ratio)
            if is_last_iter:  # only the search's final evaluation is a candidate (:124-125)
                self._scores.append((result, pipe))
            return -result[0]

        gss(negated_score, MIN_FRAMERATE_RATIO, MAX_FRAMERATE_RATIO)
        return self

    def transform(self, *_):
        window = self.max_offset_samples
        kept = [entry for entry in self._scores if window is None or abs(entry[0][1]) <= window]
        if not kept:
            raise FailedToFindAlignmentException(
                "Synchronization failed; consider passing --max-offset-seconds with a number larger than %s"
                % (self.max_offset_seconds,))
        best = max(kept, key=lambda entry: entry[0][0])  # first of equal scores, like the reference's max()
        return best[0], best[1]

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
