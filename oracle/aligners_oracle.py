"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference aligner.

Restates, function by function, what ``/root/reference/ffsubsync/aligners.py`` and
``/root/reference/ffsubsync/golden_section_search.py`` compute, in plain numpy (complex128
pocketfft, exactly the arithmetic the reference performs).  Pinned against outputs of the
unmodified reference by ``tests/golden/make_golden.py`` -> ``tests/golden/*.npz|json`` and
``tests/test_oracle_golden.py`` (parity pinned).

Citations are ``file:line`` into ``/root/reference/ffsubsync/``.
"""
import math

import numpy as np


class OracleAlignmentError(Exception):
    """Stands in for FailedToFindAlignmentException (aligners.py:20)."""


def as_pm1(x):
    """aligners.py:51-57 -- '0101' strings -> ints; then 2*float64(x) - 1."""
    if isinstance(x, str):
        x = [int(ch) for ch in x]
    return 2.0 * np.asarray(x).astype(float) - 1.0


def fft_length(n_ref, n_sub):
    """aligners.py:67-68 -- N = 2**ceil(log2(R+S)) evaluated with the same float ops."""
    return int(2 ** math.ceil(math.log(n_ref + n_sub, 2)))


def convolve_full(ref, sub):
    """aligners.py:50-74 -- the length-N 'convolve' array of FFTAligner.fit.

    convolve[k] = sum_i sub'[i] * ref'[i + d] with d = N-1-S-k.
    """
    r = as_pm1(ref)
    s = as_pm1(sub)
    if r.size == 0 or s.size == 0:  # aligners.py:58-66
        raise OracleAlignmentError(
            "cannot align empty speech data (reference length=%d, subtitle length=%d)" % (r.size, s.size)
        )
    n = fft_length(r.size, s.size)
    lead = n - r.size - s.size  # aligners.py:69 'extra_zeros'
    sub_padded = np.concatenate([np.zeros(lead + r.size), s])  # aligners.py:70
    ref_padded = np.concatenate([r, np.zeros(s.size + lead)])[::-1]  # aligners.py:71-73
    spec = np.fft.fft(sub_padded) * np.fft.fft(ref_padded)
    return np.real(np.fft.ifft(spec)), s.size  # aligners.py:74


def mask_extreme_offsets(conv, n_sub, max_offset_samples):
    """aligners.py:31-43 -- copy, then Python-slice-assign -inf outside the lag window."""
    out = np.array(conv, copy=True)
    if max_offset_samples is None:
        return out
    lo = len(out) - 1 + (-max_offset_samples) - n_sub
    hi = len(out) - 1 + max_offset_samples - n_sub
    out[:lo] = -np.inf  # Python negative-index semantics are part of the contract
    out[hi:] = -np.inf
    return out


def fft_align(ref, sub, max_offset_samples=None):
    """FFTAligner(max_offset_samples).fit(ref, sub, get_score=True).transform()
    (aligners.py:45-48, 50-86) -> (score: np.float64, offset: int)."""
    conv, n_sub = convolve_full(ref, sub)
    masked = mask_extreme_offsets(conv, n_sub, max_offset_samples)
    k = int(np.argmax(masked))  # first maximum == largest offset on ties
    return masked[k], len(masked) - 1 - k - n_sub


def max_score_align(ref, candidates, max_offset_samples=None):
    """MaxScoreAligner(FFTAligner, None, sr, max_s).fit_transform(ref, candidates)
    for raw-array candidates (aligners.py:131-167) -> ((score, offset), index).

    The per-candidate solve is repeated in full, as the reference does (aligners.py:136-151);
    candidates whose |offset| exceeds the limit are dropped (aligners.py:156-159) and the
    first candidate of maximal score wins (aligners.py:166).
    """
    scored = [(fft_align(ref, c, max_offset_samples), i) for i, c in enumerate(candidates)]
    if max_offset_samples is not None:
        scored = [s for s in scored if abs(s[0][1]) <= max_offset_samples]
    if not scored:
        raise OracleAlignmentError("Synchronization failed; consider passing --max-offset-seconds")
    best = scored[0]
    for item in scored[1:]:
        if item[0][0] > best[0][0]:
            best = item
    return best


INV_PHI = (math.sqrt(5.0) - 1.0) / 2.0
INV_PHI2 = (3.0 - math.sqrt(5.0)) / 2.0


def gss_trace(f, a, b, tol=1e-4):
    """golden_section_search.py:15-74 -- returns (interval, [(x, is_last) ...]) so tests can
    compare the exact evaluation sequence as well as the bracketing interval."""
    a, b = min(a, b), max(a, b)
    h = b - a
    trace = []
    if h <= tol:
        return (a, b), trace
    n = int(math.ceil(math.log(tol / h) / math.log(INV_PHI)))

    def ev(x, last):
        trace.append((x, last))
        return f(x, last)

    c = a + INV_PHI2 * h
    d = a + INV_PHI * h
    yc = ev(c, n == 1)
    yd = ev(d, n == 1)
    for k in range(n - 1):
        h = INV_PHI * h
        if yc < yd:  # strict (golden_section_search.py:56)
            b, d, yd = d, c, yc
            c = a + INV_PHI2 * h
            yc = ev(c, k == n - 2)
        else:
            a, c, yc = c, d, yd
            d = a + INV_PHI * h
            yd = ev(d, k == n - 2)
    return ((a, d) if yc < yd else (c, b)), trace


def speech_boundaries(frames):
    """speech_transformers.py:310-317 -- first/last frame index with value > 0.5, or None."""
    nz = np.nonzero(np.asarray(frames) > 0.5)[0]
    if nz.size == 0:
        return None, None
    return int(nz.min()), int(nz.max())
