"""CPU tests of the run-boundary correlation model (oracle/runs_model.py mirrors the index arithmetic of
csrc/ffs_runs.h): exact counts against a direct evaluation, and (score, offset) against the golden-pinned restatement of
the reference (oracle/aligners_oracle.py, aligners.py:31-80)."""
import numpy as np
import pytest

from oracle import aligners_oracle as orc
from oracle import runs_model as rm
from workloads import synth


def _direct(s, r, d_lo, d_hi):
    S, R = len(s), len(r)
    out = []
    for d in range(d_lo, d_hi + 1):
        i0, i1 = max(0, -d), min(S, R - d)
        if i1 <= i0:
            out.append((0, 0, 0, 0))
            continue
        a, b = s[i0:i1].astype(int), r[i0 + d:i1 + d].astype(int)
        out.append(((a & b).sum(), a.sum(), b.sum(), i1 - i0))
    return np.array(out).T


def _vector(rng, n):
    x = np.zeros(n, np.uint8)
    for _ in range(rng.randint(0, max(1, n // 4) + 1)):
        a = rng.randint(0, n)
        x[a:min(n, a + rng.randint(1, 12))] = 1
    mode = rng.rand()
    if mode < 0.1:
        x[:] = 1
    elif mode < 0.2:
        x[:] = 0
    return x


def test_counts_equal_direct_evaluation():
    rng = np.random.RandomState(1)
    for trial in range(400):
        S, R = rng.randint(1, 400), rng.randint(1, 400)
        if trial % 7 == 0:  # lengths that are multiples of the word size, runs that reach the end
            S, R = 32 * rng.randint(1, 8), 32 * rng.randint(1, 8)
        s, r = _vector(rng, S), _vector(rng, R)
        if trial % 5 == 0:
            s[-1] = r[-1] = 1
        d_lo = rng.randint(-S + 1, R)
        d_hi = rng.randint(d_lo, R)
        got = rm.window_counts(s, r, d_lo, d_hi)
        want = _direct(s, r, d_lo, d_hi)
        for g, w, name in zip(got, want, ("n11", "n1x", "nx1", "ov")):
            assert (g == w).all(), (trial, name, S, R, d_lo, d_hi)


def test_boundaries_and_ones_before():
    x = np.array([1, 1, 0, 0, 1, 0, 1, 1], np.uint8)
    q, cq = rm.boundaries(x)
    assert q.tolist() == [0, 2, 4, 5, 6, 8] and cq.tolist() == [0, 2, 2, 3, 3, 5]
    lb, ones = rm.ones_before(q, cq, 5, np.arange(-2, 11))
    assert ones.tolist() == [0, 0, 0, 1, 2, 2, 2, 3, 3, 4, 5, 5, 5]
    assert lb.tolist() == [0, 0, 0, 1, 1, 2, 2, 3, 4, 5, 5, 6, 6]


@pytest.mark.parametrize("max_offset", [6000, None, 150])
def test_best_lag_equals_reference_restatement(max_offset):
    """A ten-minute seven-ratio problem with the subtitle amplitudes min(1/ratio, 1) (speech_transformers.py:977): the
    model's (score, offset) against FFTAligner's, lag window as aligners.py:31-43 masks it."""
    sp = synth.make_pair_spec(3, duration_s=600)
    ref, cands = synth.pair_arrays(sp)
    for j, c in enumerate(cands):
        want_s, want_o = orc.fft_align(ref.astype(float), c.astype(float) * sp.cand_amp[j], max_offset)
        S, R = len(c), len(ref)
        n = orc.fft_length(R, S)
        if max_offset is None:
            d_lo, d_hi = -S + 1, R - 1  # lags with a non-empty overlap; the others are exactly 0 (< the maximum here)
        else:
            d_lo, d_hi = max(-max_offset + 1, -S + 1), min(max_offset, R - 1)  # k in [lo, hi) <=> d in (-mo, mo]
            assert n - 1 - max_offset - S >= 0
        got_s, got_o = rm.best_lag(c, ref, d_lo, d_hi, (0.0, sp.cand_amp[j]), (0.0, 1.0))
        assert got_o == want_o and got_s == pytest.approx(want_s, rel=1e-12)
