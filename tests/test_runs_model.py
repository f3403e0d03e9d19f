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


def test_list_from_intervals_equals_the_painted_vector():
    """k_rasterize_runs' rule (sorted starts, a run begins beyond every earlier end, touching intervals merge, ones in
    front = ends minus starts of the runs in front) against painting the intervals into a vector; list_bits32 and
    bits_from_list against the same vector."""
    rng = np.random.RandomState(11)
    for trial in range(300):
        n = int(rng.randint(1, 500))
        k = int(rng.randint(0, 40))
        a = rng.randint(0, n + 1, k)
        b = np.minimum(n, a + rng.randint(-3, 30, k))
        if trial % 4 == 0 and k > 2:  # touching chains and duplicates
            a[1:] = b[:-1]
            b = np.minimum(n, a + rng.randint(0, 9, k))
        x = np.zeros(n, np.uint8)
        for s_, e_ in zip(a, b):
            if s_ < e_:
                x[s_:e_] = 1
        pos, ones_before, ones = rm.list_from_intervals(a, b, n)
        wq, wc = rm.boundaries(x)
        assert np.array_equal(pos, wq) and np.array_equal(ones_before, wc) and ones == int(x.sum()), trial
        assert np.array_equal(rm.bits_from_list(pos, n), x)
        for start in (-40, -31, -1, 0, 5, n - 33, n - 1, n + 7):
            want = 0
            for i in range(32):
                if 0 <= start + i < n and x[start + i]:
                    want |= 1 << i
            assert rm.list_bits32(pos, start) == want, (trial, start)


def test_list_rasteriser_model_equals_the_reference_rasters():
    """The same rule on the reference's own rasters (tests/golden/raster_golden.npz: SubtitleScaler +
    SubtitleSpeechTransformer of the unmodified reference): intervals from the golden-pinned raster oracle -> list ->
    bits == the committed raster."""
    import os

    from oracle import raster_oracle as ro

    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raster_golden.npz"))
    ratios = [float(r) for r in gold["ratios"]]
    for name in ("a", "b"):
        s, e, m = gold[name + "_start_us"], gold[name + "_end_us"], gold[name + "_meta"]
        for j, r in enumerate(ratios):
            want = (gold["%s_r%d" % (name, j)] != 0).astype(np.uint8)
            iv = ro.intervals(s, e, m, r, 100, 0, want.size)
            pos, ones_before, ones = rm.list_from_intervals(iv[:, 0], iv[:, 1], want.size)
            assert np.array_equal(rm.bits_from_list(pos, want.size), want), (name, j)
            assert ones == int(want.sum())


def test_fp32_prefilter_never_drops_the_maximum():
    """The margin of k_runs_corr's fp32 prefilter (edge term refreshed every four lags): over random short problems and
    the production-size window of a 2 h pair, every lag whose exact score equals the maximum survives."""
    rng = np.random.RandomState(5)
    cases = []
    for _ in range(60):
        S, R = int(rng.randint(40, 400)), int(rng.randint(40, 400))
        cases.append((_vector(rng, S), _vector(rng, R), float(rng.choice([1.0, 0.96, 0.959]))))
    sp = synth.make_pair_spec(1, duration_s=7200)
    ref, cands = synth.pair_arrays(sp)
    cases += [(cands[j], ref, float(sp.cand_amp[j])) for j in (0, 2)]
    for s, r, amp in cases:
        S, R = len(s), len(r)
        d_lo, d_hi = max(-S + 1, -5999), min(R - 1, 6000)
        n11, n1x, nx1, ov = rm.window_counts(s, r, d_lo, d_hi) if S < 1000 else _fast_counts(s, r, d_lo, d_hi)
        keep, margin, exact = rm.prefilter_keeps_the_maximum(n11, n1x, nx1, ov, d_lo, -1.0, 2.0 * amp - 1.0, -1.0, 1.0, R, S)
        assert keep[exact == exact.max()].all()
        assert keep.sum() <= max(64, exact.size // 8) or S < 1000  # (and it does filter: a few per cent of a real window)


def _fast_counts(s, r, d_lo, d_hi):
    """Direct counts for long vectors through cumulative sums and an FFT-free sliding product (numpy correlate on a
    window of lags would be O(N W); the run model itself is exact but slow in pure Python at 2 h)."""
    S, R = len(s), len(r)
    P, CP = rm.boundaries(s)
    Q, CQ = rm.boundaries(r)
    d = np.arange(d_lo, d_hi + 1, dtype=np.int64)
    i0, i1 = np.maximum(0, -d), np.minimum(S, R - d)
    ov = np.maximum(0, i1 - i0)
    _, b1 = rm.ones_before(P, CP, int(s.sum()), np.clip(i1, 0, S))
    _, b0 = rm.ones_before(P, CP, int(s.sum()), np.clip(i0, 0, S))
    _, r1 = rm.ones_before(Q, CQ, int(r.sum()), np.clip(i1 + d, 0, R))
    _, r0 = rm.ones_before(Q, CQ, int(r.sum()), np.clip(i0 + d, 0, R))
    # n11(d) = sum over runs of s of ones of r in [a + d, e + d)
    a, e = P[0::2], P[1::2]
    n11 = np.zeros(d.size, dtype=np.int64)
    for k in range(a.size):
        _, hi_ = rm.ones_before(Q, CQ, int(r.sum()), np.clip(e[k] + d, 0, R))
        _, lo_ = rm.ones_before(Q, CQ, int(r.sum()), np.clip(a[k] + d, 0, R))
        n11 += hi_ - lo_
    return n11, np.where(ov > 0, b1 - b0, 0), np.where(ov > 0, r1 - r0, 0), ov


def test_one_scan_identity_of_the_round6_kernel():
    """k_runs_corr (round 6) gets g and n11 at every thread's first lag from ONE block scan: with hs = the sum of the
    thread's LPT second differences, ws = sum_k (LPT - k) h[k] and the thread index t,
        g_c = g_0 - A,    n11_c = n11_0 + LPT t g_0 - LPT ((t - 1) A - B) - C,
    A / B / C the exclusive prefixes of hs / t hs / ws -- against the two running sums it replaces, in the kernel's own
    wrap-around 32-bit arithmetic."""
    rng = np.random.RandomState(11)
    for trial in range(20):
        lpt, threads = 24, int(rng.choice([1, 7, 512]))
        h = rng.randint(-40, 41, size=lpt * threads).astype(np.int64)
        if trial % 4 == 0:
            h = rng.randint(-32768, 32768, size=lpt * threads).astype(np.int64)  # what 16-bit cells can hold
        g0, n0 = int(rng.randint(-3000, 3000)), int(rng.randint(0, 700000))
        # reference: n11(d + 1) = n11(d) + g(d) - h(d), g(d + 1) = g(d) - h(d)
        g = g0 - np.concatenate([[0], np.cumsum(h)[:-1]])
        n11 = n0 + np.concatenate([[0], np.cumsum(g - h)[:-1]])
        hb = h.reshape(threads, lpt)
        hs = hb.sum(axis=1)
        ws = (hb * (lpt - np.arange(lpt))).sum(axis=1)
        t = np.arange(threads)
        excl = lambda v: np.concatenate([[0], np.cumsum(v)[:-1]])
        A, B, C = excl(hs), excl(t * hs), excl(ws)
        wrap = lambda v: ((np.asarray(v, dtype=np.int64) + 2 ** 31) % 2 ** 32) - 2 ** 31
        g_c = wrap(g0 - A)
        n11_c = wrap(n0 + lpt * t * g0 - lpt * ((t - 1) * A - B) - C)
        assert np.array_equal(g_c, wrap(g[::lpt])), trial
        assert np.array_equal(n11_c, wrap(n11[::lpt])), trial


def test_run_per_lane_walk_adds_the_boundary_coincidences():
    """Round 6 walks one candidate RUN per lane: per reference run read, + at (start, start) and (end, end), - at
    (end of the reference run, start) and (start of the reference run, end).  Same second-difference histogram as the sum
    over all boundary pairs (p, q) of db[p] drho[q] at lag q - p (the round-5 walk), lag window and all."""
    rng = np.random.RandomState(3)
    for trial in range(30):
        R, S = int(rng.randint(200, 3000)), int(rng.randint(200, 3000))

        def runs(n):
            cuts = np.unique(rng.randint(0, n + 1, size=2 * int(rng.randint(1, 40))))
            cuts = cuts[: cuts.size // 2 * 2]
            return cuts[0::2], cuts[1::2]  # starts, ends (end = one past the last one; may equal n)

        ps, pe = runs(S)
        qs, qe = runs(R)
        d_lo, d_hi = int(rng.randint(-S, 0)), int(rng.randint(0, R))
        width = d_hi - d_lo  # h is needed for the lags d_lo .. d_hi - 1
        by_boundaries = np.zeros(width, dtype=np.int64)
        P = np.concatenate([ps, pe]); sp = np.concatenate([np.ones(ps.size), -np.ones(pe.size)])
        Q = np.concatenate([qs, qe]); sq = np.concatenate([np.ones(qs.size), -np.ones(qe.size)])
        for p, a in zip(P, sp):
            for q, b in zip(Q, sq):
                d = q - p - d_lo
                if 0 <= d < width:
                    by_boundaries[d] += int(a * b)
        by_runs = np.zeros(width, dtype=np.int64)
        for s_, e_ in zip(ps, pe):           # a lane
            for a_, b_ in zip(qs, qe):       # the reference runs it reads
                for lag, sign in ((a_ - s_, 1), (b_ - s_, -1), (a_ - e_, -1), (b_ - e_, 1)):
                    d = lag - d_lo
                    if 0 <= d < width:
                        by_runs[d] += sign
        assert np.array_equal(by_runs, by_boundaries), trial
