"""TEST INFRASTRUCTURE ONLY -- numpy model of the run-boundary correlation (the device's `k_runs_*` kernels).

What is modelled.  For two 0/1 activity vectors b (candidate, length S) and rho (reference, length R) the count
``n11(d) = sum_i b[i] * rho[i+d]`` -- the one data-dependent term of the reference's correlation
``convolve[k]``, aligners.py:70-74, for two-level inputs (c(d) is an affine function of n11, of the two
one-sided counts n1x / nx1 and of the overlap length) -- is piecewise linear in the lag d: with the run boundaries
``db[p] = b[p] - b[p-1]`` (+1 where a run of ones starts, -1 one past its end) and ``drho[q]`` likewise,

    g(d)  = n11(d) - n11(d-1) = sum_q drho[q] * b[q-d]
    h(d)  = g(d) - g(d+1)     = sum over boundary pairs (p, q) with q - p = d of db[p] * drho[q]

so every lag of a window [D0, D1] follows from n11(D0), g(D0) and the sparse second difference h by two running sums:
g(d+1) = g(d) - h(d), n11(d+1) = n11(d) + g(d+1).  All integers: the result is the exact correlation, no transform.

The functions below follow the device code's steps (boundary lists with cumulative ones, lower-bound search that also
yields the ones-count in front of a position, packed 16-bit accumulation, two scans, incremental one-sided counts) so
that the index arithmetic is checked on the CPU against a direct evaluation (tests/test_runs_model.py).
"""
import numpy as np

QSENT = 0x3FFFFFFF  # the staged lists' sentinel: beyond every position (vectors are shorter than 2^30 samples)


def boundaries(bits):
    """Sorted boundary positions Q of a 0/1 vector (even entries: run starts, odd entries: one past run ends) and
    CQ[k] = number of ones in front of position Q[k].  (k_runs_extract)"""
    x = np.asarray(bits).astype(np.int8)
    ext = np.concatenate([[0], x, [0]])
    q = np.flatnonzero(ext[1:] != ext[:-1]).astype(np.int64)  # position p: x[p] != x[p-1], p in [0, len]
    cum = np.concatenate([[0], np.cumsum(x, dtype=np.int64)])
    return q, cum[q]


def ones_before(q, cq, total, x):
    """(lb, number of ones of the vector in [0, x)) for positions x (any integers): lb = first k with Q[k] >= x.
    lb odd: x lies inside a run (a < x <= e), lb even: in a gap."""
    x = np.asarray(x, dtype=np.int64)
    lb = np.searchsorted(q, x, side="left")
    qe = np.concatenate([q, [np.iinfo(np.int64).max]])
    ce = np.concatenate([cq, [total]])
    inside = (lb & 1) == 1
    return lb, np.where(inside, ce[lb] - (qe[lb] - x), ce[lb])


def window_counts(sub_bits, ref_bits, d_lo, d_hi):
    """n11(d), n1x(d), nx1(d), overlap(d) for every lag d in [d_lo, d_hi] (inclusive) from the run boundaries alone."""
    sub_bits = np.asarray(sub_bits).astype(np.uint8)
    ref_bits = np.asarray(ref_bits).astype(np.uint8)
    S, R = sub_bits.size, ref_bits.size
    P, CP = boundaries(sub_bits)
    Q, CQ = boundaries(ref_bits)
    tot_s, tot_r = int(sub_bits.sum()), int(ref_bits.sum())
    W = d_hi - d_lo + 1
    sign_p = np.where(np.arange(P.size) & 1, -1, 1)
    # one search per candidate boundary: lower bound in Q of p + d_lo, ones of rho in front of it
    lb, ones = ones_before(Q, CQ, tot_r, P + d_lo)
    n11_0 = int(-(sign_p * ones).sum())            # n11(d_lo) = -sum_p db[p] * Rcum(p + d_lo)
    g_0 = int(-(sign_p * (lb & 1)).sum())          # g(d_lo)   = -sum_p db[p] * rho[p + d_lo - 1]
    # h(d), d in [d_lo, d_hi - 1], accumulated 16 bits per lag in 32-bit words exactly as the kernel's ds_add_u32 does
    # (round 5: the walk takes the reference's boundaries two at a time -- one run: start, end -- from the even entry at
    # or in front of the lower bound; a start in front of p + d_lo fails the window test like one beyond it)
    words = np.zeros((W + 1) // 2 + 1, dtype=np.uint32)
    qs = np.concatenate([Q, [QSENT, QSENT]])
    wlim = max(W - 2, 0)
    for k, p in enumerate(P):
        x = int(p) + d_lo
        j = int(lb[k]) & ~1
        while True:
            da, db = int(qs[j]) - x, int(qs[j + 1]) - x
            for dd, s in ((da, int(sign_p[k])), (db, -int(sign_p[k]))):
                if 0 <= dd <= wlim:
                    words[dd >> 1] = np.uint32((int(words[dd >> 1]) + (s << (16 * (dd & 1)))) & 0xFFFFFFFF)
            if db > wlim:
                break
            j += 2
    w = words.astype(np.int64)
    lo = ((w & 0xFFFF) ^ 0x8000) - 0x8000
    hi = ((((w - lo) >> 16) & 0xFFFF) ^ 0x8000) - 0x8000  # (w - lo) is a multiple of 65536; its quotient as a signed 16-bit value
    h = np.empty(2 * w.size, dtype=np.int64)
    h[0::2], h[1::2] = lo, hi
    h = h[:W]
    g = g_0 - np.concatenate([[0], np.cumsum(h[:-1])])          # g(d_lo + i)
    n11 = n11_0 + np.concatenate([[0], np.cumsum(g[1:])])       # n11(d_lo + i)
    d = np.arange(d_lo, d_hi + 1, dtype=np.int64)
    i0 = np.maximum(0, -d)
    i1 = np.minimum(S, R - d)
    ov = np.maximum(0, i1 - i0)
    _, b1 = ones_before(P, CP, tot_s, np.clip(i1, 0, S))
    _, b0 = ones_before(P, CP, tot_s, np.clip(i0, 0, S))
    n1x = np.where(ov > 0, b1 - b0, 0)
    _, r1 = ones_before(Q, CQ, tot_r, np.clip(i1 + d, 0, R))
    _, r0 = ones_before(Q, CQ, tot_r, np.clip(i0 + d, 0, R))
    nx1 = np.where(ov > 0, r1 - r0, 0)
    return n11, n1x, nx1, ov


def two_level_scores(n11, n1x, nx1, ov, s0, s1, r0, r1):
    """Exact correlation of the mapped two-level vectors from the counts (s0, s1, r0, r1 = mapped levels 2x-1,
    aligners.py:55-57): same expression as the device's exact_score()."""
    n10, n01 = n1x - n11, nx1 - n11
    n00 = ov - n11 - n10 - n01
    return n11 * (s1 * r1) + n10 * (s1 * r0) + n01 * (s0 * r1) + n00 * (s0 * r0)


def best_lag(sub_bits, ref_bits, d_lo, d_hi, s_levels=(0.0, 1.0), r_levels=(0.0, 1.0)):
    """(score, lag) over the window: maximum score, ties to the LARGEST lag (np.argmax's first k, aligners.py:45-48)."""
    n11, n1x, nx1, ov = window_counts(sub_bits, ref_bits, d_lo, d_hi)
    m = lambda v: 2.0 * float(v) - 1.0
    sc = two_level_scores(n11.astype(np.float64), n1x.astype(np.float64), nx1.astype(np.float64), ov.astype(np.float64),
                          m(s_levels[0]), m(s_levels[1]), m(r_levels[0]), m(r_levels[1]))
    best = sc.max()
    i = int(np.flatnonzero(sc == best)[-1])
    return float(best), d_lo + i


# ---- round 5: boundary lists as a data format -------------------------------------------------------------------------
def list_from_intervals(starts, ends, length):
    """The boundary list of ``samples[a:b] = 1 for (a, b) in zip(starts, ends)`` on a zero vector of ``length`` samples,
    the way k_rasterize_runs builds it: intervals in order of their starts (already clamped to [0, length], empty ones
    dropped), a run begins where a start lies beyond every earlier end (touching intervals merge), its end is the largest
    end seen when the next run begins; ones in front of run r = ends of the runs in front of it minus their starts.
    Returns (positions, ones_before, ones)."""
    starts = np.asarray(starts, dtype=np.int64)
    ends = np.asarray(ends, dtype=np.int64)
    keep = starts < ends
    starts, ends = starts[keep], ends[keep]
    order = np.argsort(starts, kind="stable")
    starts, ends = starts[order], ends[order]
    pos, ones_before = [], []
    cmax, sum_a, sum_e, n_runs = -1, 0, 0, 0
    for a, b in zip(starts.tolist(), ends.tolist()):
        if a > cmax:  # a new run (the first one: cmax = -1)
            sum_e += cmax if cmax >= 0 else 0
            t = sum_e - sum_a  # ones in front of this run
            if n_runs:
                pos.append(cmax)
                ones_before.append(t)
            pos.append(a)
            ones_before.append(t)
            sum_a += a
            n_runs += 1
        cmax = max(cmax, b)
    ones = 0
    if n_runs:
        ones = sum_e + cmax - sum_a
        pos.append(cmax)
        ones_before.append(ones)
    assert all(0 <= p <= length for p in pos)
    return np.array(pos, dtype=np.int64), np.array(ones_before, dtype=np.int64), int(ones)


def bits_from_list(pos, length):
    """The vector a boundary list describes (k_runs_expand / list_bits32: the value at x is the parity of the boundaries
    at positions <= x)."""
    x = np.zeros(length + 1, dtype=np.int64)
    np.add.at(x, np.asarray(pos, dtype=np.int64), 1)
    return (np.cumsum(x)[:length] & 1).astype(np.uint8)


def list_bits32(pos, start):
    """32 samples [start, start + 32) from the list, as the device's list_bits32 computes them (one search, then the
    boundaries inside the window flip everything from their position up)."""
    pos = np.asarray(pos, dtype=np.int64)
    if start + 32 <= 0 or pos.size == 0:
        return 0
    lo = int(np.searchsorted(pos, start, side="right"))  # boundaries at positions <= start
    m = 0xFFFFFFFF if (lo & 1) else 0
    for k in range(lo, pos.size):
        off = int(pos[k]) - start
        if off >= 32:
            break
        m ^= (0xFFFFFFFF << off) & 0xFFFFFFFF
    return m


def prefilter_keeps_the_maximum(n11, n1x, nx1, ov, d_lo, s0, s1, r0, r1, R, S, lpt=24):
    """The fp32 prefilter of k_runs_corr on a window's exact counts: a'(d) = f11 n11(d) + E(d0) with the edge term
    E = f1x n1x + fx1 nx1 + f0 ov refreshed every four lags of a thread's `lpt` consecutive ones, all in float32; lags
    with a' >= max a' - margin are kept.  Returns (kept mask, margin, exact scores)."""
    f32 = np.float32
    k0, k1x, kx1, k11 = s0 * r0, r0 * (s1 - s0), s0 * (r1 - r0), (s1 - s0) * (r1 - r0)
    f0, f1x, fx1, f11 = f32(k0), f32(k1x), f32(kx1), f32(k11)
    estep = abs(f1x) + abs(fx1) + abs(f0)
    margin = f32(24.0) * f32(1.1920929e-7) * (estep + abs(f11)) * f32(max(R, S)) * f32(1.01) + f32(6.0) * estep * f32(1.01) + f32(1e-3)
    W = n11.size
    a = np.empty(W, dtype=np.float32)
    for i in range(W):
        i4 = (i % lpt) // 4 * 4 + (i // lpt) * lpt  # the lag whose edge term this one uses
        e = f32(f1x * f32(n1x[i4]) + f32(fx1 * f32(nx1[i4]) + f32(f0 * f32(ov[i4]))))  # (fma chains: at most as many roundings)
        a[i] = f32(f11 * f32(n11[i]) + e)
    exact = two_level_scores(n11.astype(np.float64), n1x.astype(np.float64), nx1.astype(np.float64), ov.astype(np.float64), s0, s1, r0, r1)
    return a >= a.max() - margin, float(margin), exact
