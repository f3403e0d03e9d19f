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
    words = np.zeros((W + 1) // 2 + 1, dtype=np.uint32)
    for k, p in enumerate(P):
        j = lb[k]
        while j < Q.size and Q[j] - p <= d_hi - 1:
            idx = int(Q[j] - p - d_lo)
            s = int(sign_p[k]) * (-1 if (j & 1) else 1)
            words[idx >> 1] = np.uint32((int(words[idx >> 1]) + (s << (16 * (idx & 1)))) & 0xFFFFFFFF)
            j += 1
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
