"""TEST INFRASTRUCTURE ONLY -- numpy model of the device FFT decomposition.

This is not the product path and not the reference algorithm: it is an index-math
model of the four-step, register-radix Stockham pipeline implemented by the HIP
kernels in ``ffsubsync_amd/csrc`` (pass A -> mid -> pass C), written so that every
index formula, twiddle table and HBM tile layout can be checked against
``numpy.fft`` on the CPU before the same formulas are transcribed to HIP.

Only ``tests/`` may import this module.
"""
import numpy as np

E = 16  # complex elements held per thread


def radices(L):
    """Stage radices for a length-L transform (first stage always 16)."""
    assert L >= 16 and (L & (L - 1)) == 0 and L <= 4096
    out = []
    rem = L
    while rem > 1:
        r = min(16, rem)
        out.append(r)
        rem //= r
    return out


def stage_twiddles(L):
    """Per-stage DIT twiddle tables, layout [r][j % Ns] (fp64 -> complex64).  (The kernels store only
    w^{1,2,3,4,8,12} for a radix-16 stage and form w^r = w^(4a) * w^b inside the butterfly; the model
    keeps the full table, which is the same mathematics.)"""
    tabs = []
    Ns = 1
    for R in radices(L):
        jm = np.arange(Ns)[None, :]
        r = np.arange(R)[:, None]
        tabs.append(np.exp(-2j * np.pi * r * jm / (Ns * R)).astype(np.complex64))
        Ns *= R
    return tabs


def stockham_fft(x, dtype=np.complex64):
    """Forward DFT along axis 0 of x[L, ...] using the device stage structure.

    Every thread u (< L/16) holds positions u + (L/16)*q, q<16, in registers; a stage of
    radix R runs nb=16/R butterflies b on register slots q = b + r*nb and scatters element
    (b, r) to position (j/Ns)*Ns*R + j%Ns + r*Ns with j = u + (L/16)*b.
    """
    L = x.shape[0]
    cur = x.astype(dtype)
    tabs = stage_twiddles(L)
    Ns = 1
    u = np.arange(L // E)
    for s, R in enumerate(radices(L)):
        nb = E // R
        nxt = np.empty_like(cur)
        for b in range(nb):
            j = u + (L // E) * b
            v = np.stack([cur[u + (L // E) * (b + r * nb)] for r in range(R)])  # [R, threads, ...]
            tw = tabs[s][:, j % Ns]  # [R, threads]
            v = v * tw.reshape(tw.shape + (1,) * (v.ndim - 2))
            k = np.arange(R)
            W = np.exp(-2j * np.pi * np.outer(k, k) / R).astype(dtype)
            v = np.tensordot(W, v, axes=(1, 0))
            base = (j // Ns) * Ns * R + (j % Ns)
            for r in range(R):
                nxt[base + r * Ns] = v[r]
        cur = nxt
        Ns *= R
    return cur


def column_fft(x, dtype=np.complex64):
    """Forward DFT along axis 0 for the column lengths of pass A / pass C: powers of two, or
    L = 3*LI as the kernels' col_fft does it -- thread u12 = 3u + g holds x[u12 + (L/16) q] =
    x[3 (u + LI/16 q) + g] (the inputs of sub-transform g), the three length-LI sub-transforms are
    twiddled by W_L^(g k') and combined by a radix-3 butterfly; thread (u, g) ends up with the outputs
    X[(u + LI g) + (LI/16) q]."""
    L = x.shape[0]
    if L % 3:
        return stockham_fft(x, dtype)
    LI, LT, LTI = L // 3, L // E, L // 3 // E
    out = np.empty(x.shape, dtype)
    w3 = np.exp(-2j * np.pi / 3)
    F = []
    for g in range(3):
        sub = np.empty((LI,) + x.shape[1:], dtype)
        for u in range(LTI):
            u12 = 3 * u + g
            for q in range(E):
                sub[u + LTI * q] = x[u12 + LT * q]  # register slot q of thread u12
        f = stockham_fft(sub, dtype)
        k = np.arange(LI).reshape((LI,) + (1,) * (x.ndim - 1))
        F.append(f * np.exp(-2j * np.pi * g * k / L).astype(dtype))
    for r in range(3):
        out[LI * r: LI * (r + 1)] = F[0] + (w3 ** r) * F[1] + (w3 ** (2 * r)) * F[2]
    return out.astype(dtype)


def column_fft_in_thread(x, NS, dtype=np.complex64):
    """Forward DFT along axis 0 for column lengths L = NS*LI (NS = 2 or 3) as the kernels' colnr_fft does it
    (k_pass_a3 / k_pass_c3): thread u of LTI = LI/16 holds x[NS*(u + LTI*q) + g] for EVERY g (NS*16 values), runs
    the NS length-LI sub-transforms one after the other, and finishes in registers
        X[k' + LI*r] = sum_g W_NS^(g r) W_L^(g k') F_g[k'],   k' = u + LTI*q,
    with W_L^(g k') = W_L^(g u) * W_(L/LTI)^(g q): one per-thread table value times a compile-time constant
    (L/LTI = 16*NS: W_48^q and W_24^q for NS = 3, W_32^q for NS = 2), both fp32-rounded."""
    L = x.shape[0]
    assert NS in (2, 3) and L % (NS * E) == 0
    LI = L // NS
    LTI = LI // E
    shape1 = (1,) * (x.ndim - 1)
    F = [stockham_fft(x[g::NS], dtype) for g in range(NS)]  # sub-transform g: x[NS*j + g], j = u + LTI*q
    out = np.empty(x.shape, dtype)
    for u in range(LTI):
        wu = [np.exp(-2j * np.pi * g * u / L).astype(dtype) for g in range(NS)]  # W_L^(g u): tw3[g*u] on the device
        for q in range(E):
            kq = [np.exp(-2j * np.pi * g * q / (E * NS)).astype(dtype) for g in range(NS)]  # W_(16 NS)^(g q)
            k = u + LTI * q
            t = [F[0][k]] + [(F[g][k] * kq[g]).astype(dtype) * wu[g] for g in range(1, NS)]
            for r in range(NS):
                acc = t[0].astype(dtype)
                for g in range(1, NS):
                    acc = acc + np.exp(-2j * np.pi * g * r / NS) * t[g]
                out[k + LI * r] = acc
    return out.astype(dtype).reshape((L,) + x.shape[1:]) if shape1 else out.astype(dtype)


def split_n(N):
    """N = N1 * N2 (N1: column length of pass A/C, N2: row length of the mid pass).  Lengths
    3 * 2^k keep the factor three in N1 (48..768 columns)."""
    if N % 3 == 0:
        assert N >= 48 * 256 and N <= 768 * 4096 and (N // 3) & (N // 3 - 1) == 0
        N2 = 4096 if N >= 48 * 4096 else N // 48
        return N // N2, N2
    p = int(np.log2(N))
    assert 1 << p == N and p >= 8
    p2 = min(12, p - 4)
    return 1 << (p - p2), 1 << p2


def tile_offset(x, k1, N1, C=64):
    """Element offset of (x, k1) in the tiled layout [x/CL][k1][x%CL] (x = n2 or m1; CL = 64 columns,
    i.e. 512-byte row chunks, as in the kernels' tile_base)."""
    return ((x // C) * N1 + k1) * C + (x % C)


def inter_twiddle(N, a, b):
    """W_N^(a*b) via the base/step tables the kernels use (both fp32-rounded)."""
    w = lambda p: np.exp(-2j * np.pi * (p % N) / N).astype(np.complex64)
    return w(a * b)


def correlate_model(ref_pm, sa_pm, sb_pm, N):
    """out[m] = sum_i (sa[i] + 1j*sb[i]) * ref[(i+m) % N] through the A/mid/C pipeline."""
    N1, N2 = split_n(N)
    z = np.zeros(N, np.complex64)
    z[: len(sa_pm)] = sa_pm
    if sb_pm is not None:
        z[: len(sb_pm)] += 1j * np.asarray(sb_pm)
    r = np.zeros(N, np.complex64)
    r[: len(ref_pm)] = ref_pm

    def pass_a(x):
        # columns n2: FFT over n1 of x[N2*n1 + n2], times W_N^(n2*k1); tiled store
        X = x.reshape(N1, N2)
        Y = column_fft(X)  # [k1, n2]
        k1 = np.arange(N1)[:, None]
        n2 = np.arange(N2)[None, :]
        Y = Y * inter_twiddle(N, n2, k1)
        T = np.empty(N, np.complex64)
        T[tile_offset(n2, k1, N1)] = Y
        return T

    def mid_rows(T):
        k1 = np.arange(N1)[:, None]
        n2 = np.arange(N2)[None, :]
        rows = T[tile_offset(n2, k1, N1)]  # [k1, n2]
        return stockham_fft(rows.T).T  # [k1, k2] = X[k1 + N1*k2]

    Rspec = np.conj(mid_rows(pass_a(r))) / N  # conj(R)/N in [k1][k2]
    Z = mid_rows(pass_a(z))
    P = Z * Rspec
    Y2 = stockham_fft(P.T).T  # [k1, m1]
    k1 = np.arange(N1)[:, None]
    m1 = np.arange(N2)[None, :]
    Y2 = Y2 * inter_twiddle(N, k1, m1)
    T2 = np.empty(N, np.complex64)
    T2[tile_offset(m1, k1, N1)] = Y2
    # pass C: columns m1, FFT over k1 -> out[m1 + N2*m2]
    cols = T2[tile_offset(m1, k1, N1)]  # [k1, m1]
    O = column_fft(cols)  # [m2, m1]
    return O.reshape(N)  # index m2*N2 + m1


def segmented_window_correlation(ref_pm, sub_pm, d_lo, d_hi, M):
    """Model of the block-segmented scheme in DESIGN.md section 8 (not implemented on the device yet):
    lags d in [d_lo, d_hi] of c(d) = sum_i sub[i] * ref[i + d] from length-M circular transforms of
    blocks of the candidate, with the K spectrum products added before ONE transform back.

    Block k covers sub[kB, (k+1)B) with B = M - (d_hi - d_lo); it is correlated with
    ref[kB + d_lo, kB + d_lo + M) (zero outside the vector), so lag d sits at index d - d_lo and no
    product wraps.  Returns c(d) for d = d_lo..d_hi (float64 arithmetic: the identity, not the error
    budget, is what this checks)."""
    sub_pm = np.asarray(sub_pm, dtype=np.float64)
    ref_pm = np.asarray(ref_pm, dtype=np.float64)
    W = d_hi - d_lo + 1
    B = M - (W - 1)
    assert B > 0
    K = -(-len(sub_pm) // B)
    acc = np.zeros(M, dtype=np.complex128)
    for k in range(K):
        s = np.zeros(M)
        blk = sub_pm[k * B:(k + 1) * B]
        s[: len(blk)] = blk
        r = np.zeros(M)
        lo = k * B + d_lo
        src_lo, src_hi = max(lo, 0), min(lo + M, len(ref_pm))
        if src_hi > src_lo:
            r[src_lo - lo: src_hi - lo] = ref_pm[src_lo:src_hi]
        acc += np.conj(np.fft.fft(s)) * np.fft.fft(r)  # sum_i s[i] r[(i + m) % M] in the spectrum domain
    return np.real(np.fft.ifft(acc))[:W], K


# ---- lag window in output-index space (mirrors load_window / in_window / in_window_t of csrc/ffs_kernels.h) ----------
NEVER = 0x40000000


def window_two_ranges(lo, hi, n, seg=False, seg_shift=0):
    """The window [lo, hi] of lags as (start, width) ranges of output indices m: lags 0..hi sit at m = d, negative ones
    at m = d + n (circular transform); block-segmented mode: m = d - seg_shift.  Returns [(a0, w0), (a1, w1)]; m passes
    iff a_i <= m <= a_i + w_i for one of them (NEVER = empty)."""
    r = [(NEVER, 0), (NEVER, 0)]
    if hi >= lo:
        if seg:
            r[0] = (lo - seg_shift, hi - lo)
        else:
            if hi >= 0:
                a = max(lo, 0)
                r[0] = (a, hi - a)
            if lo < 0:
                b = min(hi, -1)
                r[1] = (n + lo, b - lo)
    return r


def window_one_range(lo, hi, n, seg=False, seg_shift=0):
    """The same window as ONE range test for 0 <= m < n: (g0, gw, inv) with  m passes iff (g0 <= m <= g0 + gw) != inv
    -- a window is one range of m or, with lags on both sides of zero, everything but one gap in the middle."""
    g0, gw, inv = NEVER, 0, False
    if hi >= lo:
        if seg:
            g0, gw = lo - seg_shift, hi - lo
        elif lo >= 0:
            g0, gw = lo, hi - lo
        elif hi < 0:
            g0, gw = n + lo, hi - lo
        else:
            inv = True
            if n + lo - 1 >= hi + 1:
                g0, gw = hi + 1, n + lo - hi - 2
    return g0, gw, inv


def paired_real_columns(ref_cols, cand_cols, dtype=np.complex64):
    """Model of the paired first-pass transform (k_pass_a<.., PM> / k_pass_a3<.., PM>): two REAL column tiles share one
    complex column transform z = ref + i*cand; with Z = column_fft(z) the stored half rows k = 0..L/2 are
        ref^[k] = (Z[k] + conj Z[(L-k) % L]) / 2,     cand^[k] = (Z[k] - conj Z[(L-k) % L]) / (2i).
    Returns (ref^[:L/2+1], cand^[:L/2+1]) -- what the half slots of the reference and of a single last candidate hold
    before the inter-pass twiddle."""
    L = ref_cols.shape[0]
    z = (np.asarray(ref_cols, np.float32) + 1j * np.asarray(cand_cols, np.float32)).astype(dtype)
    Z = column_fft(z, dtype)
    k = np.arange(L // 2 + 1)
    m = np.conj(Z[(L - k) % L])
    return (0.5 * (Z[k] + m)).astype(dtype), (-0.5j * (Z[k] - m)).astype(dtype)


def paired_hermitian_columns(A_half, B_half, L, dtype=np.complex64):
    """Model of the paired last-pass transform (k_pass_c3<.., PAIRED>): two columns whose spectra are Hermitian in the
    column index (a single real candidate's product spectrum), each given by its stored rows 0..L/2, share one complex
    column transform of V = A + i*B (rows above L/2 rebuilt by conjugation): Re = column A's values, Im = column B's.
    Forward-transform convention as in the kernels (the row pass has already conjugated)."""
    k = np.arange(L)
    idx = np.where(k <= L // 2, k, L - k)
    full = lambda h: np.where((k <= L // 2).reshape((L,) + (1,) * (h.ndim - 1)), h[idx], np.conj(h[idx]))
    V = (full(np.asarray(A_half)) + 1j * full(np.asarray(B_half))).astype(dtype)
    out = column_fft(V, dtype)
    return out.real, out.imag
