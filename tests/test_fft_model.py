"""The numpy model of the device FFT pipeline (oracle/fft_model.py) against numpy.fft: validates the
stage structure, twiddle tables, tile layout and lag indexing that the HIP kernels transcribe."""
import numpy as np
import pytest

from oracle import fft_model as fm


@pytest.mark.parametrize("L", [16, 32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_stockham_stages_match_numpy(L):
    rng = np.random.RandomState(L)
    x = rng.randn(L, 2) + 1j * rng.randn(L, 2)
    got = fm.stockham_fft(x, np.complex128)
    assert np.abs(got - np.fft.fft(x, axis=0)).max() < 1e-5 * np.sqrt(L)


@pytest.mark.parametrize("L", [48, 96, 192, 384, 768])
def test_radix3_columns_match_numpy(L):
    rng = np.random.RandomState(L)
    x = rng.randn(L, 3) + 1j * rng.randn(L, 3)
    got = fm.column_fft(x, np.complex128)
    assert np.abs(got - np.fft.fft(x, axis=0)).max() < 1e-5 * np.sqrt(L)


@pytest.mark.parametrize("L,NS", [(192, 3), (384, 3), (768, 3), (512, 2)])
def test_sub_transforms_in_one_thread_match_numpy(L, NS):
    """colnr_fft (k_pass_a3 / k_pass_c3): all NS sub-transforms of a column in one thread, last radix step in
    registers with W_L^(g k') = W_L^(g u) * W_(16 NS)^(g q)."""
    rng = np.random.RandomState(L + NS)
    x = rng.randn(L, 2) + 1j * rng.randn(L, 2)
    got = fm.column_fft_in_thread(x, NS, np.complex128)
    assert np.abs(got - np.fft.fft(x, axis=0)).max() < 1e-5 * np.sqrt(L)  # the stage tables are fp32-rounded
    got32 = fm.column_fft_in_thread(x.astype(np.complex64), NS, np.complex64)
    assert np.abs(got32 - np.fft.fft(x, axis=0)).max() < 2e-5 * np.sqrt(L)


@pytest.mark.parametrize("N", [4096, 8192, 1 << 15, 1 << 17, 3 << 12, 3 << 14, 3 << 16])
def test_pipeline_matches_direct_correlation(N):
    rng = np.random.RandomState(N % 97)
    R, Sa, Sb = N // 2 - 3, N // 3, N // 2 - 100
    ref = 2.0 * (rng.rand(R) > 0.6) - 1
    sa = 2.0 * (rng.rand(Sa) > 0.6) - 1
    sb = 0.92 * (2.0 * (rng.rand(Sb) > 0.6) - 1)
    out = fm.correlate_model(ref, sa, sb, N)

    def direct(s):
        a = np.zeros(N)
        a[: len(s)] = s
        b = np.zeros(N)
        b[:R] = ref
        return np.real(np.fft.ifft(np.conj(np.fft.fft(a)) * np.fft.fft(b)))

    tol = 5e-7 * np.log2(N) * np.sqrt(R * Sb)
    assert np.abs(out.real - direct(sa)).max() < tol
    assert np.abs(out.imag - direct(sb)).max() < tol


def test_split_rule():
    for p in range(12, 25):
        n1, n2 = fm.split_n(1 << p)
        assert n1 * n2 == 1 << p and 16 <= n1 <= 4096 and 256 <= n2 <= 4096
    for p in range(12, 21):
        n1, n2 = fm.split_n(3 << p)
        assert n1 * n2 == 3 << p and n1 % 3 == 0 and 48 <= n1 <= 768 and 256 <= n2 <= 4096


@pytest.mark.parametrize("R,S,d_lo,d_hi,M", [(5000, 5300, -600, 600, 2048), (5000, 4100, -599, 600, 4096),
                                              (72000, 75075, -600, 600, 32768), (3000, 9000, -50, 2000, 4096)])
def test_block_segmented_accumulation_identity(R, S, d_lo, d_hi, M):
    """The next-round scheme of DESIGN.md section 8: window lags from K block transforms whose
    spectrum products are summed before one inverse transform."""
    rng = np.random.RandomState(R + S)
    ref = 2.0 * (rng.rand(R) < 0.4) - 1
    sub = 0.97 * (2.0 * (rng.rand(S) < 0.4) - 1)
    got, K = fm.segmented_window_correlation(ref, sub, d_lo, d_hi, M)
    assert K >= 2
    rp = np.concatenate([np.zeros(max(0, -d_lo)), ref, np.zeros(S + d_hi)])
    off = max(0, -d_lo)
    exp = np.array([np.dot(sub, rp[off + d: off + d + S]) for d in range(d_lo, d_hi + 1)])
    assert np.abs(got - exp).max() < 1e-7 * max(1.0, np.abs(exp).max())


def test_one_range_window_test_equals_the_two_range_one():
    """The last pass tests a lag window with ONE unsigned range compare (in_window_t in csrc/ffs_kernels.h); the model of
    that mapping against the two-range definition it replaced, for every output index of random windows -- one-sided,
    two-sided, empty, covering everything, block-segmented."""
    rng = np.random.RandomState(5)
    for trial in range(400):
        n = int(rng.choice([16, 48, 64, 96, 256, 768]))
        seg = bool(trial % 5 == 4)
        if seg:
            shift = int(rng.randint(-n, 1))
            lo = shift + int(rng.randint(0, n))
            hi = lo + int(rng.randint(-1, shift + n - lo))     # 0 <= lo - shift <= hi - shift < n  (or empty)
        else:
            shift = 0
            lo = int(rng.randint(-n + 1, n))
            hi = int(rng.randint(lo - 1, min(n - 1, lo + n - 1) + 1))   # at most n lags, hi < n (hi = lo - 1: empty)
        m = np.arange(n)
        two = np.zeros(n, bool)
        for a, w in fm.window_two_ranges(lo, hi, n, seg, shift):
            two |= (m >= a) & (m <= a + w)
        g0, gw, inv = fm.window_one_range(lo, hi, n, seg, shift)
        one = ((m >= g0) & (m <= g0 + gw)) != inv
        assert np.array_equal(one, two), (trial, n, lo, hi, seg, shift)
        # and both are the lags of the window
        if hi >= lo:
            lag = (m + shift) if seg else np.where(m <= hi, m, m - n) if hi >= 0 else m - n
            if not seg and hi >= 0:
                lag = np.where(m <= hi, m, m - n)
            assert np.array_equal(two, (lag >= lo) & (lag <= hi)), (trial, n, lo, hi, seg, shift)


def test_two_real_columns_through_one_complex_transform():
    """The identity behind the paired first pass (reference + single last candidate of a transform group): the half
    spectra separated from one complex column transform equal the two real transforms, for power-of-two and 3*2^k
    column lengths."""
    rng = np.random.RandomState(11)
    for L in (16, 32, 64, 256, 48, 192, 384):
        ref = (rng.rand(L, 5) < 0.4).astype(np.float32) * 2 - 1
        cand = (rng.rand(L, 5) < 0.6).astype(np.float32) * 0.96
        r_half, c_half = fm.paired_real_columns(ref, cand)
        want_r = np.fft.fft(ref.astype(np.float64), axis=0)[: L // 2 + 1]
        want_c = np.fft.fft(cand.astype(np.float64), axis=0)[: L // 2 + 1]
        tol = 2e-5 * L
        assert np.abs(r_half - want_r).max() < tol, L
        assert np.abs(c_half - want_c).max() < tol, L


def test_two_hermitian_columns_through_one_complex_transform():
    """The identity behind the paired last pass: two column spectra that are Hermitian in the column index (real
    outputs) ride one complex transform as real and imaginary part."""
    rng = np.random.RandomState(12)
    for L in (16, 64, 48, 192):
        a, b = rng.randn(L, 4), rng.randn(L, 4)
        A, B = np.fft.ifft(a, axis=0) * L, np.fft.ifft(b, axis=0) * L   # spectra whose FORWARD transform is L*a, L*b
        ra, rb = fm.paired_hermitian_columns(A[: L // 2 + 1], B[: L // 2 + 1], L)
        assert np.abs(ra - L * a).max() < 3e-4 * L, L
        assert np.abs(rb - L * b).max() < 3e-4 * L, L
