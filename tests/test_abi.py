"""libffsalign.so loads on a CPU-only box and exports every symbol include/ffsubsync_amd.h declares
(no compute calls here)."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

from ffsubsync_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "ffsubsync_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ffs_[a-z_0-9]+)\s*\(", text)))


def test_library_is_built():
    assert os.path.exists(_native.library_path()), "run __graft_entry__.build() first"


def test_exports_every_declared_symbol():
    lib = ctypes.CDLL(_native.library_path())
    syms = _header_symbols()
    assert set(syms) == set(_native.EXPORTED_SYMBOLS)
    for s in syms:
        assert getattr(lib, s) is not None


def test_host_only_entry_points():
    lib = _native.load()
    assert lib.ffs_version() >= 200
    assert _native.fft_length(0, 5) == 0
    for x in list(range(2, 3000)) + [2 ** k + d for k in range(4, 25) for d in (-1, 0, 1)]:
        # aligners.py:67-68
        assert _native.fft_length(x - 1, 1) == int(2 ** math.ceil(math.log(x, 2)))


def test_plan_length_never_exceeds_reference_length():
    import numpy as np

    rng = np.random.RandomState(0)
    for _ in range(2000):
        r, s = int(rng.randint(1, 900000)), int(rng.randint(1, 900000))
        mo = [None, 0, 100, 6000, 10 ** 7][rng.randint(5)]
        n_ref, n = _native.fft_length(r, s), _native.plan_length(r, s, mo)
        m = n // 3 if n % 3 == 0 else n  # a power of two, or three times one (column passes take a factor 3)
        assert n <= n_ref and (m & (m - 1)) == 0
        if mo is None:
            # every lag with a non-empty overlap, d in (-S, R): a circular correlation of length >= R+S-1
            assert n >= r + s - 1 and (n == n_ref or n == n_ref // 4 * 3)
        elif n > 2:
            # the prefixes that can reach the lag window fit without wrapping onto themselves
            assert n >= max(min(s, r + mo), min(r, s + mo))
    assert _native.plan_length(720000, 750751, 6000) == 3 << 18   # 786 432 >= 750 751 + 6000 + 1
    assert _native.plan_length(720000, 790000, 6000) == 3 << 18   # only 726 000 candidate samples reach the window
    assert _native.plan_length(790000, 720000, 6000) == 3 << 18   # ... and 726 000 reference samples
    assert _native.plan_length(790000, 790000, 6000) == 1 << 20   # needs 796 001
    assert _native.plan_length(720000, 750751, 10 ** 7) == 3 << 19  # window as wide as the reference's: R+S-1 lags
    assert _native.plan_length(720000, 750751, None) == 3 << 19 and _native.fft_length(720000, 750751) == 1 << 21
    assert _native.plan_length(1 << 20, 1 << 20, None) == 1 << 21   # R+S = 2^21 exactly
    assert _native.plan_length(3000, 3000, 100) == 4096            # below the 3*2^k range


def test_bad_arguments_are_reported_not_thrown():
    lib = _native.load()
    handle = ctypes.c_void_p()
    rc = lib.ffs_plan_create(0, 1000, 1, 7, ctypes.byref(handle))  # not a power of two
    assert rc == -1 and b"power of two" in lib.ffs_last_error()
    with pytest.raises(_native.NativeError):
        _native.check(rc)


def test_batched_rasteriser_validates_its_tables_before_touching_the_gpu():
    lib = _native.load()
    i64 = lambda *v: np.array(v, dtype=np.int64)
    s, e = i64(0, 1_000_000), i64(500_000, 2_000_000)
    first, count, word, length = i64(0), i64(2), i64(0), i64(202)
    ratio = np.array([1.0])
    out = ctypes.c_void_p(0x1000)  # never dereferenced: every case below fails validation first

    def call(n_subs=2, first=first, count=count, word=word, length=length, n_vec=1, out_words=7, out=out, vec_first=None):
        return lib.ffs_rasterize_batch_bits(s.ctypes.data, e.ctypes.data, None, n_subs,
                                            first.ctypes.data if vec_first is None else vec_first, count.ctypes.data,
                                            ratio.ctypes.data, word.ctypes.data, length.ctypes.data, n_vec, 100.0, 0.0, out,
                                            out_words, None)

    assert call(n_vec=-1) == -1
    assert call(vec_first=0) == -1 and b"null vector table" in lib.ffs_last_error()
    assert call(count=i64(3)) == -1 and b"subtitle range" in lib.ffs_last_error()
    assert call(first=i64(-1)) == -1
    assert call(word=i64(1)) == -1 and b"output range" in lib.ffs_last_error()  # 7 words needed from word 1 of 7
    assert call(length=i64(1 << 31)) != 0
    assert call(out=ctypes.c_void_p(0x1002)) == -1 and b"misaligned" in lib.ffs_last_error()
    assert call(n_vec=0, out_words=0) == 0  # nothing to do is not an error
    assert lib.ffs_raster_lengths(None, None, 1, 100.0, None) == -1
    assert lib.ffs_raster_lengths(None, None, 0, 100.0, None) == 0


def test_result_struct_layouts():
    assert _native.CAND_RESULT_DTYPE.fields["offset"][1] == 8 and _native.CAND_RESULT_DTYPE.fields["flags"][1] == 20
    assert _native.PAIR_RESULT_DTYPE.fields["best_cand"][1] == 16
