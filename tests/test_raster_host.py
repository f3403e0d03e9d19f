"""Subtitle rasteriser, host side (no GPU): the oracle against the committed outputs of the
unmodified reference classes, and the library's host-only interval arithmetic (timedelta
microsecond rounding, round-half-even, Python slice clamping) against the oracle."""
import os

import numpy as np
import pytest

from ffsubsync_amd import _native
from oracle import raster_oracle as ro

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "raster_golden.npz"))
RATIOS = GOLD["ratios"]
CASES = ["a", "b", "late_start"]


def _from_intervals(n, iv):
    out = np.zeros(n, np.uint8)
    for a, b in iv:
        out[a:b] = 1
    return out


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_rasters(name):
    s, e, m = GOLD[name + "_start_us"], GOLD[name + "_end_us"], GOLD[name + "_meta"]
    ss = int(GOLD[name + "_start_seconds"])
    for j, r in enumerate(RATIOS):
        arr = ro.rasterize(s, e, m, float(r), 100, ss)
        assert np.array_equal((arr > 0).astype(np.uint8), GOLD["%s_r%d" % (name, j)])
        assert set(np.unique(arr)) <= {0.0, min(1.0 / float(r), 1.0)}


@pytest.mark.parametrize("name", CASES)
def test_library_intervals_match_reference_rasters(name):
    s, e, m = GOLD[name + "_start_us"], GOLD[name + "_end_us"], GOLD[name + "_meta"]
    ss = int(GOLD[name + "_start_seconds"])
    for j, r in enumerate(RATIOS):
        want = GOLD["%s_r%d" % (name, j)]
        n = _native.raster_length(e, float(r), 100.0)
        assert n == want.size
        iv = _native.raster_intervals(s, e, m, float(r), 100.0, float(ss), n)
        assert np.array_equal(_from_intervals(n, iv), want)


def test_library_intervals_random_ratios_and_offsets():
    """gss evaluates arbitrary ratios in [0.9, 1.1]; negative starts wrap like Python slices."""
    rng = np.random.RandomState(0)
    for trial in range(60):
        s, e, m = ro.synth_subtitles(100 + trial, n=40, minutes=2.0)
        ratio = float(rng.uniform(0.9, 1.1))
        ss = float(rng.choice([0.0, 3.0, 17.0, 500.0]))
        want = ro.rasterize(s, e, m, ratio, 100, ss)
        n = _native.raster_length(e, ratio, 100.0)
        assert n == want.size
        iv = _native.raster_intervals(s, e, m, ratio, 100.0, ss, n)
        assert np.array_equal(_from_intervals(n, iv), (want > 0).astype(np.uint8)), (trial, ratio, ss)


def test_empty_and_all_metadata():
    z = np.zeros(0, np.int64)
    assert _native.raster_length(z, 1.0, 100.0) == 2
    s, e, m = ro.synth_subtitles(5, n=10, minutes=1.0)
    n = _native.raster_length(e, 1.0, 100.0)
    assert _native.raster_intervals(s, e, np.ones_like(m), 1.0, 100.0, 0.0, n).shape == (0, 2)


def test_raster_lengths_of_many_vectors_equal_the_per_track_scan():
    """ffs_raster_lengths takes each track's largest end time only: the scaling (division, multiplication, microsecond
    rounding) is monotone, so that is the track's largest scaled end -- checked against the scan over every subtitle,
    with ratios that land on half microseconds."""
    rng = np.random.RandomState(3)
    ends, ratios, want = [], [], []
    for trial in range(300):
        s, e, m = ro.synth_subtitles(700 + trial, n=int(rng.randint(1, 30)), minutes=float(rng.uniform(0.2, 200.0)))
        if trial % 7 == 0:
            e = e // 2 * 2 + 1  # odd microsecond counts: x 0.5 / 1.5 / 2.5 hits the half-even branch
        ratio = float(rng.choice([0.5, 1.5, 2.5, 1.0, 1.001, 0.999, 25.0 / 23.976, rng.uniform(0.9, 1.1)]))
        ends.append(int(e.max()))
        ratios.append(ratio)
        want.append(_native.raster_length(e, ratio, 100.0))
    got = _native.raster_lengths(np.array(ends), np.array(ratios), 100.0)
    assert got.tolist() == want
    assert _native.raster_lengths(np.zeros(2, np.int64), np.array([1.0, 1.1]), 100.0).tolist() == [2, 2]
    assert _native.raster_lengths(np.zeros(0, np.int64), np.zeros(0), 100.0).size == 0
