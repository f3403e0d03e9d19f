"""TEST INFRASTRUCTURE ONLY -- restatement of the subtitle rasterisation that produces every
``substring`` the aligner sees: ``SubtitleScaler.fit`` (ffsubsync/subtitle_transformers.py:35-47)
followed by ``SubtitleSpeechTransformer.fit`` (ffsubsync/speech_transformers.py:957-980), on plain
(start, end, is_metadata) records.  Uses ``datetime.timedelta`` itself for the microsecond rounding,
exactly as the reference does.  Pinned to the unmodified reference classes by
tests/golden/make_golden.py -> tests/golden/raster_golden.npz.
"""
from datetime import timedelta

import numpy as np


def scale(start_us, end_us, ratio):
    """subtitle_transformers.py:35-47: timedelta(seconds=total_seconds * scale_factor)."""
    out = []
    for s, e in zip(start_us, end_us):
        ts, te = timedelta(microseconds=int(s)), timedelta(microseconds=int(e))
        out.append((timedelta(seconds=ts.total_seconds() * ratio), timedelta(seconds=te.total_seconds() * ratio)))
    return out


def rasterize(start_us, end_us, is_metadata, ratio, sample_rate=100, start_seconds=0):
    """speech_transformers.py:957-980 -> float64 samples with amplitude min(1/ratio, 1)."""
    subs = scale(start_us, end_us, ratio)
    max_time = 0
    for _, te in subs:
        max_time = max(max_time, te.total_seconds())
    samples = np.zeros(int(max_time * sample_rate) + 2, dtype=float)
    for (ts, te), meta in zip(subs, is_metadata):
        if meta:
            continue
        start = int(round((ts.total_seconds() - start_seconds) * sample_rate))
        duration = te.total_seconds() - ts.total_seconds()
        end = start + int(round(duration * sample_rate))
        samples[start:end] = min(1.0 / ratio, 1.0)
    return samples


def intervals(start_us, end_us, is_metadata, ratio, sample_rate=100, start_seconds=0, length=None):
    """The [start, end) sample intervals ``rasterize`` paints (speech_transformers.py:966-977), clamped the way
    ``samples[start:end] = ...`` clamps them on a vector of ``length`` samples (Python slice semantics: negative indices
    count from the end); metadata lines and empty slices dropped.  [n, 2] int64."""
    subs = scale(start_us, end_us, ratio)
    if length is None:
        max_time = 0
        for _, te in subs:
            max_time = max(max_time, te.total_seconds())
        length = int(max_time * sample_rate) + 2
    out = []
    for (ts, te), meta in zip(subs, is_metadata):
        if meta:
            continue
        start = int(round((ts.total_seconds() - start_seconds) * sample_rate))
        end = start + int(round((te.total_seconds() - ts.total_seconds()) * sample_rate))
        a, b, _ = slice(start, end).indices(length)
        if a < b:
            out.append((a, b))
    return np.array(out, dtype=np.int64).reshape(-1, 2)


def synth_subtitles(seed, n=180, minutes=10.0):
    """Seeded subtitle records with microsecond timestamps (as parsed srt times are), a few of them
    flagged as metadata."""
    rng = np.random.RandomState(seed)
    gaps = rng.uniform(0.2, 6.0, n)
    durs = rng.uniform(0.4, 5.0, n)
    ends = np.cumsum(gaps + durs)
    starts = ends - durs
    scale_t = minutes * 60.0 / ends[-1]
    start_us = np.rint(starts * scale_t * 1e3).astype(np.int64) * 1000  # srt has millisecond stamps
    end_us = np.rint(ends * scale_t * 1e3).astype(np.int64) * 1000
    meta = (rng.rand(n) < 0.05).astype(np.uint8)
    return start_us, end_us, meta
