"""Multi-segment sparse reference (host orchestration): the reference's own checks
(tests/test_multi_segment.py:28-108) against the mirror, with extraction and probing stubbed."""
import numpy as np
import pytest

import ffsubsync_amd.speech_transformers as st

SR = 100


def _transformer(**overrides):
    kwargs = dict(vad="webrtc", sample_rate=SR, frame_rate=48000, non_speech_label=0.0, segment_duration=60)
    kwargs.update(overrides)
    return st.MultiSegmentVideoSpeechTransformer(**kwargs)


def test_segment_starts():
    t = _transformer(segment_count=8)
    starts = t._segment_starts(600.0)
    assert len(starts) == 8 and starts == sorted(starts) and starts[0] == 0
    assert all(0 <= s <= 600 - t.segment_duration for s in starts)
    assert t._segment_starts(40.0) == [0]
    t = _transformer(segment_count=6, skip_intro_outro=True)
    starts = t._segment_starts(900.0)
    assert starts[0] >= t.START_MARGIN_SECONDS and starts[-1] <= 900 - t.END_MARGIN_SECONDS - t.segment_duration
    assert _transformer(vad="subs_then_webrtc").vad == "webrtc" and _transformer(vad="fused:union").vad == "fused:union"


def test_fit_assembles_sparse_signal_and_tolerates_failures(monkeypatch):
    monkeypatch.setattr(st, "_probe_duration", lambda *a, **k: 120.0)
    t = _transformer(segment_count=3, segment_duration=10)
    monkeypatch.setattr(t, "_extract_segment_speech", lambda fname, start: (start, np.ones(10 * SR)))
    speech = t.fit("ref.mkv").transform()
    assert len(speech) == int(120 * SR) + 2
    starts = t._segment_starts(120.0)
    for s in starts:
        assert np.all(speech[s * SR: s * SR + 10 * SR] == 1.0)
    assert speech[starts[0] * SR + 10 * SR + 5] == 0.0

    failing = starts[0]

    def flaky(fname, start):
        if start == failing:
            raise RuntimeError("ffmpeg blew up")
        return start, np.ones(10 * SR)

    monkeypatch.setattr(t, "_extract_segment_speech", flaky)
    speech = t.fit("ref.mkv").transform()
    assert np.all(speech[failing * SR: failing * SR + 10 * SR] == 0.0)
    assert np.all(speech[starts[1] * SR: starts[1] * SR + 10 * SR] == 1.0)

    monkeypatch.setattr(t, "_extract_segment_speech", lambda fname, start: (start, np.zeros(10 * SR)))
    with pytest.raises(ValueError, match="Unable to detect speech"):
        t.fit("ref.mkv")
    monkeypatch.setattr(st, "_probe_duration", lambda *a, **k: (_ for _ in ()).throw(OSError("no ffprobe")))
    with pytest.raises(ValueError, match="needs the reference duration"):
        t.fit("ref.mkv")
