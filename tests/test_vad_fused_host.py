"""Fusion arithmetic of the fused detector, ported from the reference's tests/test_vad_fused.py:21-54
(the two underlying detector factories are stubbed exactly as the reference's tests stub them)."""
import numpy as np
import pytest

import ffsubsync_amd.speech_transformers as st


def _stub_factories(monkeypatch, webrtc_result, silero_result):
    monkeypatch.setattr(st, "_make_webrtcvad_detector", lambda *a, **k: (lambda seg: np.asarray(webrtc_result, dtype=float)))
    monkeypatch.setattr(st, "_make_silero_detector", lambda *a, **k: (lambda seg: np.asarray(silero_result, dtype=float)))


def test_fused_intersection_is_elementwise_min(monkeypatch):
    _stub_factories(monkeypatch, [1.0, 1.0, 0.0], [1.0, 0.0, 0.0])
    assert list(st._make_fused_detector(100, 48000, 0.0, "intersection")(b"")) == [1.0, 0.0, 0.0]


def test_fused_union_is_elementwise_max(monkeypatch):
    _stub_factories(monkeypatch, [1.0, 1.0, 0.0], [1.0, 0.0, 0.0])
    assert list(st._make_fused_detector(100, 48000, 0.0, "union")(b"")) == [1.0, 1.0, 0.0]


def test_fused_weighted_is_silero_heavy_and_default(monkeypatch):
    _stub_factories(monkeypatch, [1.0, 0.0], [0.0, 1.0])
    assert np.allclose(st._make_fused_detector(100, 48000, 0.0, "weighted")(b""), [0.4, 0.6])
    assert np.allclose(st._make_fused_detector(100, 48000, 0.0)(b""), [0.4, 0.6])


def test_fused_clips_to_common_length(monkeypatch):
    _stub_factories(monkeypatch, [1.0, 1.0, 1.0], [1.0, 1.0])
    assert len(st._make_fused_detector(100, 48000, 0.0, "union")(b"")) == 2


def test_fused_rejects_unknown_strategy():
    with pytest.raises(ValueError, match="unknown fused VAD strategy"):
        st._make_fused_detector(100, 48000, 0.0, "bogus")


def test_transformer_selects_detector_by_substring(monkeypatch):
    _stub_factories(monkeypatch, [1.0, 0.0, 1.0], [0.0, 0.0, 1.0])
    t = st.PCMSpeechTransformer("fused:union", 100, 48000, 0.0).fit(b"\\x00\\x00" * 480)
    assert t.transform().tolist() == [1.0, 0.0, 1.0]
    with pytest.raises(ValueError, match="unknown vad"):
        st.PCMSpeechTransformer("nonsense").fit(b"\\x00\\x00")
