"""CPU checks of the VAD oracle's own wiring (parity with a real detector is unpinned, see
oracle/vad_oracle.py)."""
import numpy as np

from oracle import vad_oracle as vo


def test_frame_geometry_and_labels():
    assert vo.frame_len(100, 48000) == 480
    pcm, state = vo.synth_pcm(480 * 50 + 100, seed=1)
    lab = vo.detect(pcm, non_speech_label=0.25)
    assert lab.size == 51 and set(np.unique(lab)) <= {0.25, 1.0}
    assert np.array_equal(lab[:50] == 1.0, state[:50])
    assert np.array_equal(lab, vo.detect_fast(pcm, non_speech_label=0.25))


def test_threshold_boundary_is_inclusive():
    blk = np.zeros(480, np.int16)
    # sum(x^2) == 1e5 * 480 exactly: 480 samples, 300 of value 400 -> 48_000_000
    blk[:300] = 400
    assert vo.frame_energy_db(blk) == 50.0 and vo.detect(blk)[0] == 1.0
    blk[0] = 399
    assert vo.detect(blk)[0] == 0.0
    assert vo.detect(np.zeros(480, np.int16))[0] == 0.0  # -200 dB


def test_chunk_loop_concatenates():
    pcm, _ = vo.synth_pcm(480 * 25000 + 7, seed=2)
    a = vo.chunked_detect(pcm)
    assert a.size == 25001 and np.array_equal(a, vo.detect_fast(pcm))
