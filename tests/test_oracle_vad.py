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


def test_tokenizer_restatement_basics():
    """Hand-checkable behaviour of the (unpinned) auditok tokenizer restatement."""
    v = np.zeros(300, bool)
    v[10:40] = True          # 30 valid frames -> token [10, 39 + 25 trailing tolerated silence]
    v[100:110] = True        # 10 < min_length 20 -> dropped ... but trailing silence counts: 10 + 25 = 35 >= 20
    out = vo.tokenize_chunk(v)
    assert out[:10].sum() == 0 and out[10:65].all() and out[65:100].sum() == 0
    assert out[100:135].all() and out[135:].sum() == 0
    # a run longer than max_length is cut into contiguous tokens; the reference's marker assignment then
    # leaves the cumulative sum at 1 until the chunk ends (only one end marker survives)
    w = np.zeros(1500, bool)
    w[50:900] = True
    out = vo.tokenize_chunk(w)
    assert out[:50].sum() == 0 and out[50:].all()
    # non-default label: silence after a token reads as the label, before the first token as 0
    out = vo.tokenize_chunk(v, non_speech_label=0.25)
    assert out[5] == 0.0 and out[70] == 0.25 and out[20] == 1.0
    # chunks are tokenised independently
    z = np.concatenate([v, v])
    assert np.array_equal(vo.tokenize(z, chunk_frames=300), np.concatenate([vo.tokenize_chunk(v)] * 2))


def test_scan_formulation_of_the_tokenizer_equals_the_state_machine():
    """oracle/vad_oracle.py::tokenize_chunk_scan (the numpy model of the device's scan-based kernel) against the
    restated state machine: random validity patterns, the reference's parameters and degenerate ones (no tolerated
    silence, min == max, tolerated silence as long as a token), three non-speech labels."""
    rng = np.random.RandomState(11)
    for trial in range(160):
        n = int(rng.choice([1, 2, 7, 60, 500, 1500]))
        p_on = rng.choice([0.02, 0.2, 0.5, 0.8, 0.95])
        runs = rng.geometric(1.0 / rng.choice([1, 3, 15, 80, 700]), size=n + 4)
        valid = np.repeat(rng.rand(runs.size) < p_on, runs)[:n]
        mn, mx, msil = [(20, 500, 25), (20, 500, 25), (3, 10, 2), (5, 5, 1), (1, 7, 0), (4, 40, 30), (2, 9, 9), (3, 12, 11)][rng.randint(8)]
        label = float(rng.choice([0.0, 0.25, -1.0]))
        tok = vo._Tokenizer(mn, mx, msil)
        marks = np.zeros(n + 1)
        for s, e in tok.tokenize(valid):
            marks[s] = 1.0
            marks[e + 1] = label - 1.0
        want = np.clip(np.cumsum(marks)[:-1], 0.0, 1.0)
        got = vo.tokenize_chunk_scan(valid, label, mn, mx, msil)
        assert np.array_equal(got, want), (trial, n, (mn, mx, msil), label)


def test_word_formulation_of_the_tokenizer_equals_the_state_machine():
    """oracle/vad_oracle.py::tokenize_chunk_words (the model of the round-6 kernel: validity and island starts as bit
    words, markers written island by island) against the restated state machine, same cases as above plus lengths
    around the 64-frame word."""
    rng = np.random.RandomState(12)
    for trial in range(240):
        n = int(rng.choice([1, 2, 7, 63, 64, 65, 128, 500, 1500]))
        p_on = rng.choice([0.02, 0.2, 0.5, 0.8, 0.95])
        runs = rng.geometric(1.0 / rng.choice([1, 3, 15, 80, 700]), size=n + 4)
        valid = np.repeat(rng.rand(runs.size) < p_on, runs)[:n]
        mn, mx, msil = [(20, 500, 25), (20, 500, 25), (3, 10, 2), (5, 5, 1), (1, 7, 0), (4, 40, 30), (2, 9, 9), (3, 12, 11),
                        (1, 1, 0), (0, 3, -1), (2, 70, 100)][rng.randint(11)]
        label = float(rng.choice([0.0, 0.25, -1.0]))
        tok = vo._Tokenizer(mn, mx, msil)
        marks = np.zeros(n + 1)
        for s, e in tok.tokenize(valid):
            marks[s] = 1.0
            marks[e + 1] = label - 1.0
        want = np.clip(np.cumsum(marks)[:-1], 0.0, 1.0)
        got = vo.tokenize_chunk_words(valid, label, mn, mx, msil)
        assert np.array_equal(got, want), (trial, n, (mn, mx, msil), label)
