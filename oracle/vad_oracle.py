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


class _Tokenizer:
    """auditok 0.1.5 ``StreamTokenizer`` in its default mode (init_min=0, init_max_silence=0, no
    STRICT_MIN_LENGTH, no DROP_TRAILING_SILENCE), restated from the published source with its own
    data-list bookkeeping (PARITY UNPINNED: auditok is not installed here).  Frames are booleans."""

    SILENCE, POSSIBLE_SILENCE, POSSIBLE_NOISE, NOISE = 0, 1, 2, 3

    def __init__(self, min_length, max_length, max_continuous_silence):
        self.min_length, self.max_length, self.max_continuous_silence = min_length, max_length, max_continuous_silence

    def tokenize(self, frames):
        self.state, self.data, self.tokens = self.SILENCE, [], []
        self.silence_length, self.start_frame, self.contiguous = 0, 0, False
        self.current = -1
        for frame in frames:
            self.current += 1
            self._process(bool(frame))
        # _post_process
        if self.state in (self.NOISE, self.POSSIBLE_SILENCE):
            if len(self.data) > 0 and len(self.data) > self.silence_length:
                self._end_of_detection()
        return self.tokens

    def _process(self, valid):
        if self.state == self.SILENCE:
            if valid:
                self.silence_length = 0
                self.start_frame = self.current
                self.data.append(valid)
                self.state = self.NOISE  # init_min == 0
                if len(self.data) >= self.max_length:
                    self._end_of_detection(True)
        elif self.state == self.NOISE:
            if valid:
                self.data.append(valid)
                if len(self.data) >= self.max_length:
                    self._end_of_detection(True)
            elif self.max_continuous_silence <= 0:
                self._end_of_detection()
                self.state = self.SILENCE
            else:
                self.silence_length = 1
                self.data.append(valid)
                self.state = self.POSSIBLE_SILENCE
                if len(self.data) == self.max_length:
                    self._end_of_detection(True)
        elif self.state == self.POSSIBLE_SILENCE:
            if valid:
                self.data.append(valid)
                self.silence_length = 0
                self.state = self.NOISE
                if len(self.data) >= self.max_length:
                    self._end_of_detection(True)
            elif self.silence_length >= self.max_continuous_silence:
                if self.silence_length < len(self.data):
                    self._end_of_detection()
                else:
                    self.data = []
                self.state = self.SILENCE
                self.silence_length = 0
            else:
                self.data.append(valid)
                self.silence_length += 1
                if len(self.data) >= self.max_length:
                    self._end_of_detection(True)

    def _end_of_detection(self, truncated=False):
        if len(self.data) >= self.min_length or (len(self.data) > 0 and self.contiguous):
            self.tokens.append((self.start_frame, self.start_frame + len(self.data) - 1))
            if truncated:
                self.start_frame = self.current + 1
                self.contiguous = True
            else:
                self.contiguous = False
        else:
            self.contiguous = False
        self.data = []


def tokenize_chunk(valid, non_speech_label=0.0, sample_rate=100):
    """One detector call of the reference's auditok wiring (speech_transformers.py:125-131, 143-150):
    tokens -> markers (assigned in order) -> clip(cumsum[:-1], 0, 1)."""
    tok = _Tokenizer(0.2 * sample_rate, int(5 * sample_rate), 0.25 * sample_rate)
    media_bstring = np.zeros(len(valid) + 1)
    for start, end in tok.tokenize(valid):
        media_bstring[start] = 1.0
        media_bstring[end + 1] = non_speech_label - 1.0
    return np.clip(np.cumsum(media_bstring)[:-1], 0.0, 1.0)


def tokenize_chunk_scan(valid, non_speech_label=0.0, min_length=20, max_length=500, max_continuous_silence=25):
    """numpy model of the PARALLEL formulation of the same smoothing (the island rule; the form k_vad_tokenize_scan had
    in rounds 4-6: per-frame index arrays, every step a scan -- tokenize_chunk_words() below models the kernel as it is
    now) -- test infrastructure like fft_model.py; tests/test_oracle_vad.py checks it against the state machine above.
    Requires max_length >= min_length.

    The tokenizer's state (and its silence counter) depends on the validity runs alone, so the frames fall into
    islands: an island starts at a valid frame that follows more than max_continuous_silence invalid ones (or no
    valid frame at all) and ends max_continuous_silence frames behind its last valid frame (at the end of the chunk
    if no longer gap follows).  Truncation cuts an island into pieces of max_length frames; full pieces are always
    delivered, the remainder r iff  max_continuous_silence < r  (island ended by a gap)  resp.  r > trailing silence
    (end of chunk)  and  r >= min_length or the tokenizer's `contiguous` flag is set: after a cut, or -- first piece --
    when the previous island ended with a cut followed by at most max_continuous_silence frames."""
    v = np.asarray(valid, bool)
    n = v.size
    if n == 0:
        return np.zeros(0)
    mn, mx, msil = int(min_length), int(max_length), int(max_continuous_silence)
    assert mx >= mn
    idx = np.arange(n)
    lastv = np.maximum.accumulate(np.where(v, idx, -1))                       # scan 1: last valid index <= i
    in_isl = v | ((lastv >= 0) & (idx - lastv <= max(msil, 0)))
    prev_in = np.concatenate([[False], in_isl[:-1]])
    isl = np.where(in_isl, np.maximum.accumulate(np.where(v & ~prev_in, idx, -1)), -1)   # scan 2: island start
    first_out = np.minimum.accumulate(np.where(~in_isl, idx, n)[::-1])[::-1]
    nxt = np.concatenate([first_out[1:], [n]])                                # scan 3: first outside index > i

    def c_in(s):
        if msil <= 0 or s == 0 or lastv[s - 1] < 0:
            return False
        lp = lastv[s - 1]
        lenp = lp + msil - isl[lp] + 1
        return lenp // mx >= 1 and lenp % mx <= msil

    def delivered(i0):
        s, e_isl = isl[i0], nxt[i0] - 1
        j = (i0 - s) // mx
        r = min(i0 + mx - 1, e_isl) - i0 + 1
        if r == mx:
            return True
        ok_len = r >= mn or (r > 0 and (j >= 1 or c_in(s)))
        if e_isl + 1 < n:
            return ok_len if msil <= 0 else (msil < r and ok_len)
        return r > 0 and r > e_isl - lastv[e_isl] and ok_len

    m = np.zeros(n)
    for i in range(n):
        if isl[i] >= 0 and (i - isl[i]) % mx == 0 and delivered(i):
            m[i] = 1.0
        elif i >= 1 and isl[i - 1] >= 0:
            off = (i - 1 - isl[i - 1]) % mx
            if (off == mx - 1 or i == nxt[i - 1]) and delivered(i - 1 - off):
                m[i] = non_speech_label - 1.0
    return np.clip(np.cumsum(m), 0.0, 1.0)                                    # scan 4


def tokenize_chunk_words(valid, non_speech_label=0.0, min_length=20, max_length=500, max_continuous_silence=25):
    """Model of k_vad_tokenize_scan as of round 6 (test infrastructure): the island formulation of
    tokenize_chunk_scan() above WITHOUT per-frame index arrays.  The validity flags are 64-bit words V; the island
    starts are a second bit array S (a valid frame whose predecessor is outside every island); three arrays with one
    entry per WORD (last valid frame / last start in front of the word, first start behind it) turn "last valid frame
    <= i", "start of the island of i" and "first frame behind the island of i" into one or two word reads.  The
    markers are then written island by island (one thread per half word of S on the device), piece by piece."""
    v = np.asarray(valid, bool)
    n = v.size
    if n == 0:
        return np.zeros(0)
    mn, mx, msil = int(min_length), int(max_length), int(max_continuous_silence)
    assert mx >= mn
    ms = max(msil, 0)
    W = (n + 63) // 64
    full = (1 << 64) - 1

    def words(bits):
        out = [0] * W
        for i in np.nonzero(bits)[0]:
            out[int(i) >> 6] |= 1 << (int(i) & 63)
        return out

    def last_in_front(words_):  # entry w: highest set bit in the words < w, -1 if none
        out, cur = [], -1
        for w in range(W):
            out.append(cur)
            if words_[w]:
                cur = 64 * w + words_[w].bit_length() - 1
        return out

    def first_behind(words_):   # entry w: lowest set bit in the words > w, -1 if none
        out, cur = [0] * W, -1
        for w in range(W - 1, -1, -1):
            out[w] = cur
            if words_[w]:
                cur = 64 * w + (words_[w] & -words_[w]).bit_length() - 1
        return out

    V = words(v)
    PL = last_in_front(V)

    def last_le(words_, front, i):
        if i < 0:
            return -1
        w = i >> 6
        m = words_[w] & (full >> (63 - (i & 63)))
        return 64 * w + m.bit_length() - 1 if m else front[w]

    def lastv(i):
        return last_le(V, PL, i)

    start = np.zeros(n, bool)
    for i in range(n):
        if v[i]:
            lv = lastv(i - 1)
            start[i] = lv < 0 or (i - 1 - lv) > ms
    S = words(start)
    SL, NS = last_in_front(S), first_behind(S)

    def nxt_of(i):  # i inside an island: first frame behind it
        w, b = i >> 6, i & 63
        m = S[w] & ((full << (b + 1)) & full)
        ns = 64 * w + (m & -m).bit_length() - 1 if m else NS[w]
        lv = lastv((ns if ns >= 0 else n) - 1)
        return min(n, lv + ms + 1)

    def c_in(s):
        if msil <= 0 or s == 0:
            return False
        lp = lastv(s - 1)
        if lp < 0:
            return False
        lenp = lp + msil - last_le(S, SL, lp) + 1
        return lenp // mx >= 1 and lenp % mx <= msil

    code = np.zeros(n, np.int8)
    for s in np.nonzero(start)[0]:
        s = int(s)
        e_isl = nxt_of(s) - 1
        i0, j = s, 0
        while i0 <= e_isl:
            e = min(i0 + mx - 1, e_isl)
            r = e - i0 + 1
            if r == mx:
                d = True
            else:
                ok_len = r >= mn or (r > 0 and (j >= 1 or c_in(s)))
                if e_isl + 1 < n:
                    d = ok_len if msil <= 0 else (msil < r and ok_len)
                else:
                    d = r > 0 and r > e_isl - lastv(e_isl) and ok_len
            if d:
                if e + 1 < n:
                    code[e + 1] = -1
                code[i0] = 1   # (a start wins over the end marker of the piece in front: assigned later)
            i0 += mx
            j += 1
    m = np.where(code > 0, 1.0, np.where(code < 0, non_speech_label - 1.0, 0.0))
    return np.clip(np.cumsum(m), 0.0, 1.0)


def tokenize(valid, non_speech_label=0.0, chunk_frames=10000, sample_rate=100):
    """Chunk loop: the reference builds the tokenizer once but every detector call starts a fresh
    tokenize() pass over its own 100 s buffer."""
    valid = np.asarray(valid)
    return np.concatenate([tokenize_chunk(valid[o:o + chunk_frames], non_speech_label, sample_rate)
                           for o in range(0, len(valid), chunk_frames)]) if len(valid) else np.zeros(0)
