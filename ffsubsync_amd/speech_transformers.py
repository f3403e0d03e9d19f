"""GPU side of ``ffsubsync.speech_transformers``: the per-frame voice-activity sweep.

What is mirrored from the reference (ffsubsync/speech_transformers.py):
  * the detector-factory seam ``_make_<name>_detector(sample_rate, frame_rate, non_speech_label)
    -> Callable[[bytes | uint8 ndarray], float64 ndarray]`` (:101, :155) -- here
    :func:`_make_energy_detector`, whose frame rule is the AudioEnergyValidator energy test
    (10*log10(mean x^2) >= 50 dB, :124) evaluated on the GPU for every 10 ms frame;
  * the 100 s chunk loop of ``VideoSpeechTransformer._fit_using_audio`` (:683-753) --
    :class:`PCMSpeechTransformer`, fed by any binary stream (the ffmpeg pipe in production);
  * ``ComputeSpeechFrameBoundariesMixin`` (:299-317).

  * the sparse-reference scatter of ``MultiSegmentVideoSpeechTransformer.fit`` (:871-890) --
    :func:`assemble_sparse_reference`; ``DeserializeSpeechTransformer`` (:987-1009) and its bulk
    counterpart :func:`load_speech_batch` (``.npz`` files straight into bit-packed HBM vectors).

ffmpeg / ffprobe spawning, the embedded-subtitle shortcut, progress bars and the thread pool of the
multi-segment path are control plane and stay with the reference's own classes:
:func:`install_detectors` puts the GPU detector behind them through the reference's factory seam.
webrtcvad / silero arithmetic lives in absent third-party wheels (seams only).
"""
from typing import Callable, List, Optional, Sequence, Union

import numpy as np

from . import _native
from .constants import DEFAULT_ENERGY_THRESHOLD_DB
from .sklearn_shim import TransformerMixin, reference_module

BYTES_PER_SAMPLE = 2  # s16le (speech_transformers.py:122, :160, :683)
WINDOWS_PER_BUFFER = 10000  # speech_transformers.py:685


def frames_per_window(sample_rate: int, frame_rate: int) -> int:
    """PCM samples per activity frame: int(1/sample_rate * frame_rate + 0.5) (speech_transformers.py:161-162)."""
    return int((1.0 / sample_rate) * frame_rate + 0.5)


def _as_int16_bytes(asegment) -> np.ndarray:
    buf = np.frombuffer(asegment, dtype=np.uint8) if isinstance(asegment, (bytes, bytearray, memoryview)) else \
        np.ascontiguousarray(asegment).view(np.uint8).ravel()
    return buf[: buf.size // BYTES_PER_SAMPLE * BYTES_PER_SAMPLE].view("<i2")


def _make_energy_detector(sample_rate: int, frame_rate: int, non_speech_label: float,
                          energy_threshold_db: float = DEFAULT_ENERGY_THRESHOLD_DB
                          ) -> Callable[[Union[bytes, np.ndarray]], np.ndarray]:
    """Detector closure with the reference's factory signature; one call = one PCM chunk."""
    frame_len = frames_per_window(sample_rate, frame_rate)

    def _detect(asegment) -> np.ndarray:
        torch = _native.require_gpu()
        pcm = _as_int16_bytes(asegment)
        if pcm.size == 0:
            return np.zeros(0, dtype=float)
        dev = torch.from_numpy(pcm.copy()).cuda()
        labels = _native.vad_energy(dev, frame_len, energy_threshold_db, non_speech_label)
        return labels.cpu().numpy().astype(float)

    return _detect


def _make_auditok_detector(sample_rate: int, frame_rate: int, non_speech_label: float,
                           energy_threshold_db: float = DEFAULT_ENERGY_THRESHOLD_DB
                           ) -> Callable[[Union[bytes, np.ndarray]], np.ndarray]:
    """GPU counterpart of the reference's auditok detector (speech_transformers.py:101-152): the
    frame-energy test (threshold 50 dB, :124) followed by the StreamTokenizer smoothing with the
    reference's parameters (min 0.2 s, max 5 s, 0.25 s of tolerated silence, :125-131) and its marker /
    cumsum rasterisation (:143-150), one tokenizer pass per call as in the reference.  Restated from
    auditok 0.1.5's published source -- parity unpinned (see oracle/vad_oracle.py)."""
    frame_len = frames_per_window(sample_rate, frame_rate)

    def _detect(asegment) -> np.ndarray:
        torch = _native.require_gpu()
        pcm = _as_int16_bytes(asegment)
        if pcm.size == 0:
            return np.zeros(0, dtype=float)
        dev = torch.from_numpy(pcm.copy()).cuda()
        valid = _native.vad_energy(dev, frame_len, energy_threshold_db, 0.0)
        labels = _native.vad_tokenize(valid, max(int(valid.numel()), 1), 0.2 * sample_rate, int(5 * sample_rate),
                                      0.25 * sample_rate, non_speech_label)
        return labels.cpu().numpy().astype(float)

    return _detect


def _make_webrtcvad_detector(sample_rate: int, frame_rate: int, non_speech_label: float):
    """Seam only (speech_transformers.py:155-183): WebRTC's GMM lives in the third-party webrtcvad
    wheel; when the reference package is importable its factory is used unchanged."""
    try:
        from ffsubsync.speech_transformers import _make_webrtcvad_detector as ref_factory  # type: ignore
    except Exception as e:  # pragma: no cover - depends on the environment
        raise ImportError("webrtcvad detector needs ffsubsync + webrtcvad installed: %s" % e)
    return ref_factory(sample_rate, frame_rate, non_speech_label)


def _make_silero_detector(sample_rate: int, frame_rate: int, non_speech_label: float):
    """Seam only (speech_transformers.py:186-236): needs torch.hub + the silero weights (network)."""
    try:
        from ffsubsync.speech_transformers import _make_silero_detector as ref_factory  # type: ignore
    except Exception as e:  # pragma: no cover - depends on the environment
        raise ImportError("silero detector needs ffsubsync + torch.hub access: %s" % e)
    return ref_factory(sample_rate, frame_rate, non_speech_label)


_FUSION_RULES = {  # speech_transformers.py:253-296: how two label vectors of equal length combine
    "weighted": lambda webrtc, silero: 0.6 * silero + 0.4 * webrtc,
    "intersection": np.minimum,
    "union": np.maximum,
}


def _make_fused_detector(sample_rate: int, frame_rate: int, non_speech_label: float,
                         fusion_strategy: str = "weighted") -> Callable[[bytes], np.ndarray]:
    """speech_transformers.py:256-296 (webrtc and silero labels combined frame by frame).  Both detectors are
    third-party CPU arithmetic (seams).  ONE implementation whether or not ffsubsync is installed: strategies,
    error text, clipping to the common length, and the two factories looked up as attributes of THIS module at call time
    (the patch point, like the seam tests/test_vad_fused.py:11-18 patches in the reference) -- they in turn hand over to
    the reference's CPU detectors when those are importable."""
    rule = _FUSION_RULES.get(fusion_strategy)
    if rule is None:
        raise ValueError("unknown fused VAD strategy %r; choose one of weighted, intersection, union" % (fusion_strategy,))
    import sys

    here = sys.modules[__name__]
    parts = [factory(sample_rate, frame_rate, non_speech_label)
             for factory in (here._make_webrtcvad_detector, here._make_silero_detector)]

    def _detect(asegment) -> np.ndarray:
        webrtc, silero = (np.asarray(part(asegment)) for part in parts)
        n = min(len(webrtc), len(silero))
        return rule(webrtc[:n], silero[:n])

    return _detect


def detect_device(pcm_dev, sample_rate: int, frame_rate: int, non_speech_label: float,
                  energy_threshold_db: float = DEFAULT_ENERGY_THRESHOLD_DB):
    """Same sweep for PCM already resident in HBM (int16 CUDA tensor) -> float32 CUDA labels."""
    return _native.vad_energy(pcm_dev, frames_per_window(sample_rate, frame_rate), energy_threshold_db,
                              non_speech_label)


def detect_pinned_stream(pcm_host, sample_rate: int, frame_rate: int, non_speech_label: float,
                         energy_threshold_db: float = DEFAULT_ENERGY_THRESHOLD_DB, staging=None, packed: bool = False):
    """The chunk loop of ``_fit_using_audio`` (speech_transformers.py:683-753) for decoded s16le PCM that
    sits in pinned host memory (int16 CPU tensor): every 100 s buffer (:683-685) is copied to one of two
    HBM staging buffers on a copy stream while the frame-energy sweep of the previous buffer runs on the
    caller's stream -- PCIe transfer and VAD overlap, the labels never leave the GPU.
    Returns the float32 CUDA label vector, or with ``packed`` the bit-packed ``DeviceRaster`` the aligner reads
    (``ffs_vad_energy_bits``: no fp32 label round trip, no packing pass; ``non_speech_label`` becomes the raster's
    low level).  ``staging`` = (buffers, copy_stream) to reuse across files."""
    torch = _native.require_gpu()
    frame_len = frames_per_window(sample_rate, frame_rate)
    chunk = frame_len * WINDOWS_PER_BUFFER  # samples per buffer; a multiple of the frame length (and of 8 frames)
    n = int(pcm_host.numel())
    n_frames = (n + frame_len - 1) // frame_len
    if packed:
        out = torch.zeros((n_frames + 31) // 32, dtype=torch.int32, device="cuda")
    else:
        out = torch.empty(n_frames, dtype=torch.float32, device="cuda")
    main = torch.cuda.current_stream()
    if n:
        own = staging is None
        if own:
            staging = ([torch.empty(chunk, dtype=torch.int16, device="cuda") for _ in range(2)], torch.cuda.Stream())
        bufs, copy_stream = staging
        # the staging blocks may still carry work of the caller's stream (fresh from the caching allocator, or the
        # previous file's last sweep): the first copies wait for it
        copy_stream.wait_stream(main)
        if own:
            for b in bufs:
                b.record_stream(copy_stream)
        filled = [torch.cuda.Event() for _ in range(2)]
        drained = [None, None]
        lib = _native.load()
        for i, o in enumerate(range(0, n, chunk)):
            k = i % 2
            m = min(chunk, n - o)
            with torch.cuda.stream(copy_stream):
                if drained[k] is not None:
                    copy_stream.wait_event(drained[k])  # the sweep that last read this staging buffer is done
                bufs[k][:m].copy_(pcm_host[o:o + m], non_blocking=True)
                filled[k].record(copy_stream)
            main.wait_event(filled[k])
            f0 = o // frame_len
            if packed:
                _native.check(lib.ffs_vad_energy_bits(bufs[k].data_ptr(), m, frame_len, float(energy_threshold_db),
                                                      out.data_ptr() + f0 // 8, main.cuda_stream))
            else:
                _native.check(lib.ffs_vad_energy(bufs[k].data_ptr(), m, frame_len, float(energy_threshold_db),
                                                 float(non_speech_label), out[f0:].data_ptr(), main.cuda_stream))
            drained[k] = torch.cuda.Event()
            drained[k].record(main)
    if packed:
        from .subtitle_raster import DeviceRaster

        return DeviceRaster(out, float(non_speech_label), 1.0, n_frames)
    return out


class ComputeSpeechFrameBoundariesMixin:
    """speech_transformers.py:299-317 with the scan on the device: first / last frame above 0.5 (``start_frame_``,
    ``end_frame_``; both stay None when nothing is) and their distance ``num_frames``."""

    start_frame_: Optional[int] = None
    end_frame_: Optional[int] = None

    @property
    def num_frames(self) -> Optional[int]:
        known = self.start_frame_ is not None and self.end_frame_ is not None
        return self.end_frame_ - self.start_frame_ if known else None

    def fit_boundaries(self, speech_frames) -> "ComputeSpeechFrameBoundariesMixin":
        torch = _native.require_gpu()
        if hasattr(speech_frames, "frames_float"):  # a DeviceRaster
            frames = speech_frames.frames_float()
        elif isinstance(speech_frames, np.ndarray) or not hasattr(speech_frames, "is_cuda"):
            frames = torch.from_numpy(np.asarray(speech_frames, dtype=np.float32)).cuda()
        else:
            frames = speech_frames.to(torch.float32)
        lo, hi = _native.speech_bounds(frames)
        if hi is not None:
            self.start_frame_, self.end_frame_ = lo, hi
        return self


class PCMSpeechTransformer(TransformerMixin):
    """The VAD leg of ``VideoSpeechTransformer`` (speech_transformers.py:320-351, 635-757) for an
    already-decoded s16le mono stream: read 100 s chunks, run the detector on each, concatenate.

    ``fit(source)`` accepts a binary file object (e.g. ``subprocess.Popen(...).stdout``), bytes or
    an int16 array.  The fitted ``video_speech_results_`` is a float64 ndarray as in the reference.
    """

    def __init__(self, vad: str = "energy", sample_rate: int = 100, frame_rate: int = 48000,
                 non_speech_label: float = 0.0, progress_handler=None) -> None:
        self.vad = vad
        self.sample_rate = sample_rate
        self.frame_rate = frame_rate
        self._non_speech_label = non_speech_label
        self.progress_handler = progress_handler
        self.video_speech_results_: Optional[np.ndarray] = None

    def _make_detector(self):
        if "fused" in self.vad:  # e.g. "fused" or "fused:intersection" (speech_transformers.py:655-665)
            strategy = self.vad.split(":", 1)[1] if ":" in self.vad else "weighted"
            return _make_fused_detector(self.sample_rate, self.frame_rate, self._non_speech_label, strategy)
        if "webrtc" in self.vad:
            return _make_webrtcvad_detector(self.sample_rate, self.frame_rate, self._non_speech_label)
        if "auditok" in self.vad:
            return _make_auditok_detector(self.sample_rate, self.frame_rate, self._non_speech_label)
        if "energy" in self.vad:
            return _make_energy_detector(self.sample_rate, self.frame_rate, self._non_speech_label)
        if "silero" in self.vad:
            return _make_silero_detector(self.sample_rate, self.frame_rate, self._non_speech_label)
        raise ValueError("unknown vad: %s" % self.vad)  # speech_transformers.py:679

    def _chunks(self, source, chunk_bytes):
        if isinstance(source, (bytes, bytearray, memoryview, np.ndarray)):
            raw = _as_int16_bytes(source).view(np.uint8)
            for o in range(0, raw.size, chunk_bytes):
                yield raw[o:o + chunk_bytes]
        else:
            while True:
                blob = source.read(chunk_bytes)
                if not blob:
                    return
                yield np.frombuffer(blob, np.uint8)

    def _progress(self, processed):
        if self.progress_handler is not None:
            try:
                self.progress_handler(processed)
            except Exception:  # a host callback must never break syncing (:731-734)
                pass

    def _fit_energy_pipelined(self, source, chunk_bytes) -> List[np.ndarray]:
        """Energy VAD with ingest overlap: chunks go through two pinned staging buffers with
        asynchronous H2D copies, the sweep of chunk i runs while chunk i+1 is being read from the
        pipe, and the labels come back in one transfer at the end."""
        torch = _native.require_gpu()
        frame_len = frames_per_window(self.sample_rate, self.frame_rate)
        pinned = [torch.empty(chunk_bytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        device = [torch.empty(chunk_bytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
        copied = [None, None]
        labels = []
        processed = 0.0
        for i, in_bytes in enumerate(self._chunks(source, chunk_bytes)):
            k = i % 2
            n = in_bytes.size // BYTES_PER_SAMPLE * BYTES_PER_SAMPLE
            processed += len(in_bytes) / float(BYTES_PER_SAMPLE) / self.frame_rate
            self._progress(processed)
            if n == 0:
                labels.append(torch.empty(0, dtype=torch.float32, device="cuda"))
                continue
            if copied[k] is not None:
                copied[k].synchronize()  # the copy that last used this staging buffer is done
            pinned[k][:n].numpy()[:] = in_bytes[:n]
            device[k][:n].copy_(pinned[k][:n], non_blocking=True)
            copied[k] = torch.cuda.Event()
            copied[k].record()
            pcm = device[k][:n].view(torch.int16)
            labels.append(_native.vad_energy(pcm, frame_len, DEFAULT_ENERGY_THRESHOLD_DB, self._non_speech_label))
        if not labels:
            return []
        return [torch.cat(labels).cpu().numpy().astype(float)]

    def fit(self, source, *_) -> "PCMSpeechTransformer":
        bytes_per_window = BYTES_PER_SAMPLE * self.frame_rate // self.sample_rate  # :683-684
        chunk_bytes = bytes_per_window * WINDOWS_PER_BUFFER
        if "energy" in self.vad and "fused" not in self.vad and "auditok" not in self.vad:
            media_bstring = self._fit_energy_pipelined(source, chunk_bytes)
        else:
            detector = self._make_detector()
            media_bstring = []
            processed = 0.0
            for in_bytes in self._chunks(source, chunk_bytes):
                processed += len(in_bytes) / float(BYTES_PER_SAMPLE) / self.frame_rate
                self._progress(processed)
                media_bstring.append(detector(in_bytes))
        if len(media_bstring) == 0:
            raise ValueError(
                "Unable to detect speech. "
                "Perhaps try specifying a different stream / track, or a different vad."
            )
        self.video_speech_results_ = np.concatenate(media_bstring)
        return self

    def transform(self, *_) -> np.ndarray:
        return self.video_speech_results_


def install_detectors(ref_speech_transformers=None) -> None:
    """The VAD seam of the reference: ``VideoSpeechTransformer._fit_using_audio`` looks the
    detector factory up as a module attribute by substring of ``--vad`` (speech_transformers.py:655-679),
    so replacing ``ffsubsync.speech_transformers._make_auditok_detector`` puts the GPU frame-energy sweep +
    token smoothing behind the reference's own class -- its ffmpeg pipe, chunk loop, embedded-subtitle
    shortcut (``subs_then_*``, :609-633), progress reporting and ``MultiSegmentVideoSpeechTransformer``
    thread pool stay exactly as they are.  (webrtc / silero keep the reference's CPU implementations.)"""
    if ref_speech_transformers is None:
        import ffsubsync.speech_transformers as ref_speech_transformers  # type: ignore
    ref_speech_transformers._make_auditok_detector = _make_auditok_detector


def assemble_sparse_reference(segment_labels, starts_seconds, total_duration: float, sample_rate: int):
    """Device-side scatter of ``MultiSegmentVideoSpeechTransformer.fit`` (speech_transformers.py:871-890):
    ``sparse = zeros(int(total_duration*sample_rate) + 2)`` and, for every sampled window,
    ``sparse[begin:end] = labels[:end-begin]`` with ``begin = int(start*sample_rate)`` clipped at the end.
    ``segment_labels`` are float32 CUDA label vectors (e.g. from :func:`detect_device`); windows are
    applied in the order given (later ones win where windows overlap, as in the reference's loop).
    Returns the float32 CUDA vector; raises the reference's error when no frame is speech."""
    torch = _native.require_gpu()
    out_len = int(total_duration * sample_rate) + 2
    lens = np.array([int(t.numel()) for t in segment_labels], dtype=np.int64)
    src_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64) if lens.size else lens
    dst = np.array([int(s * sample_rate) for s in starts_seconds], dtype=np.int64)
    if lens.size:
        labels = torch.cat([t.to(torch.float32).reshape(-1) for t in segment_labels])
    else:
        labels = torch.empty(0, dtype=torch.float32, device="cuda")
    # overlapping windows must be applied in order: one launch per run of non-overlapping windows
    sparse = None
    order = list(range(lens.size))
    while order:
        run, rest, covered = [], [], []
        for i in order:
            lo_, hi_ = int(dst[i]), int(dst[i] + lens[i])
            if rest or any(lo_ < b and a < hi_ for a, b in covered):
                rest.append(i)
            else:
                run.append(i)
                covered.append((lo_, hi_))
        part = _native.scatter_segments(labels, src_off[run], dst[run], lens[run], out_len)
        if sparse is None:
            sparse = part
        else:  # later windows overwrite earlier ones where they overlap
            for i in run:
                a, b = int(dst[i]), min(int(dst[i] + lens[i]), out_len)
                if b > a:
                    sparse[a:b] = part[a:b]
        order = rest
    if sparse is None:
        sparse = torch.zeros(out_len, dtype=torch.float32, device="cuda")
    if not bool((sparse > 0).any()):
        raise ValueError("Unable to detect speech in any sampled segment. "
                         "Perhaps try specifying a different stream / track, or a different vad.")
    return sparse


def serialize_speech(fname: str, speech) -> None:
    """``--serialize-speech``: np.savez_compressed(<ref>.npz, speech=...) (ffsubsync/ffsubsync.py:639-644)."""
    np.savez_compressed(fname, speech=np.asarray(speech))


class _DeserializeStandIn(TransformerMixin):
    """speech_transformers.py:987-1009 for boxes without ffsubsync: a ``.npy`` / ``.npz`` (key ``speech``) activity
    vector; every sample below 1.0 becomes ``non_speech_label``."""

    def __init__(self, non_speech_label: float) -> None:
        self._non_speech_label, self.deserialized_speech_results_ = non_speech_label, None

    def fit(self, fname, *_):
        data = np.load(fname)
        if hasattr(data, "files"):  # an .npz archive
            if "speech" not in data.files:
                raise ValueError('could not find "speech" array in serialized file; only contains: %s' % data.files)
            data = data["speech"]
        labels = np.array(data, dtype=float)
        self.deserialized_speech_results_ = np.where(labels < 1.0, self._non_speech_label, labels)
        return self

    def transform(self, *_) -> np.ndarray:
        assert self.deserialized_speech_results_ is not None
        return self.deserialized_speech_results_


def __getattr__(name: str):
    """``DeserializeSpeechTransformer`` (the bulk on-disk format's reader, speech_transformers.py:987-1009) is the
    reference's own class when ffsubsync is importable, resolved on first use rather than at import."""
    if name == "DeserializeSpeechTransformer":
        ref = reference_module("speech_transformers")
        cls = ref.DeserializeSpeechTransformer if ref is not None else _DeserializeStandIn
        globals()[name] = cls
        return cls
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


def load_speech_batch(fnames: Sequence[str], non_speech_label: float = 0.0):
    """Bulk ``.npz`` / ``.npy`` -> HBM feeder: every file is read and thresholded exactly as
    ``DeserializeSpeechTransformer.fit`` does (``speech[speech < 1.0] = non_speech_label``,
    speech_transformers.py:993-1005 -- so every vector is two-level: {non_speech_label, >= 1.0 values}),
    bit-packed on the host and uploaded in ONE transfer; returns one bit-packed
    ``subtitle_raster.DeviceRaster`` per file, which the aligners and ``batch.pack_pairs`` take as is.
    A file whose speech values are not all equal (anything above 1.0) is returned as a float64 array instead."""
    from .subtitle_raster import DeviceRaster

    torch = _native.require_gpu()
    loaded, chunks, total = [], [], 0
    for fname in fnames:
        speech = __getattr__("DeserializeSpeechTransformer")(non_speech_label).fit(fname).transform()
        speech = np.asarray(speech, dtype=float).ravel()
        hi = float(speech.max()) if speech.size else 1.0
        is_hi = speech >= 1.0
        if speech.size and not np.all(speech[is_hi] == hi):
            loaded.append((speech, None, 0))
            continue
        packed = np.packbits(is_hi, bitorder="little")
        loaded.append((None, (float(non_speech_label), hi if is_hi.any() else 1.0, speech.size), total))
        chunks.append((total, packed))
        total += (packed.size + 63) // 64 * 64
    host = np.zeros(max(total, 64), dtype=np.uint8)
    for o, c in chunks:
        host[o:o + c.size] = c
    dev = torch.from_numpy(host).cuda().view(torch.int32)
    out = []
    for arr, meta, o in loaded:
        if meta is None:
            out.append(arr)
        else:
            lo, hi, n = meta
            out.append(DeviceRaster(dev[o // 4: o // 4 + (n + 31) // 32], lo, hi, n))
    return out
