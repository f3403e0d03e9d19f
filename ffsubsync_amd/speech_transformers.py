"""GPU side of ``ffsubsync.speech_transformers``: the per-frame voice-activity sweep.

What is mirrored from the reference (ffsubsync/speech_transformers.py):
  * the detector-factory seam ``_make_<name>_detector(sample_rate, frame_rate, non_speech_label)
    -> Callable[[bytes | uint8 ndarray], float64 ndarray]`` (:101, :155) -- here
    :func:`_make_energy_detector`, whose frame rule is the AudioEnergyValidator energy test
    (10*log10(mean x^2) >= 50 dB, :124) evaluated on the GPU for every 10 ms frame;
  * the 100 s chunk loop of ``VideoSpeechTransformer._fit_using_audio`` (:683-753) --
    :class:`PCMSpeechTransformer`, fed by any binary stream (the ffmpeg pipe in production);
  * ``ComputeSpeechFrameBoundariesMixin`` (:299-317).

ffmpeg spawning, ffprobe, progress bars, webrtcvad/silero and the auditok token smoothing stay on
the host in the reference and are out of scope.
"""
import os
import subprocess
from datetime import timedelta
from typing import Callable, List, NamedTuple, Optional, Union

import numpy as np

from . import _native
from .constants import DEFAULT_ENERGY_THRESHOLD_DB
from .sklearn_shim import TransformerMixin

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


_FUSION_STRATEGIES = ("weighted", "intersection", "union")  # speech_transformers.py:253


def _make_fused_detector(sample_rate: int, frame_rate: int, non_speech_label: float,
                         fusion_strategy: str = "weighted") -> Callable[[bytes], np.ndarray]:
    """speech_transformers.py:256-296: combine two detectors frame by frame -- ``intersection``
    (minimum), ``union`` (maximum) or ``weighted`` (0.6 * silero + 0.4 * webrtc) -- after clipping
    both label vectors to their common length.  The two factories are looked up as module attributes
    at call time, the seam the reference's tests/test_vad_fused.py:11-18 patches."""
    if fusion_strategy not in _FUSION_STRATEGIES:
        raise ValueError(
            "unknown fused VAD strategy %r; choose one of %s" % (fusion_strategy, ", ".join(_FUSION_STRATEGIES))
        )
    import sys

    mod = sys.modules[__name__]
    webrtc_detector = mod._make_webrtcvad_detector(sample_rate, frame_rate, non_speech_label)
    silero_detector = mod._make_silero_detector(sample_rate, frame_rate, non_speech_label)

    def _detect(asegment) -> np.ndarray:
        webrtc_result = webrtc_detector(asegment)
        silero_result = silero_detector(asegment)
        n = min(len(webrtc_result), len(silero_result))
        webrtc_result, silero_result = webrtc_result[:n], silero_result[:n]
        if fusion_strategy == "intersection":
            return np.minimum(webrtc_result, silero_result)
        if fusion_strategy == "union":
            return np.maximum(webrtc_result, silero_result)
        return 0.6 * silero_result + 0.4 * webrtc_result

    return _detect


def detect_device(pcm_dev, sample_rate: int, frame_rate: int, non_speech_label: float,
                  energy_threshold_db: float = DEFAULT_ENERGY_THRESHOLD_DB):
    """Same sweep for PCM already resident in HBM (int16 CUDA tensor) -> float32 CUDA labels."""
    return _native.vad_energy(pcm_dev, frames_per_window(sample_rate, frame_rate), energy_threshold_db,
                              non_speech_label)


class ComputeSpeechFrameBoundariesMixin:
    """speech_transformers.py:299-317."""

    def __init__(self) -> None:
        self.start_frame_: Optional[int] = None
        self.end_frame_: Optional[int] = None

    @property
    def num_frames(self) -> Optional[int]:
        if self.start_frame_ is None or self.end_frame_ is None:
            return None
        return self.end_frame_ - self.start_frame_

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


class ProgressInfo(NamedTuple):
    """ffsubsync/speech_transformers.py:40-53: what ``progress_handler`` receives per PCM chunk."""

    processed_seconds: float
    total_seconds: Optional[float]

    @property
    def fraction(self) -> Optional[float]:
        if not self.total_seconds:
            return None
        return min(1.0, self.processed_seconds / self.total_seconds)


class VideoSpeechTransformer(PCMSpeechTransformer):
    """Same constructor and fitted attribute as the reference's ``VideoSpeechTransformer``
    (speech_transformers.py:320-351): ``fit(fname)`` decodes the reference's audio with an ffmpeg
    subprocess to s16le mono PCM on a pipe (:681-682) and runs the chunked VAD loop on the GPU.
    Only the decode command needed for that is built here; the reference's embedded-subtitle
    shortcut, ffprobe duration probing, remote-URL temp extraction and GUI/VLC progress plumbing are
    control plane and stay with the reference (a ``subs_then_*`` vad falls straight through to audio)."""

    def __init__(self, vad: str, sample_rate: int, frame_rate: int, non_speech_label: float,
                 start_seconds: int = 0, ffmpeg_path: Optional[str] = None, ref_stream: Optional[str] = None,
                 vlc_mode: bool = False, gui_mode: bool = False, max_duration_seconds: Optional[float] = None,
                 extract_audio_first: bool = False, progress_handler=None) -> None:
        super(VideoSpeechTransformer, self).__init__(vad, sample_rate, frame_rate, non_speech_label, None)
        self.start_seconds = start_seconds
        self.ffmpeg_path = ffmpeg_path
        self.ref_stream = ref_stream
        self.vlc_mode = vlc_mode
        self.gui_mode = gui_mode
        self.max_duration_seconds = max_duration_seconds
        self.extract_audio_first = extract_audio_first
        self._video_progress_handler = progress_handler

    def _decode_command(self, fname: str) -> List[str]:
        exe = "ffmpeg" if self.ffmpeg_path is None else os.path.join(self.ffmpeg_path, "ffmpeg")
        cmd = [exe]
        if self.start_seconds > 0:
            cmd += ["-ss", str(timedelta(seconds=self.start_seconds))]
        if self.max_duration_seconds is not None:
            cmd += ["-t", str(timedelta(seconds=self.max_duration_seconds))]
        cmd += ["-loglevel", "fatal", "-nostdin", "-i", fname]
        if self.ref_stream is not None and self.ref_stream.startswith("0:a:"):
            cmd += ["-map", self.ref_stream]
        cmd += ["-f", "s16le", "-ac", "1", "-acodec", "pcm_s16le", "-af", "aresample=async=1",
                "-ar", str(self.frame_rate), "-"]
        return cmd

    def fit(self, fname: str, *_) -> "VideoSpeechTransformer":
        total = self.max_duration_seconds
        if self._video_progress_handler is not None:
            handler = self._video_progress_handler
            self.progress_handler = lambda processed: handler(ProgressInfo(processed, total))
        process = subprocess.Popen(self._decode_command(fname), stdin=subprocess.DEVNULL, stdout=subprocess.PIPE)
        try:
            super(VideoSpeechTransformer, self).fit(process.stdout)
        finally:
            process.wait()
        return self


def _probe_duration(fname: str, ffmpeg_path: Optional[str] = None) -> float:
    """Container duration in seconds via an ffprobe subprocess (the reference uses ffmpeg-python's
    probe for the same number, speech_transformers.py:849-858)."""
    exe = "ffprobe" if ffmpeg_path is None else os.path.join(ffmpeg_path, "ffprobe")
    out = subprocess.check_output([exe, "-v", "error", "-show_entries", "format=duration", "-of",
                                   "default=noprint_wrappers=1:nokey=1", fname], stdin=subprocess.DEVNULL)
    return float(out.decode().strip())


class MultiSegmentVideoSpeechTransformer(TransformerMixin):
    """Sparse reference signal from a few sampled windows (speech_transformers.py:760-903): VAD runs on
    ``segment_count`` windows of ``segment_duration`` seconds spread evenly over the reference (each
    through its own :class:`VideoSpeechTransformer`, up to ``parallel_workers`` at a time -- every worker
    thread drives the GPU through its own handles) and the labels are scattered into an otherwise
    zero full-length vector, which the aligner consumes unchanged."""

    START_MARGIN_SECONDS: int = 30
    END_MARGIN_SECONDS: int = 60

    def __init__(self, vad: str, sample_rate: int, frame_rate: int, non_speech_label: float,
                 segment_count: int = 8, segment_duration: int = 60, skip_intro_outro: bool = False,
                 parallel_workers: int = 4, ffmpeg_path: Optional[str] = None, ref_stream: Optional[str] = None,
                 vlc_mode: bool = False, gui_mode: bool = False) -> None:
        self.vad = vad.split("subs_then_")[-1]  # sampling is audio-only (:795-797)
        self.sample_rate = sample_rate
        self.frame_rate = frame_rate
        self._non_speech_label = non_speech_label
        self.segment_count = segment_count
        self.segment_duration = segment_duration
        self.skip_intro_outro = skip_intro_outro
        self.parallel_workers = parallel_workers
        self.ffmpeg_path = ffmpeg_path
        self.ref_stream = ref_stream
        self.vlc_mode = vlc_mode
        self.gui_mode = gui_mode
        self.video_speech_results_: Optional[np.ndarray] = None

    def _segment_starts(self, total_duration: float) -> List[int]:
        """Start seconds of the sampled windows (:813-834): evenly spaced over the usable span, margins
        honoured when they leave room, clamped into range and de-duplicated."""
        window = self.segment_duration
        if total_duration <= window:
            return [0]
        first = float(self.START_MARGIN_SECONDS if self.skip_intro_outro else 0)
        last = total_duration - (self.END_MARGIN_SECONDS if self.skip_intro_outro else 0)
        if last - first < window:
            first, last = 0.0, total_duration
        span = last - first - window
        count = max(1, self.segment_count)
        if span <= 0 or count == 1:
            return [int(max(0.0, min(first, total_duration - window)))]
        picks = [int(round(first + i * span / (count - 1))) for i in range(count)]
        top = int(total_duration) - window
        return sorted({max(0, min(p, top)) for p in picks})

    def _extract_segment_speech(self, fname: str, start: int):
        seg = VideoSpeechTransformer(self.vad, self.sample_rate, self.frame_rate, self._non_speech_label,
                                     start_seconds=start, ffmpeg_path=self.ffmpeg_path, ref_stream=self.ref_stream,
                                     vlc_mode=self.vlc_mode, gui_mode=self.gui_mode,
                                     max_duration_seconds=self.segment_duration)
        seg.fit(fname)
        return start, seg.transform()

    def fit(self, fname: str, *_) -> "MultiSegmentVideoSpeechTransformer":
        from concurrent.futures import ThreadPoolExecutor, as_completed

        try:
            total_duration = float(_probe_duration(fname, self.ffmpeg_path))
        except Exception as e:
            raise ValueError("multi-segment sync needs the reference duration, but probing "
                             "'%s' failed: %s" % (fname, e))
        starts = self._segment_starts(total_duration)
        sparse = np.zeros(int(total_duration * self.sample_rate) + 2, dtype=float)
        with ThreadPoolExecutor(max_workers=max(1, min(self.parallel_workers, len(starts)))) as pool:
            pending = {pool.submit(self._extract_segment_speech, fname, s): s for s in starts}
            for fut in as_completed(pending):
                try:
                    start, labels = fut.result()
                except Exception:  # one bad window must not sink the sync (:878-886)
                    continue
                lo = int(start * self.sample_rate)
                hi = min(lo + len(labels), len(sparse))
                if hi > lo:
                    sparse[lo:hi] = labels[: hi - lo]
        if not np.any(sparse > 0):
            raise ValueError("Unable to detect speech in any sampled segment. "
                             "Perhaps try specifying a different stream / track, or a different vad.")
        self.video_speech_results_ = sparse
        return self

    def transform(self, *_) -> np.ndarray:
        return self.video_speech_results_


def serialize_speech(fname: str, speech) -> None:
    """``--serialize-speech``: np.savez_compressed(<ref>.npz, speech=...) (ffsubsync/ffsubsync.py:639-644)."""
    np.savez_compressed(fname, speech=np.asarray(speech))


class DeserializeSpeechTransformer(TransformerMixin):
    """speech_transformers.py:987-1009: a ``.npy`` / ``.npz`` (key ``speech``) reference activity
    vector; every sample below 1.0 becomes ``non_speech_label``.  The bulk on-disk format for batch
    jobs that feed precomputed reference vectors straight to the device."""

    def __init__(self, non_speech_label: float) -> None:
        super(DeserializeSpeechTransformer, self).__init__()
        self._non_speech_label: float = non_speech_label
        self.deserialized_speech_results_: Optional[np.ndarray] = None

    def fit(self, fname, *_) -> "DeserializeSpeechTransformer":
        speech = np.load(fname)
        if hasattr(speech, "files"):
            if "speech" in speech.files:
                speech = speech["speech"]
            else:
                raise ValueError(
                    'could not find "speech" array in '
                    "serialized file; only contains: %s" % speech.files
                )
        speech = np.array(speech, dtype=float)
        speech[speech < 1.0] = self._non_speech_label
        self.deserialized_speech_results_ = speech
        return self

    def transform(self, *_) -> np.ndarray:
        assert self.deserialized_speech_results_ is not None
        return self.deserialized_speech_results_
