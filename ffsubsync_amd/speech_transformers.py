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
from typing import Callable, List, Optional, Union

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
        if "energy" in self.vad or "auditok" in self.vad:
            return _make_energy_detector(self.sample_rate, self.frame_rate, self._non_speech_label)
        raise ValueError("unknown vad: %s" % self.vad)  # speech_transformers.py:679

    def fit(self, source, *_) -> "PCMSpeechTransformer":
        detector = self._make_detector()
        bytes_per_window = BYTES_PER_SAMPLE * self.frame_rate // self.sample_rate  # :683-684
        chunk_bytes = bytes_per_window * WINDOWS_PER_BUFFER
        if isinstance(source, (bytes, bytearray, memoryview, np.ndarray)):
            raw = _as_int16_bytes(source).view(np.uint8)

            def reader():
                for o in range(0, raw.size, chunk_bytes):
                    yield raw[o:o + chunk_bytes]
        else:
            def reader():
                while True:
                    blob = source.read(chunk_bytes)
                    if not blob:
                        return
                    yield np.frombuffer(blob, np.uint8)

        media_bstring: List[np.ndarray] = []
        processed = 0.0
        for in_bytes in reader():
            processed += len(in_bytes) / float(BYTES_PER_SAMPLE) / self.frame_rate
            if self.progress_handler is not None:
                try:
                    self.progress_handler(processed)
                except Exception:  # a host callback must never break syncing (:731-734)
                    pass
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
