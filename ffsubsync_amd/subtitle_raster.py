"""On-device subtitle rasteriser: the step immediately before the aligner.

Mirrors ``SubtitleScaler`` (ffsubsync/subtitle_transformers.py:29-50) + ``SubtitleSpeechTransformer``
(ffsubsync/speech_transformers.py:946-984): subtitle intervals are sent to the GPU once (a few KB)
and rasterised there into the 100 Hz activity vector, for any number of framerate ratios, instead of
building a 720 KB float array per ratio on the host and copying seven of them over PCIe.  The
vectors stay in HBM as :class:`DeviceRaster` objects, which ``FFTAligner`` / ``MaxScoreAligner``
accept directly.
"""
from datetime import timedelta
from typing import Any, Iterable, List, Optional, Sequence

import zlib

import numpy as np

from . import _native
from .sklearn_shim import TransformerMixin
from .speech_transformers import ComputeSpeechFrameBoundariesMixin


class DeviceRaster:
    """A two-level activity vector in HBM: 0 -> ``lo``, 1 -> ``hi``.

    ``bits`` is either a uint8 CUDA tensor of 0/1 bytes (``n`` omitted) or, with ``n`` given, the
    bit-packed form the kernels prefer (int32 words, sample i = bit i & 31 of word i >> 5,
    ``_native.FFS_DTYPE_U1``): an eighth of the HBM bytes."""

    def __init__(self, bits, lo: float = 0.0, hi: float = 1.0, n: Optional[int] = None, runs=None,
                 runs_bound: int = 0) -> None:
        self.bits = bits
        self.lo = float(lo)
        self.hi = float(hi)
        self.packed = n is not None
        self.n = int(bits.numel()) if n is None else int(n)
        # the same vector as a boundary list (``ffs_runs_list`` block, FFS_DTYPE_RUNS) with a host-known upper bound of
        # its length: the aligner then neither extracts the boundaries from the bits nor waits for their count
        self.runs = runs
        self.runs_bound = int(runs_bound)

    def attach_runs(self) -> "DeviceRaster":
        """Extract the vector's boundary list once (one pass over the bits on the device, one 4-byte read-back of its
        length), so that every later solve starts from the list.  A vector too dense for the run-boundary path
        (32 768 boundaries or more) keeps its bits only."""
        if self.runs is None and self.n > 0:
            cap = min(32768, self.n + 2)
            block = _native.runs_from_bits(self.packed_words(), self.n, cap)
            count = int(block[0].item())
            if count < cap:
                self.runs, self.runs_bound = block, max(count, 1)
        return self

    def __len__(self) -> int:
        return self.n

    @property
    def size(self) -> int:
        return self.n

    @classmethod
    def from_host(cls, values, lists: bool = True) -> Optional["DeviceRaster"]:
        """Bit-packed device copy of a two-level host vector (e.g. a VAD label vector: {non_speech_label, 1.0}); None
        when the samples take more than two values.  One pass to find the levels, ``packbits``, one small upload;
        ``lists``: also its boundary list (:meth:`attach_runs`)."""
        import torch

        v = np.asarray(values, dtype=float).ravel()
        if v.size == 0:
            return None
        lo, hi = float(v.min()), float(v.max())
        is_hi = v == hi
        if not (np.isfinite(lo) and np.isfinite(hi)) or not bool(np.all(is_hi | (v == lo))):
            return None
        packed = np.packbits(is_hi if hi != lo else np.zeros(v.size, bool), bitorder="little")
        host = np.zeros((v.size + 31) // 32 * 4, dtype=np.uint8)
        host[: packed.size] = packed
        out = cls(torch.from_numpy(host).cuda().view(torch.int32), lo, hi, v.size)
        return out.attach_runs() if lists else out

    def bytes01(self):
        """0/1 uint8 CUDA tensor of the samples."""
        return _native.unpack_bits(self.bits, self.n) if self.packed else self.bits

    def packed_words(self):
        """int32 CUDA tensor of the bit-packed samples (converted on the device when held as bytes)."""
        return self.bits if self.packed else _native.pack_bits(self.bits)

    def __array__(self, dtype=None, copy=None):
        host = self.bytes01().cpu().numpy()
        out = np.where(host != 0, self.hi, self.lo).astype(float)
        return out if dtype is None else out.astype(dtype)

    def frames_float(self):
        """float32 CUDA tensor of the sample values (for the boundary scan)."""
        import torch

        b = self.bytes01()
        return torch.where(b != 0, torch.tensor(self.hi, device=b.device),
                           torch.tensor(self.lo, device=b.device)).to(torch.float32)


def _microseconds(td: timedelta) -> int:
    return (td.days * 86400 + td.seconds) * 10 ** 6 + td.microseconds


def subtitle_records(subs: Iterable[Any], is_metadata=None):
    """(start_us, end_us, meta) int64/uint8 arrays from objects with ``.start`` / ``.end`` timedeltas.
    ``is_metadata(content, is_first_or_last)`` is the reference's text heuristic
    (speech_transformers.py:926-943); it is host-side text logic and is taken from an importable
    ffsubsync when not supplied (no subtitle is skipped if neither is available)."""
    subs = list(subs)
    if is_metadata is None:
        try:
            from ffsubsync.speech_transformers import _is_metadata as is_metadata  # type: ignore
        except Exception:
            is_metadata = None
    start_us = np.array([_microseconds(s.start) for s in subs], dtype=np.int64)
    end_us = np.array([_microseconds(s.end) for s in subs], dtype=np.int64)
    meta = np.zeros(len(subs), dtype=np.uint8)
    if is_metadata is not None:
        for i, s in enumerate(subs):
            content = getattr(s, "content", None)
            if content is not None:
                meta[i] = 1 if is_metadata(content, i == 0 or i + 1 == len(subs)) else 0
    return start_us, end_us, meta


def rasterize_candidates(start_us, end_us, meta, ratios: Sequence[float], sample_rate: int = 100,
                         start_seconds: float = 0) -> List[DeviceRaster]:
    """One DeviceRaster per framerate ratio: times scaled by the ratio (SubtitleScaler), amplitude
    min(1/ratio, 1) (speech_transformers.py:977)."""
    out = []
    for r in ratios:
        words, n = _native.rasterize_subtitles(start_us, end_us, meta, r, sample_rate, start_seconds, packed=True)
        out.append(DeviceRaster(words, 0.0, min(1.0 / r, 1.0), n))
    _attach_interval_lists(out, start_us, end_us, meta, ratios, sample_rate, start_seconds)
    return out


def _attach_interval_lists(rasters, start_us, end_us, meta, ratios, sample_rate, start_seconds) -> None:
    """The rasters' boundary lists straight from the subtitle intervals (``ffs_rasterize_batch_runs``: the merged
    intervals ARE the list; one launch for all ratios, bound = two entries per subtitle)."""
    import torch

    count = int(np.size(start_us))
    if start_seconds > 0 or count == 0 or any(r.n <= 0 for r in rasters):
        return  # (negative start samples wrap around in Python slices: bits only)
    cap = 2 * count + 2
    stride = (_native.runs_list_bytes(cap) + 63) // 64 * 64
    k = len(rasters)
    data = torch.empty(k * stride, dtype=torch.uint8, device=rasters[0].bits.device)
    _native.rasterize_batch_runs(start_us, end_us, meta, np.zeros(k, np.int64), np.full(k, count, np.int64),
                                 np.asarray(list(ratios), dtype=np.float64), np.arange(k, dtype=np.int64) * stride,
                                 np.full(k, cap, np.int64), np.array([r.n for r in rasters], dtype=np.int64), data,
                                 sample_rate, float(start_seconds))
    for i, r in enumerate(rasters):
        r.runs = data[i * stride: i * stride + _native.runs_list_bytes(cap)].view(torch.int32)
        r.runs_bound = max(2 * count, 2)


class DeviceSubtitleSpeechTransformer(TransformerMixin, ComputeSpeechFrameBoundariesMixin):
    """Drop-in for ``SubtitleSpeechTransformer`` as the ``speech_extract`` step of the subtitle
    pipeline (it receives the already scaled subtitles from the ``scale`` step): same constructor,
    same fitted attributes (``subtitle_speech_results_`` is a :class:`DeviceRaster`, ``max_time_``,
    ``start_frame_`` / ``end_frame_`` / ``num_frames``)."""

    def __init__(self, sample_rate: int, start_seconds: int = 0, framerate_ratio: float = 1.0,
                 is_metadata=None) -> None:
        super(DeviceSubtitleSpeechTransformer, self).__init__()
        self.sample_rate = sample_rate
        self.start_seconds = start_seconds
        self.framerate_ratio = framerate_ratio
        self._is_metadata = is_metadata
        self.subtitle_speech_results_: Optional[DeviceRaster] = None
        self.max_time_: Optional[float] = None

    def fit(self, subs, *_) -> "DeviceSubtitleSpeechTransformer":
        start_us, end_us, meta = subtitle_records(subs, self._is_metadata)
        max_time = max([0] + [e / 10 ** 6 for e in end_us.tolist()])
        self.max_time_ = max_time - self.start_seconds
        words, n = _native.rasterize_subtitles(start_us, end_us, meta, 1.0, self.sample_rate, self.start_seconds,
                                               packed=True)
        self.subtitle_speech_results_ = DeviceRaster(words, 0.0, min(1.0 / self.framerate_ratio, 1.0), n)
        _attach_interval_lists([self.subtitle_speech_results_], start_us, end_us, meta, [1.0], self.sample_rate,
                               self.start_seconds)
        self.fit_boundaries(self.subtitle_speech_results_.frames_float())
        return self

    def transform(self, *_) -> DeviceRaster:
        assert self.subtitle_speech_results_ is not None
        return self.subtitle_speech_results_


def _vector_key(values):
    """Identity of a fitted vector's CONTENT: shape, dtype and a CRC-32 over every byte of it (about 2 ms for a 2 h
    float64 vector, against ~1 ms per solve saved by the cached device copy on every later fit), so any in-place edit --
    DeserializeSpeechTransformer's thresholding, a caller's post-processing of ``video_speech_results_`` -- is seen."""
    if not isinstance(values, np.ndarray) or values.size == 0:
        return None
    flat = np.ascontiguousarray(values).reshape(-1)
    return (values.shape, values.dtype.str, zlib.crc32(flat.view(np.uint8)))


def _device_copy_of(transformer, values):
    """The bit-packed device copy of a fitted reference vector, made once per vector content: cached on the transformer
    under the vector's shape, dtype and a checksum of ALL its bytes (a vector edited in place is uploaded again; a new
    array that happens to reuse the old one's ``id()`` is keyed by what it holds, not by where it lives)."""
    key = _vector_key(values)
    cached = transformer.__dict__.get("_ffs_device_copy")
    if key is not None and cached is not None and cached[0] == key:
        return cached[1]
    raster = None
    if isinstance(values, np.ndarray) and values.size:
        flat = np.ascontiguousarray(values, dtype=np.float64).ravel()
        found = _native.two_level_pack(flat)  # the same level-detection rule as the aligner's own host path (_Vec)
        if found is not None:
            import torch

            lo, hi, packed = found
            raster = DeviceRaster(torch.from_numpy(packed.view(np.int32).copy()).cuda(), lo, hi, flat.size).attach_runs()
    out = values if raster is None else raster  # more than two levels (fused / weighted labels): the host floats
    if key is not None:
        transformer.__dict__["_ffs_device_copy"] = (key, out)  # (the key holds no reference to the host vector)
    return out


def install_device_rasters(ref_speech_transformers, ref_main=None) -> None:
    """``install(device_rasters=True)``: see :func:`ffsubsync_amd.install`.  Idempotent."""
    ref_st = ref_speech_transformers
    ref_st.SubtitleSpeechTransformer = DeviceSubtitleSpeechTransformer  # looked up by subpipe_maker at call time (:81-87)
    if ref_main is not None and hasattr(ref_main, "SubtitleSpeechTransformer"):
        ref_main.SubtitleSpeechTransformer = DeviceSubtitleSpeechTransformer
    for name in ("VideoSpeechTransformer", "MultiSegmentVideoSpeechTransformer", "DeserializeSpeechTransformer"):
        cls = getattr(ref_st, name, None)
        if cls is None or getattr(cls.transform, "_ffs_wrapped", False):
            continue

        def transform(self, *args, _orig=cls.transform):
            return _device_copy_of(self, _orig(self, *args))

        transform._ffs_wrapped = True
        cls.transform = transform
