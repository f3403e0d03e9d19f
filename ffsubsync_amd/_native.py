"""ctypes binding of ``libffsalign.so`` (the C ABI declared in ``include/ffsubsync_amd.h``).

There is deliberately no fallback: if the HIP library is missing or no MI355X is visible, the
product path raises -- it never degrades to a CPU implementation.
"""
import ctypes
import math
import os
import threading
from typing import Dict, Optional, Tuple

import numpy as np

_LIB_NAME = "libffsalign.so"
_lib = None
_lib_lock = threading.Lock()

FFS_DTYPE_U8 = 0
FFS_DTYPE_F32 = 1
FFS_DTYPE_F64 = 3  # float64 samples (fp32 transforms nominate, fp64 re-evaluation of the caller's own samples)
FFS_DTYPE_U1 = 2  # one bit per sample, numpy.packbits(..., bitorder="little") order, 32-bit words
FFS_DTYPE_RUNS = 5  # the vector's boundary list (an `ffs_runs_list` device block: 16-byte header + 8-byte entries)
FLAG_EMPTY_WINDOW = 1
FLAG_AMBIGUOUS = 2
FLAG_FILTERED = 4
FLAG_DIRECT = 8

MAX_FFT_LENGTH = 1 << 24

CAND_RESULT_DTYPE = np.dtype(
    [("score", "<f8"), ("offset", "<i8"), ("score_f32", "<f4"), ("flags", "<i4")], align=True
)
PAIR_RESULT_DTYPE = np.dtype(
    [("score", "<f8"), ("offset", "<i8"), ("best_cand", "<i4"), ("flags", "<i4")], align=True
)
assert CAND_RESULT_DTYPE.itemsize == 24 and PAIR_RESULT_DTYPE.itemsize == 24

# every symbol include/ffsubsync_amd.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = (
    "ffs_fft_length",
    "ffs_plan_length",
    "ffs_plan_create",
    "ffs_plan_destroy",
    "ffs_plan_workspace_bytes",
    "ffs_align_batch",
    "ffs_align_batch_typed",
    "ffs_align_batch_runs",
    "ffs_runs_list_bytes",
    "ffs_runs_from_bits",
    "ffs_runs_from_bits_batch",
    "ffs_runs_to_bits",
    "ffs_rasterize_batch_runs",
    "ffs_correlate_full",
    "ffs_vad_energy",
    "ffs_vad_energy_bits",
    "ffs_speech_bounds",
    "ffs_vad_tokenize",
    "ffs_raster_length",
    "ffs_raster_intervals",
    "ffs_rasterize_subtitles",
    "ffs_rasterize_subtitles_bits",
    "ffs_raster_lengths",
    "ffs_rasterize_batch_bits",
    "ffs_two_level_pack",
    "ffs_pack_bits",
    "ffs_scatter_segments",
    "ffs_comm_unique_id",
    "ffs_comm_create",
    "ffs_gather_results",
    "ffs_comm_destroy",
    "ffs_plan_set_algorithm",
    "ffs_plan_runs_stats",
    "ffs_plan_profile",
    "ffs_plan_profile_read",
    "ffs_last_error",
    "ffs_version",
)
KERNEL_NAMES = ("pass_a", "mid", "pass_c", "nominees", "rescore", "runs_extract", "runs_corr", "levels")
FFS_ALGO_AUTO, FFS_ALGO_FFT, FFS_ALGO_RUNS = 0, 1, 2
ALGORITHMS = {"auto": FFS_ALGO_AUTO, "fft": FFS_ALGO_FFT, "runs": FFS_ALGO_RUNS}


def algorithm_code(algorithm) -> int:
    """FFS_ALGO_* code of "auto" / "fft" / "runs" (any case, surrounding blanks ignored) or of a code itself."""
    if isinstance(algorithm, str):
        name = algorithm.strip().lower()
        if name not in ALGORITHMS:
            raise ValueError("unknown algorithm %r: expected one of %s" % (algorithm, ", ".join(sorted(ALGORITHMS))))
        return ALGORITHMS[name]
    if isinstance(algorithm, (int, np.integer)) and int(algorithm) in ALGORITHMS.values():
        return int(algorithm)
    raise ValueError("unknown algorithm %r: expected one of %s or an FFS_ALGO_* code" % (algorithm, ", ".join(sorted(ALGORITHMS))))


def env_algorithm() -> str:
    """FFS_ALGORITHM of the environment, validated ("auto" when unset or empty)."""
    name = os.environ.get("FFS_ALGORITHM", "").strip().lower() or "auto"
    if name not in ALGORITHMS:
        raise ValueError("FFS_ALGORITHM=%r: expected one of %s" % (os.environ.get("FFS_ALGORITHM"), ", ".join(sorted(ALGORITHMS))))
    return name


class NativeError(RuntimeError):
    """A call into libffsalign.so returned a negative FFS_E_* code."""

    def __init__(self, code: int, message: str) -> None:
        super().__init__("libffsalign error %d: %s" % (code, message))
        self.code = code


def library_path() -> str:
    """In-tree libffsalign.so; FFS_LIBRARY_PATH selects another build of it (A/B runs of compile-time variants)."""
    return os.environ.get("FFS_LIBRARY_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load():
    """Load the shared library once; raise ImportError loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            raise ImportError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C ffsubsync_amd/csrc` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback." % path
            )
        # torch first: its wheel bundles a HIP runtime (libamdhip64) with the same SONAME the library links
        # against.  Loaded in that order both share one runtime; the other way round the process ends
        # up with two, and the second one finds no device ("no ROCm-capable device is detected").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(path)
        c = ctypes
        lib.ffs_fft_length.restype = c.c_int64
        lib.ffs_fft_length.argtypes = [c.c_int64, c.c_int64]
        lib.ffs_plan_length.restype = c.c_int64
        lib.ffs_plan_length.argtypes = [c.c_int64, c.c_int64, c.c_int64]
        lib.ffs_plan_create.restype = c.c_int
        lib.ffs_plan_create.argtypes = [c.c_int, c.c_int64, c.c_int, c.c_int, c.POINTER(c.c_void_p)]
        lib.ffs_plan_destroy.restype = c.c_int
        lib.ffs_plan_destroy.argtypes = [c.c_void_p]
        lib.ffs_plan_workspace_bytes.restype = c.c_int64
        lib.ffs_plan_workspace_bytes.argtypes = [c.c_void_p]
        lib.ffs_align_batch.restype = c.c_int
        lib.ffs_align_batch.argtypes = [
            c.c_void_p, c.c_int, c.c_int, c.c_int,
            c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
            c.c_int64, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p,
        ]
        lib.ffs_align_batch_typed.restype = c.c_int
        lib.ffs_align_batch_typed.argtypes = [
            c.c_void_p, c.c_int, c.c_int, c.c_void_p,
            c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
            c.c_int64, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p,
        ]
        lib.ffs_align_batch_runs.restype = c.c_int
        lib.ffs_align_batch_runs.argtypes = [
            c.c_void_p, c.c_int, c.c_int, c.c_void_p,
            c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p,
            c.c_int64, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p,
        ]
        lib.ffs_runs_list_bytes.restype = c.c_int64
        lib.ffs_runs_list_bytes.argtypes = [c.c_int64]
        lib.ffs_runs_from_bits.restype = c.c_int
        lib.ffs_runs_from_bits.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_int64, c.c_void_p]
        lib.ffs_runs_from_bits_batch.restype = c.c_int
        lib.ffs_runs_from_bits_batch.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p]
        lib.ffs_runs_to_bits.restype = c.c_int
        lib.ffs_runs_to_bits.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p]
        lib.ffs_rasterize_batch_runs.restype = c.c_int
        lib.ffs_rasterize_batch_runs.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p,
                                                 c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_double,
                                                 c.c_double, c.c_void_p, c.c_int64, c.c_void_p]
        lib.ffs_correlate_full.restype = c.c_int
        lib.ffs_correlate_full.argtypes = [
            c.c_void_p, c.c_int,
            c.c_void_p, c.c_int64, c.c_double, c.c_double,
            c.c_void_p, c.c_int64, c.c_double, c.c_double,
            c.c_void_p, c.c_int64, c.c_double, c.c_double,
            c.c_void_p, c.c_void_p, c.c_void_p,
        ]
        lib.ffs_vad_energy.restype = c.c_int
        lib.ffs_vad_energy.argtypes = [c.c_void_p, c.c_int64, c.c_int, c.c_double, c.c_float, c.c_void_p, c.c_void_p]
        lib.ffs_vad_energy_bits.restype = c.c_int
        lib.ffs_vad_energy_bits.argtypes = [c.c_void_p, c.c_int64, c.c_int, c.c_double, c.c_void_p, c.c_void_p]
        lib.ffs_vad_tokenize.restype = c.c_int
        lib.ffs_vad_tokenize.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_int, c.c_int, c.c_int, c.c_float,
                                         c.c_void_p, c.c_void_p]
        lib.ffs_speech_bounds.restype = c.c_int
        lib.ffs_speech_bounds.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p]
        lib.ffs_raster_length.restype = c.c_int64
        lib.ffs_raster_length.argtypes = [c.c_void_p, c.c_int64, c.c_double, c.c_double]
        lib.ffs_raster_intervals.restype = c.c_int64
        lib.ffs_raster_intervals.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_double, c.c_double,
                                             c.c_double, c.c_int64, c.c_void_p]
        lib.ffs_rasterize_subtitles.restype = c.c_int
        lib.ffs_rasterize_subtitles.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_double, c.c_double,
                                                c.c_double, c.c_void_p, c.c_int64, c.c_void_p]
        lib.ffs_rasterize_subtitles_bits.restype = c.c_int
        lib.ffs_rasterize_subtitles_bits.argtypes = lib.ffs_rasterize_subtitles.argtypes
        lib.ffs_raster_lengths.restype = c.c_int
        lib.ffs_raster_lengths.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_double, c.c_void_p]
        lib.ffs_two_level_pack.restype = c.c_int
        lib.ffs_two_level_pack.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p]
        lib.ffs_rasterize_batch_bits.restype = c.c_int
        lib.ffs_rasterize_batch_bits.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p,
                                                 c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_double, c.c_double,
                                                 c.c_void_p, c.c_int64, c.c_void_p]
        lib.ffs_pack_bits.restype = c.c_int
        lib.ffs_pack_bits.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_double, c.c_void_p, c.c_void_p]
        lib.ffs_scatter_segments.restype = c.c_int
        lib.ffs_scatter_segments.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_void_p, c.c_int64,
                                             c.c_void_p]
        lib.ffs_comm_unique_id.restype = c.c_int
        lib.ffs_comm_unique_id.argtypes = [c.c_void_p]
        lib.ffs_comm_create.restype = c.c_int
        lib.ffs_comm_create.argtypes = [c.c_int, c.c_int, c.c_int, c.c_void_p, c.POINTER(c.c_void_p)]
        lib.ffs_gather_results.restype = c.c_int
        lib.ffs_gather_results.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p]
        lib.ffs_comm_destroy.restype = c.c_int
        lib.ffs_comm_destroy.argtypes = [c.c_void_p]
        lib.ffs_plan_set_algorithm.restype = c.c_int
        lib.ffs_plan_set_algorithm.argtypes = [c.c_void_p, c.c_int]
        lib.ffs_plan_runs_stats.restype = c.c_int
        lib.ffs_plan_runs_stats.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
        lib.ffs_plan_profile.restype = c.c_int
        lib.ffs_plan_profile.argtypes = [c.c_void_p, c.c_int]
        lib.ffs_plan_profile_read.restype = c.c_int
        lib.ffs_plan_profile_read.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p]
        lib.ffs_last_error.restype = c.c_char_p
        lib.ffs_last_error.argtypes = []
        lib.ffs_version.restype = c.c_int
        lib.ffs_version.argtypes = []
        _lib = lib
        return _lib


def check(code: int) -> None:
    if code != 0:
        raise NativeError(code, load().ffs_last_error().decode("utf-8", "replace"))


def fft_length(ref_len: int, sub_len: int) -> int:
    return int(load().ffs_fft_length(int(ref_len), int(sub_len)))


def plan_length(ref_len: int, sub_len: int, max_offset_samples: Optional[int]) -> int:
    """Transform length the device needs: the reference's N without a lag window, possibly shorter
    (alias-free for the windowed lags) with one."""
    mo = -1 if max_offset_samples is None else int(max_offset_samples)
    return int(load().ffs_plan_length(int(ref_len), int(sub_len), mo))


_gpu_seen = None  # torch, once a device has been seen (torch.cuda.is_available() costs ~2.5 us a call, three per solve)


def require_gpu():
    """Return torch after checking a HIP device is visible (the product path needs one)."""
    global _gpu_seen
    if _gpu_seen is not None:
        return _gpu_seen
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "ffsubsync_amd needs an AMD Instinct GPU (ROCm/HIP device) -- none is visible and there is no CPU fallback"
        )
    _gpu_seen = torch
    return torch


def current_stream_ptr(torch) -> int:
    """Raw hipStream_t of torch's current stream on the current device."""
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:  # (no Stream object: ~1 us instead of ~7)
        return int(raw(torch.cuda.current_device()))
    return int(torch.cuda.current_stream().cuda_stream)


class Plan:
    """Owns one ``ffs_plan`` (twiddle tables + HBM workspace for one transform length)."""

    def __init__(self, n_fft: int, pairs_in_flight: int = 2, max_cand: int = 8, device: Optional[int] = None) -> None:
        torch = require_gpu()
        self.lib = load()
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.n_fft = int(n_fft)
        self.pairs_in_flight = int(pairs_in_flight)
        self.max_cand = int(max_cand)
        handle = ctypes.c_void_p()
        check(self.lib.ffs_plan_create(self.device, self.n_fft, self.pairs_in_flight, self.max_cand, ctypes.byref(handle)))
        self.handle = handle

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.ffs_plan_workspace_bytes(self.handle))

    def set_algorithm(self, algorithm, _from_env: bool = False) -> None:
        """"auto" (default: run-boundary path for short boundary lists, transforms otherwise), "fft" (transforms only) or
        "runs" (run-boundary path without the coincidence budget); identical results either way.  The choice belongs to
        whoever owns the plan: a plan obtained from ``get_plan`` (a per-thread cache shared between callers) gets
        FFS_ALGORITHM re-applied at every hand-out."""
        check(self.lib.ffs_plan_set_algorithm(self.handle, algorithm_code(algorithm)))
        if not _from_env:
            self._algorithm_explicit = True

    def runs_stats(self):
        """(calls that tried the run-boundary path, their sub-batches, sub-batches that went through the transforms)."""
        v = (ctypes.c_int64 * 4)()
        check(self.lib.ffs_plan_runs_stats(self.handle, ctypes.byref(v, 0), ctypes.byref(v, 8), ctypes.byref(v, 16), None))
        return int(v[0]), int(v[1]), int(v[2])

    def runs_boundaries_last_call(self) -> int:
        """Boundary-list entries of all vectors of the most recent run-boundary call."""
        v = ctypes.c_int64()
        check(self.lib.ffs_plan_runs_stats(self.handle, None, None, None, ctypes.byref(v)))
        return int(v.value)

    def profile(self, enable: bool) -> None:
        check(self.lib.ffs_plan_profile(self.handle, 1 if enable else 0))

    def profile_read(self):
        """{kernel name: (total ms, launches)} accumulated since the last read (synchronises)."""
        ms = np.zeros(len(KERNEL_NAMES), dtype=np.float64)
        n = np.zeros(len(KERNEL_NAMES), dtype=np.int64)
        check(self.lib.ffs_plan_profile_read(self.handle, ms.ctypes.data, n.ctypes.data))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(KERNEL_NAMES)}

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.ffs_plan_destroy(self.handle)
            self.handle = None

    def __del__(self) -> None:  # pragma: no cover - interpreter shutdown ordering
        try:
            self.close()
        except Exception:
            pass

    def align_batch(self, n_pairs: int, n_cand: int, dtype, vec_ptr: np.ndarray, vec_len: np.ndarray,
                    vec_lo: np.ndarray, vec_hi: np.ndarray, max_offset_samples: Optional[int],
                    filter_max_offset: Optional[int], cand_out, pair_out, stream: Optional[int] = None,
                    vec_max_boundaries: Optional[np.ndarray] = None) -> None:
        """Asynchronous batched solve; ``cand_out``/``pair_out`` are uint8 CUDA tensors of
        n_pairs*n_cand*24 and n_pairs*24 bytes.  Host arrays are consumed before returning.
        ``dtype``: one FFS_DTYPE_* for every vector, or a (reference type, candidate type) tuple / an array with one
        entry per vector (``ffs_align_batch_runs``: e.g. a float64 reference against bit-packed candidates, or
        boundary lists, FFS_DTYPE_RUNS).  ``vec_max_boundaries``: host-known upper bounds of the lists' lengths (int32 per
        vector, 0 = unknown): with all of them known and within the coincidence budget the call does not wait for the
        device."""
        torch = require_gpu()
        n_vec = n_pairs * (1 + n_cand)
        vec_ptr = np.ascontiguousarray(vec_ptr, dtype=np.uint64)
        vec_len = np.ascontiguousarray(vec_len, dtype=np.int64)
        vec_lo = np.ascontiguousarray(vec_lo, dtype=np.float64)
        vec_hi = np.ascontiguousarray(vec_hi, dtype=np.float64)
        if not (vec_ptr.size == vec_len.size == vec_lo.size == vec_hi.size == n_vec):
            raise ValueError("descriptor arrays must have n_pairs*(1+n_cand) entries")
        if cand_out.numel() * cand_out.element_size() < n_pairs * n_cand * 24 or \
                pair_out.numel() * pair_out.element_size() < n_pairs * 24:
            raise ValueError("result buffers too small")
        st = current_stream_ptr(torch) if stream is None else stream
        if not isinstance(dtype, (int, np.integer)):
            if isinstance(dtype, tuple) and len(dtype) == 2:
                dtype = np.tile(np.array([dtype[0]] + [dtype[1]] * n_cand, dtype=np.int32), n_pairs)
            vec_dtype = np.ascontiguousarray(dtype, dtype=np.int32)
            if vec_dtype.size != n_vec:
                raise ValueError("one element type per vector")
            bound = None
            if vec_max_boundaries is not None:
                bound = np.ascontiguousarray(vec_max_boundaries, dtype=np.int32)
                if bound.size != n_vec:
                    raise ValueError("one boundary bound per vector")
            check(self.lib.ffs_align_batch_runs(
                self.handle, n_pairs, n_cand, vec_dtype.ctypes.data,
                vec_ptr.ctypes.data, vec_len.ctypes.data, vec_lo.ctypes.data, vec_hi.ctypes.data,
                None if bound is None else bound.ctypes.data,
                -1 if max_offset_samples is None else int(max_offset_samples),
                -1 if filter_max_offset is None else int(filter_max_offset),
                cand_out.data_ptr(), pair_out.data_ptr(), st,
            ))
            return
        check(self.lib.ffs_align_batch(
            self.handle, n_pairs, n_cand, int(dtype),
            vec_ptr.ctypes.data, vec_len.ctypes.data, vec_lo.ctypes.data, vec_hi.ctypes.data,
            -1 if max_offset_samples is None else int(max_offset_samples),
            -1 if filter_max_offset is None else int(filter_max_offset),
            cand_out.data_ptr(), pair_out.data_ptr(), st,
        ))

    def correlate_full(self, dtype: int, ref, ref_levels, a, a_levels, b=None, b_levels=(0.0, 1.0), lens=None):
        """Raw fp32 correlation arrays out_a[m], out_b[m] (m = lag mod n_fft) as CUDA tensors.  ``lens`` =
        (R, Sa[, Sb]) in samples when the tensors are bit-packed (default: the tensors' element counts)."""
        torch = require_gpu()
        out_a = torch.empty(self.n_fft, dtype=torch.float32, device=ref.device)
        out_b = torch.empty(self.n_fft, dtype=torch.float32, device=ref.device) if b is not None else None
        n_ref, n_a = (ref.numel(), a.numel()) if lens is None else (int(lens[0]), int(lens[1]))
        n_b = 0 if b is None else (b.numel() if lens is None else int(lens[2]))
        check(self.lib.ffs_correlate_full(
            self.handle, dtype,
            ref.data_ptr(), n_ref, float(ref_levels[0]), float(ref_levels[1]),
            a.data_ptr(), n_a, float(a_levels[0]), float(a_levels[1]),
            b.data_ptr() if b is not None else None, n_b,
            float(b_levels[0]), float(b_levels[1]),
            out_a.data_ptr(), out_b.data_ptr() if out_b is not None else None, current_stream_ptr(torch),
        ))
        return out_a, out_b


class _PlanCache(threading.local):
    """Per-thread plan cache (a plan may be driven by one host thread at a time; the reference runs
    detectors/aligners from a small thread pool, speech_transformers.py:872-873).  Thread-local storage
    dies with its thread, so a worker that exits releases its plans (``Plan.__del__`` destroys them);
    within a thread the cache is an LRU bounded by an HBM budget."""

    def __init__(self) -> None:
        self.plans: "Dict[Tuple[int, int, int, int], Plan]" = {}
        self.order: list = []


_plan_cache = _PlanCache()
PLAN_CACHE_BYTES = int(os.environ.get("FFS_PLAN_CACHE_BYTES", str(8 << 30)))  # per thread


def get_plan(n_fft: int, pairs_in_flight: int = 1, max_cand: int = 8, device: Optional[int] = None) -> Plan:
    """Cached plan for (device, n_fft, pairs_in_flight, max_cand) of the calling thread."""
    torch = require_gpu()
    dev = torch.cuda.current_device() if device is None else int(device)
    key = (dev, int(n_fft), int(pairs_in_flight), int(max_cand))
    cache = _plan_cache
    plan = cache.plans.get(key)
    if plan is not None and getattr(plan, "handle", None) is None:  # closed behind the cache's back: forget it
        del cache.plans[key]
        cache.order.remove(key)
        plan = None
    if plan is None:
        plan = Plan(n_fft, pairs_in_flight, max_cand, dev)  # hipMalloc + table upload: no lock held
        cache.plans[key] = plan
    else:
        cache.order.remove(key)
    cache.order.append(key)
    # Cached plans are SHARED between unrelated callers of the thread: every hand-out re-applies FFS_ALGORITHM (auto | fft |
    # runs; results are identical either way), so a Plan.set_algorithm() by one user of a cached plan lasts until the next
    # get_plan() for that key and never leaks into another caller's solves.  To pin an algorithm, own the plan
    # (``Plan(...)`` / ``BatchAligner(algorithm=...)``).
    plan.set_algorithm(env_algorithm(), _from_env=True)
    plan._algorithm_explicit = False
    # evict least recently used plans beyond the budget (never the one just asked for)
    total = sum(p.workspace_bytes for p in cache.plans.values())
    while total > PLAN_CACHE_BYTES and len(cache.order) > 1:
        old = cache.order.pop(0)
        victim = cache.plans.pop(old)
        total -= victim.workspace_bytes
        victim.close()
    return plan


def clear_plan_cache() -> None:
    """Destroy the calling thread's cached plans (frees their HBM)."""
    for plan in _plan_cache.plans.values():
        plan.close()
    _plan_cache.plans.clear()
    _plan_cache.order.clear()


def vad_energy(pcm, frame_len: int, threshold_db: float, non_speech_label: float):
    """labels (float32 CUDA tensor) for an int16 CUDA tensor of mono PCM."""
    torch = require_gpu()
    n = pcm.numel()
    n_frames = (n + frame_len - 1) // frame_len
    labels = torch.empty(n_frames, dtype=torch.float32, device=pcm.device)
    if n:
        check(load().ffs_vad_energy(pcm.data_ptr(), n, int(frame_len), float(threshold_db), float(non_speech_label),
                                    labels.data_ptr(), current_stream_ptr(torch)))
    return labels


def vad_energy_bits(pcm, frame_len: int, threshold_db: float, out=None, first_frame: int = 0):
    """Bit-packed labels (int32 CUDA words, FFS_DTYPE_U1 order) for an int16 CUDA tensor of mono PCM.
    ``out`` / ``first_frame``: write into an existing zero-initialised word tensor at a frame offset that is a
    multiple of 8 (chunk-by-chunk sweeps of one file); returns (words, n_frames)."""
    torch = require_gpu()
    n = pcm.numel()
    n_frames = (n + frame_len - 1) // frame_len
    if first_frame % 8:
        raise ValueError("first_frame must be a multiple of 8")
    if out is None:
        out = torch.zeros((first_frame + n_frames + 31) // 32, dtype=torch.int32, device=pcm.device)
    elif out.numel() * out.element_size() * 8 < first_frame + n_frames:
        raise ValueError("output word buffer too small")
    elif out.device != pcm.device or not out.is_contiguous():
        raise ValueError("output word buffer must be contiguous and on the PCM's device")
    if n:
        check(load().ffs_vad_energy_bits(pcm.data_ptr(), n, int(frame_len), float(threshold_db),
                                         out.data_ptr() + first_frame // 8, current_stream_ptr(torch)))
    return out, n_frames


def vad_tokenize(valid, chunk_frames: int, min_length: int, max_length: int, max_continuous_silence: int,
                 non_speech_label: float):
    """auditok-style token smoothing of a float32 CUDA validity vector (see ffs_vad_tokenize)."""
    torch = require_gpu()
    out = torch.empty_like(valid)
    if valid.numel():
        check(load().ffs_vad_tokenize(valid.data_ptr(), valid.numel(), int(chunk_frames), int(math.ceil(min_length)),
                                      int(max_length), int(math.ceil(max_continuous_silence)), float(non_speech_label), out.data_ptr(),
                                      current_stream_ptr(torch)))
    return out


def speech_bounds(frames):
    """(first, last) index with frames > 0.5 for a float32 CUDA tensor, or (None, None)."""
    torch = require_gpu()
    out = torch.empty(2, dtype=torch.int64, device=frames.device)
    check(load().ffs_speech_bounds(frames.data_ptr() if frames.numel() else None, frames.numel(), out.data_ptr(),
                                   current_stream_ptr(torch)))
    lo, hi = (int(v) for v in out.cpu())
    return (None, None) if hi < 0 else (lo, hi)


def _us_arrays(start_us, end_us, is_metadata):
    start_us = np.ascontiguousarray(start_us, dtype=np.int64)
    end_us = np.ascontiguousarray(end_us, dtype=np.int64)
    if start_us.shape != end_us.shape:
        raise ValueError("start_us and end_us must have the same length")
    meta = None if is_metadata is None else np.ascontiguousarray(is_metadata, dtype=np.uint8)
    return start_us, end_us, meta


def raster_length(end_us, ratio: float, sample_rate: float) -> int:
    end_us = np.ascontiguousarray(end_us, dtype=np.int64)
    return int(load().ffs_raster_length(end_us.ctypes.data, end_us.size, float(ratio), float(sample_rate)))


def raster_intervals(start_us, end_us, is_metadata, ratio, sample_rate, start_seconds, out_len) -> np.ndarray:
    """Host-only: [n, 2] int32 array of the clamped [start, end) sample intervals."""
    start_us, end_us, meta = _us_arrays(start_us, end_us, is_metadata)
    iv = np.zeros((start_us.size + 1, 2), dtype=np.int32)
    n = load().ffs_raster_intervals(start_us.ctypes.data, end_us.ctypes.data, None if meta is None else meta.ctypes.data,
                                    start_us.size, float(ratio), float(sample_rate), float(start_seconds), int(out_len),
                                    iv.ctypes.data)
    return iv[:n]


def rasterize_subtitles(start_us, end_us, is_metadata, ratio, sample_rate=100.0, start_seconds=0.0, packed=False):
    """The subtitle track rescaled by ``ratio`` as a CUDA tensor: 0/1 bytes (uint8[n]), or with
    ``packed`` one bit per sample (int32[ceil(n/32)], FFS_DTYPE_U1) -- returns (tensor, n) then."""
    torch = require_gpu()
    start_us, end_us, meta = _us_arrays(start_us, end_us, is_metadata)
    n = raster_length(end_us, ratio, sample_rate)
    fn = load().ffs_rasterize_subtitles_bits if packed else load().ffs_rasterize_subtitles
    out = torch.empty((n + 31) // 32, dtype=torch.int32, device="cuda") if packed else \
        torch.empty(n, dtype=torch.uint8, device="cuda")
    check(fn(start_us.ctypes.data, end_us.ctypes.data, None if meta is None else meta.ctypes.data, start_us.size,
             float(ratio), float(sample_rate), float(start_seconds), out.data_ptr(), n, current_stream_ptr(torch)))
    return (out, n) if packed else out


def raster_lengths(track_end_us_max, ratio, sample_rate=100.0) -> np.ndarray:
    """Host-only: raster lengths of many vectors at once (``ffs_raster_lengths``): vector v is a track whose largest
    end time is ``track_end_us_max[v]`` microseconds, scaled by ``ratio[v]``."""
    ends = np.ascontiguousarray(track_end_us_max, dtype=np.int64)
    ratio = np.ascontiguousarray(ratio, dtype=np.float64)
    if ends.shape != ratio.shape:
        raise ValueError("one ratio per vector")
    out = np.zeros(ends.size, dtype=np.int64)
    check(load().ffs_raster_lengths(ends.ctypes.data, ratio.ctypes.data, ends.size, float(sample_rate), out.ctypes.data))
    return out


def rasterize_batch_bits(start_us, end_us, is_metadata, vec_sub_first, vec_sub_count, vec_ratio, vec_out_word, vec_len,
                         out, sample_rate=100.0, start_seconds=0.0) -> None:
    """``ffs_rasterize_batch_bits``: every vector of a batch from ONE call, interval arithmetic on the device.  The
    tracks are concatenated in ``start_us`` / ``end_us`` / ``is_metadata``; vector v covers subtitles
    [vec_sub_first[v], +vec_sub_count[v]) scaled by vec_ratio[v] and lands as vec_len[v] bits at word vec_out_word[v]
    of ``out`` (int32 / uint8 CUDA tensor, zeroed by the call)."""
    torch = require_gpu()
    start_us, end_us, meta = _us_arrays(start_us, end_us, is_metadata)
    i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    first, count, word, length = i64(vec_sub_first), i64(vec_sub_count), i64(vec_out_word), i64(vec_len)
    ratio = np.ascontiguousarray(vec_ratio, dtype=np.float64)
    n_vec = first.size
    if not (count.size == word.size == length.size == ratio.size == n_vec):
        raise ValueError("the vector tables must have the same length")
    out_words = out.numel() * out.element_size() // 4
    check(load().ffs_rasterize_batch_bits(start_us.ctypes.data, end_us.ctypes.data, None if meta is None else meta.ctypes.data,
                                          start_us.size, first.ctypes.data, count.ctypes.data, ratio.ctypes.data,
                                          word.ctypes.data, length.ctypes.data, n_vec, float(sample_rate),
                                          float(start_seconds), out.data_ptr(), out_words, current_stream_ptr(torch)))


def runs_list_bytes(cap: int) -> int:
    """Bytes of an ``ffs_runs_list`` block with room for ``cap`` entries (16-byte header + 8 bytes per entry)."""
    return 16 + 8 * int(cap)


def rasterize_batch_runs(start_us, end_us, is_metadata, vec_sub_first, vec_sub_count, vec_ratio, vec_out_off, vec_cap,
                         vec_len, out, sample_rate=100.0, start_seconds=0.0) -> None:
    """``ffs_rasterize_batch_runs``: every vector of a batch as its BOUNDARY LIST (FFS_DTYPE_RUNS), no bitmap.  Vector v
    covers subtitles [vec_sub_first[v], +vec_sub_count[v]) scaled by vec_ratio[v]; its list block starts at byte
    vec_out_off[v] of ``out`` (CUDA tensor; offsets multiples of 8) and holds up to vec_cap[v] >= 2 * count + 1 entries.
    ``start_us`` / ``end_us`` / ``is_metadata``: host arrays, or int64 / int64 / uint8 CUDA tensors (tracks uploaded once,
    sorted by start time: nothing of them is copied per call)."""
    torch = require_gpu()
    i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    if hasattr(start_us, "data_ptr"):  # device-resident tables
        first, count, off, cap, length = i64(vec_sub_first), i64(vec_sub_count), i64(vec_out_off), i64(vec_cap), i64(vec_len)
        ratio = np.ascontiguousarray(vec_ratio, dtype=np.float64)
        n_vec = first.size
        if not (count.size == off.size == cap.size == length.size == ratio.size == n_vec):
            raise ValueError("the vector tables must have the same length")
        check(load().ffs_rasterize_batch_runs(start_us.data_ptr(), end_us.data_ptr(),
                                              None if is_metadata is None else is_metadata.data_ptr(), start_us.numel(),
                                              first.ctypes.data, count.ctypes.data, ratio.ctypes.data, off.ctypes.data,
                                              cap.ctypes.data, length.ctypes.data, n_vec, float(sample_rate),
                                              float(start_seconds), out.data_ptr(), out.numel() * out.element_size(),
                                              current_stream_ptr(torch)))
        return
    start_us, end_us, meta = _us_arrays(start_us, end_us, is_metadata)
    first, count, off, cap, length = i64(vec_sub_first), i64(vec_sub_count), i64(vec_out_off), i64(vec_cap), i64(vec_len)
    ratio = np.ascontiguousarray(vec_ratio, dtype=np.float64)
    n_vec = first.size
    if not (count.size == off.size == cap.size == length.size == ratio.size == n_vec):
        raise ValueError("the vector tables must have the same length")
    check(load().ffs_rasterize_batch_runs(start_us.ctypes.data, end_us.ctypes.data, None if meta is None else meta.ctypes.data,
                                          start_us.size, first.ctypes.data, count.ctypes.data, ratio.ctypes.data,
                                          off.ctypes.data, cap.ctypes.data, length.ctypes.data, n_vec, float(sample_rate),
                                          float(start_seconds), out.data_ptr(), out.numel() * out.element_size(),
                                          current_stream_ptr(torch)))


def runs_from_bits(words, n: int, cap: Optional[int] = None, out=None):
    """Boundary list (``ffs_runs_list`` block, int32 CUDA tensor of 4 + 2 * cap words) of a FFS_DTYPE_U1 vector of ``n``
    samples: one pass over the bits.  ``cap`` defaults to room for every possible boundary count below 32 768."""
    torch = require_gpu()
    cap = 32768 if cap is None else int(cap)
    if out is None:
        out = torch.empty(4 + 2 * cap, dtype=torch.int32, device=words.device)
    elif out.numel() * out.element_size() < runs_list_bytes(cap):
        raise ValueError("list block too small")
    check(load().ffs_runs_from_bits(words.data_ptr(), int(n), out.data_ptr(), cap, current_stream_ptr(torch)))
    return out


def runs_from_bits_batch(bits_ptr, lens, list_ptr, caps) -> None:
    """``ffs_runs_from_bits_batch``: n vectors (device pointers to FFS_DTYPE_U1 words, lengths in samples) into n list
    blocks (device pointers, capacities in entries) with one launch."""
    torch = require_gpu()
    bits_ptr = np.ascontiguousarray(bits_ptr, dtype=np.uint64)
    list_ptr = np.ascontiguousarray(list_ptr, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.int64)
    caps = np.ascontiguousarray(caps, dtype=np.int64)
    if not (bits_ptr.size == list_ptr.size == lens.size == caps.size):
        raise ValueError("the vector tables must have the same length")
    check(load().ffs_runs_from_bits_batch(bits_ptr.ctypes.data, lens.ctypes.data, list_ptr.ctypes.data, caps.ctypes.data,
                                          bits_ptr.size, current_stream_ptr(torch)))


def runs_to_bits(block, n: int, out=None):
    """FFS_DTYPE_U1 words (int32 CUDA tensor) of the ``n``-sample vector an ``ffs_runs_list`` block describes (written
    into ``out`` when given: at least ``packed_words(n)`` int32)."""
    torch = require_gpu()
    if out is None:
        out = torch.empty(packed_words(n), dtype=torch.int32, device=block.device)
    assert out.dtype == torch.int32 and out.numel() >= packed_words(n)
    check(load().ffs_runs_to_bits(block.data_ptr(), int(n), out.data_ptr(), current_stream_ptr(torch)))
    return out


def runs_list_host(block) -> Tuple[np.ndarray, np.ndarray, int]:
    """(positions, ones_before, ones) of a list block, on the host (tests, diagnostics)."""
    raw = block.view(require_gpu().int32).cpu().numpy()
    n, ones = int(raw[0]), int(raw[1])
    e = raw[4:4 + 2 * n].reshape(n, 2)
    return e[:, 0].copy(), e[:, 1].copy(), ones


def two_level_pack(values: np.ndarray):
    """Host-only (``ffs_two_level_pack``): (lo, hi, packed) for a contiguous float64 vector whose samples take two
    finite levels -- ``packed`` = the samples as little-endian bits (uint8[4 * ceil(n/32)], bit i = (values[i] == hi))
    -- or None when they do not."""
    assert values.dtype == np.float64 and values.flags.c_contiguous and values.ndim == 1 and values.size > 0
    levels = np.zeros(2, dtype=np.float64)
    words = np.empty((values.size + 31) // 32, dtype=np.uint32)
    rc = load().ffs_two_level_pack(values.ctypes.data, values.size, levels.ctypes.data, levels[1:].ctypes.data,
                                   words.ctypes.data)
    if rc < 0:
        check(rc)
    return (float(levels[0]), float(levels[1]), words.view(np.uint8)) if rc == 1 else None


def packed_words(n: int) -> int:
    return (int(n) + 31) // 32


def pack_bits(src, threshold: float = 0.5, out=None):
    """FFS_DTYPE_U1 image (int32 CUDA tensor of ceil(n/32) words) of a two-level CUDA vector:
    uint8 -> bit = (byte != 0); float32 -> bit = (x > threshold)."""
    torch = require_gpu()
    n = src.numel()
    if src.dtype == torch.uint8:
        dt = FFS_DTYPE_U8
    elif src.dtype == torch.float32:
        dt = FFS_DTYPE_F32
    else:
        raise TypeError("pack_bits needs a uint8 or float32 tensor, got %s" % src.dtype)
    if out is None:
        out = torch.empty(packed_words(n), dtype=torch.int32, device=src.device)
    elif out.numel() * out.element_size() < packed_words(n) * 4:
        raise ValueError("output too small")
    if n:
        check(load().ffs_pack_bits(src.data_ptr(), dt, n, float(threshold), out.data_ptr(), current_stream_ptr(torch)))
    return out


def unpack_bits(words, n: int):
    """uint8 0/1 CUDA tensor of the first n samples of a FFS_DTYPE_U1 vector (torch ops; diagnostics and
    the boundary scan -- the hot kernels read the packed words directly)."""
    torch = require_gpu()
    w = words.view(torch.int32).reshape(-1)
    shifts = torch.arange(32, device=w.device, dtype=torch.int32)
    return ((w[:, None] >> shifts[None, :]) & 1).to(torch.uint8).reshape(-1)[:n]


def scatter_segments(labels, src_off, dst_start, seg_len, out_len: int):
    """float32 CUDA vector of ``out_len`` zeros with labels[src_off[i]:src_off[i]+seg_len[i]] copied to
    dst_start[i] for every sampled window (see ffs_scatter_segments)."""
    torch = require_gpu()
    src_off = np.ascontiguousarray(src_off, dtype=np.int64)
    dst_start = np.ascontiguousarray(dst_start, dtype=np.int64)
    seg_len = np.ascontiguousarray(seg_len, dtype=np.int64)
    if not (src_off.size == dst_start.size == seg_len.size):
        raise ValueError("segment arrays must have the same length")
    if src_off.size and int((src_off + seg_len).max()) > labels.numel():
        raise ValueError("segment reaches beyond the label buffer")
    out = torch.empty(int(out_len), dtype=torch.float32, device=labels.device)
    check(load().ffs_scatter_segments(labels.data_ptr() if labels.numel() else None, src_off.ctypes.data,
                                      dst_start.ctypes.data, seg_len.ctypes.data, int(src_off.size), out.data_ptr(),
                                      int(out_len), current_stream_ptr(torch)))
    return out


class Comm:
    """RCCL communicator behind ``ffs_gather_results`` (one per process / GPU).  ``unique_id()`` on rank 0,
    publish the 128 bytes (e.g. through torch.distributed's store), then ``Comm(rank, world, id)`` everywhere."""

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        check(load().ffs_comm_unique_id(buf))
        return buf.raw

    def __init__(self, rank: int, world: int, uid: bytes, device: Optional[int] = None) -> None:
        torch = require_gpu()
        self.lib = load()
        self.rank, self.world = int(rank), int(world)
        self.device = torch.cuda.current_device() if device is None else int(device)
        if len(uid) != 128:
            raise ValueError("unique id must be 128 bytes")
        handle = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(uid, 128)
        check(self.lib.ffs_comm_create(self.device, self.rank, self.world, buf, ctypes.byref(handle)))
        self.handle = handle

    def gather_pair_results(self, local, out=None):
        """All-gather this rank's uint8 tensor of n*24 result bytes; returns world*n*24 bytes in rank order."""
        torch = require_gpu()
        n_local = local.numel() // 24
        if out is None:
            out = torch.empty(self.world * n_local * 24, dtype=torch.uint8, device=local.device)
        check(self.lib.ffs_gather_results(self.handle, local.data_ptr(), n_local, out.data_ptr(), current_stream_ptr(torch)))
        return out

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.ffs_comm_destroy(self.handle)
            self.handle = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
