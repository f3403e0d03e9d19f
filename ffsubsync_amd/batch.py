"""Batched, device-resident driver for many (reference, candidates) alignment problems.

This is the throughput path behind the headline metric (seven-ratio MaxScoreAligner solves per
second): all activity vectors of a batch live back to back in one HBM buffer -- bit-packed
(``FFS_DTYPE_U1``, one bit per 10 ms frame) or as 0/1 bytes -- one call into ``ffs_align_batch`` solves
every problem, and across GPUs the problems are sharded by pair (no data exchange during solves)
with one all-gather of the 24-byte results over RCCL (``ffs_gather_results``).
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native


@dataclass
class DeviceBatch:
    """Packed two-level activity vectors in HBM.  Row p of the descriptor arrays holds pair p's
    reference followed by its candidates."""

    data: "object"  # torch.uint8 CUDA tensor (raw bytes of the buffer, whatever the element type)
    offs: np.ndarray  # [n_pairs, 1+n_cand] byte offsets into data
    lens: np.ndarray  # [n_pairs, 1+n_cand] samples
    lo: np.ndarray
    hi: np.ndarray
    dtype: int = _native.FFS_DTYPE_U8
    ref_dtype: Optional[int] = None  # element type of the references when it differs from the candidates' (`dtype`)
    # FFS_DTYPE_RUNS vectors: host-known upper bounds of the boundary lists' lengths ([n_pairs, 1+n_cand] int32, 0 =
    # unknown).  With all of them known a solve within the coincidence budget never waits for the device.
    bounds: Optional[np.ndarray] = None

    @property
    def call_dtype(self):
        """What ``Plan.align_batch`` takes: one type, or (reference type, candidate type) when the roles differ."""
        return self.dtype if self.ref_dtype in (None, self.dtype) else (self.ref_dtype, self.dtype)

    @property
    def n_pairs(self) -> int:
        return self.offs.shape[0]

    @property
    def n_cand(self) -> int:
        return self.offs.shape[1] - 1

    def select_candidates(self, index: Sequence[int]) -> "DeviceBatch":
        """View with one candidate per pair (``index[p]`` of pair p's candidates), sharing ``data``:
        the single-ratio FFTAligner problem of every pair."""
        rows = np.arange(self.n_pairs)
        cols = 1 + np.asarray(index, dtype=np.int64)
        pick = lambda a: np.ascontiguousarray(np.stack([a[:, 0], a[rows, cols]], axis=1))
        return DeviceBatch(self.data, pick(self.offs), pick(self.lens), pick(self.lo), pick(self.hi), self.dtype, self.ref_dtype,
                           None if self.bounds is None else pick(self.bounds))

    def required_fft_length(self, max_offset_samples: Optional[int] = None, reference_length: bool = False) -> int:
        """Plan length for the whole batch: the shortest alias-free transform for the lags the solve has
        to evaluate (``_native.plan_length``), or with ``reference_length`` the reference's own
        N = 2^ceil(log2(R+S))."""
        n = 2
        for p in range(self.n_pairs):
            for j in range(1, self.offs.shape[1]):
                r, s = int(self.lens[p, 0]), int(self.lens[p, j])
                n = max(n, _native.fft_length(r, s) if reference_length else _native.plan_length(r, s, max_offset_samples))
        return n

    def to_bits(self) -> "DeviceBatch":
        """The same batch bit-packed (FFS_DTYPE_U1): one device pass over the byte buffer.  Vector offsets
        are multiples of 64 bytes, so every packed vector starts on an 8-byte boundary."""
        if self.dtype == _native.FFS_DTYPE_U1:
            return self
        if self.dtype != _native.FFS_DTYPE_U8 or self.ref_dtype not in (None, self.dtype):
            raise ValueError("only 0/1 byte batches can be bit-packed")
        assert not (self.offs % 64).any() and self.data.numel() % 64 == 0
        words = _native.pack_bits(self.data)
        return DeviceBatch(words.view(self.data.dtype), self.offs // 8, self.lens, self.lo, self.hi, _native.FFS_DTYPE_U1)

    def to_runs(self, cap: int = 32768) -> "DeviceBatch":
        """The same batch as boundary lists (FFS_DTYPE_RUNS): one ``ffs_runs_from_bits`` pass per vector over a bit-packed
        batch.  Conversions pay off when the vectors are solved more than once (a reference against many subtitle
        files, the steps of a golden-section search); ``cap`` entries per list."""
        torch = _native.require_gpu()
        if self.dtype == _native.FFS_DTYPE_RUNS and self.ref_dtype in (None, self.dtype):
            return self
        if self.dtype != _native.FFS_DTYPE_U1 or self.ref_dtype not in (None, self.dtype):
            raise ValueError("only bit-packed batches can be converted to boundary lists")
        block = (_native.runs_list_bytes(cap) + 63) // 64 * 64
        n = self.offs.size
        data = torch.empty(n * block, dtype=torch.uint8, device=self.data.device)
        offs = (np.arange(n, dtype=np.int64) * block).reshape(self.offs.shape)
        _native.runs_from_bits_batch(self.data.data_ptr() + self.offs.ravel().astype(np.uint64), self.lens.ravel(),
                                     data.data_ptr() + offs.ravel().astype(np.uint64), np.full(n, cap, dtype=np.int64))
        return DeviceBatch(data, offs, self.lens, self.lo, self.hi, _native.FFS_DTYPE_RUNS)


def _layout(lens: np.ndarray, bytes_per_vec: np.ndarray):
    padded = (bytes_per_vec + 63) // 64 * 64
    offs = np.concatenate([[0], np.cumsum(padded.ravel())[:-1]]).reshape(lens.shape).astype(np.int64)
    return offs, int(padded.sum())


def pack_pairs(pairs, packed: bool = True) -> DeviceBatch:
    """DeviceBatch from HBM-resident two-level vectors: ``pairs`` is a list of
    (reference, [candidates]) of ``subtitle_raster.DeviceRaster`` objects (bytes or bit-packed).  The
    vectors are copied device-to-device into one buffer at 64-byte aligned offsets, bit-packed by
    default."""
    torch = _native.require_gpu()
    n_pairs, n_vec = len(pairs), 1 + len(pairs[0][1])
    flat = []
    for ref, cands in pairs:
        if len(cands) != n_vec - 1:
            raise ValueError("all pairs need the same number of candidates")
        flat.append(ref)
        flat.extend(cands)
    lens = np.array([len(v) for v in flat], dtype=np.int64).reshape(n_pairs, n_vec)
    nbytes = (lens + 31) // 32 * 4 if packed else lens
    offs, total = _layout(lens, nbytes)
    data = torch.zeros(total, dtype=torch.uint8, device=flat[0].bits.device)
    for v, o, nb in zip(flat, offs.ravel(), nbytes.ravel()):
        src = v.packed_words().view(torch.uint8) if packed else v.bytes01()
        data[int(o): int(o) + int(nb)] = src[: int(nb)]
    lo = np.array([v.lo for v in flat], dtype=np.float64).reshape(n_pairs, n_vec)
    hi = np.array([v.hi for v in flat], dtype=np.float64).reshape(n_pairs, n_vec)
    return DeviceBatch(data, offs, lens, lo, hi, _native.FFS_DTYPE_U1 if packed else _native.FFS_DTYPE_U8)


class TrackSet:
    """Subtitle tracks ((start_us, end_us, is_metadata) triples) laid out once for repeated batched rasterisation: the
    concatenated interval tables and each track's extent stay on the host, ``rasterize`` turns any selection of
    (track, ratio) vectors into bit-packed rasters with ONE ``ffs_rasterize_batch_bits`` call."""

    def __init__(self, tracks, arena: "Optional[TrackSet]" = None) -> None:
        """``arena``: an earlier TrackSet whose host tables may be overwritten (the caller is done with it: its rasteriser
        calls have returned -- ``ffs_rasterize_batch_*`` stages pageable tables before it returns) -- a stream of batches
        then fills the same few megabytes again instead of faulting in fresh pages for every batch."""
        n = len(tracks)
        self.counts = np.fromiter((len(t[0]) for t in tracks), dtype=np.int64, count=n)
        self.firsts = np.zeros(n, dtype=np.int64)
        if n:
            np.cumsum(self.counts[:-1], out=self.firsts[1:])
        total = int(self.counts.sum())
        # one preallocated table per column, filled in place by ONE concatenation each (no per-track temporaries: arrays that
        # already have the column's type are copied straight in)
        base = getattr(arena, "_base", None) if arena is not None and arena._dev is None and arena._pinned is None else None
        if base is None or base[0].size < total:
            base = (np.empty(total + total // 8, np.int64), np.empty(total + total // 8, np.int64), np.empty(total + total // 8, np.uint8))
        self._base = base
        self.start_us, self.end_us = base[0][:total], base[1][:total]
        if total:
            np.concatenate([t[0] for t in tracks], out=self.start_us, casting="unsafe")
            np.concatenate([t[1] for t in tracks], out=self.end_us, casting="unsafe")
        if all(t[2] is None for t in tracks):
            self.meta = None
        else:  # a track without flags has no metadata lines
            self.meta = base[2][:total]
            if total:
                np.concatenate([np.zeros(len(t[0]), np.uint8) if t[2] is None else t[2] for t in tracks], out=self.meta, casting="unsafe")
        # largest end of every track in one vectorised pass (round 5: 512 np.max calls per batch were half of the constructor)
        self.end_max = np.zeros(n, dtype=np.int64)
        filled = np.flatnonzero(self.counts > 0)
        if filled.size:
            self.end_max[filled] = np.maximum.reduceat(self.end_us, self.firsts[filled])
        self._dev = None  # (start_us, end_us, is_metadata) CUDA tensors once to_device() has uploaded them
        self._pinned = None  # pinned host tensors behind start_us / end_us / meta once pin() has moved them there

    def pin(self) -> "TrackSet":
        """Move the interval tables into pinned host memory (sorted by start time inside every track): ``rasterize_runs``
        then uploads them straight from there, without the staging copy a pageable array needs (measured, 256 pairs x 8
        vectors: 0.31 instead of 0.43 ms per synchronised call; in a stream of unsynchronised batches the staged path is
        the faster one -- profiles/ingest_profile.py).  The TrackSet must stay alive until the stream has passed the calls
        that used it."""
        torch = _native.require_gpu()
        if self._pinned is None:
            order = np.arange(self.start_us.size)
            for f, c in zip(self.firsts, self.counts):
                seg = self.start_us[f:f + c]
                if c > 1 and (seg[1:] < seg[:-1]).any():
                    order[f:f + c] = f + np.argsort(seg, kind="stable")
            pin = lambda a: torch.from_numpy(np.ascontiguousarray(a[order])).pin_memory()
            self._pinned = (pin(self.start_us), pin(self.end_us), None if self.meta is None else pin(self.meta))
            self.start_us, self.end_us = self._pinned[0].numpy(), self._pinned[1].numpy()
            if self.meta is not None:
                self.meta = self._pinned[2].numpy()
        return self

    def to_device(self) -> "TrackSet":
        """Upload the interval tables once (sorted by start time inside every track): ``rasterize_runs`` then copies
        nothing but its 32-byte-per-vector table -- for tracks that are rasterised again and again (the steps of a
        golden-section search, one subtitle file against many references)."""
        torch = _native.require_gpu()
        if self._dev is None:
            order = np.arange(self.start_us.size)
            for f, c in zip(self.firsts, self.counts):
                seg = self.start_us[f:f + c]
                if c > 1 and (seg[1:] < seg[:-1]).any():
                    order[f:f + c] = f + np.argsort(seg, kind="stable")
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a[order])).cuda()
            self._dev = (up(self.start_us), up(self.end_us), None if self.meta is None else up(self.meta))
        return self

    def rasterize(self, track_of, ratio, sample_rate: int = 100, start_seconds: float = 0):
        """Vector v = track ``track_of[v]`` with its times scaled by ``ratio[v]`` (``SubtitleScaler`` +
        ``SubtitleSpeechTransformer``, speech_transformers.py:957-980).  Returns (uint8 CUDA buffer, byte offset of every
        vector, length of every vector in samples); interval arithmetic on the device, no per-vector tensors."""
        torch = _native.require_gpu()
        track_of = np.asarray(track_of, dtype=np.int64).ravel()
        ratio = np.ascontiguousarray(ratio, dtype=np.float64).ravel()
        lens = _native.raster_lengths(self.end_max[track_of], ratio, sample_rate)
        offs, total = _layout(lens, (lens + 31) // 32 * 4)
        data = torch.empty(total, dtype=torch.uint8, device="cuda")
        _native.rasterize_batch_bits(self.start_us, self.end_us, self.meta, self.firsts[track_of], self.counts[track_of], ratio,
                                     offs // 4, lens, data, sample_rate, start_seconds)
        return data, offs, lens

    def rasterize_runs(self, track_of, ratio, sample_rate: int = 100, start_seconds: float = 0):
        """The same vectors as BOUNDARY LISTS (FFS_DTYPE_RUNS; ``ffs_rasterize_batch_runs``): no bitmap is written, the
        merged intervals are the list.  Returns (uint8 CUDA buffer, byte offset of every vector's list block, length of
        every vector in samples, upper bound of every list's length: two entries per subtitle).  ``start_seconds`` as in
        ``rasterize`` (speech_transformers.py:968-972); the list form needs ``start_seconds <= 0`` (a positive one makes
        negative start samples, which wrap around in the reference's Python slices)."""
        if start_seconds > 0:
            raise ValueError("boundary lists need start_seconds <= 0 (negative start samples wrap around in Python slices)")
        torch = _native.require_gpu()
        track_of = np.asarray(track_of, dtype=np.int64).ravel()
        ratio = np.ascontiguousarray(ratio, dtype=np.float64).ravel()
        lens = _native.raster_lengths(self.end_max[track_of], ratio, sample_rate)
        counts = self.counts[track_of]
        caps = 2 * counts + 2
        offs, total = _layout(lens, 16 + 8 * caps)
        data = torch.empty(max(total, 64), dtype=torch.uint8, device="cuda")
        s_us, e_us, meta = self._dev if self._dev is not None else (self.start_us, self.end_us, self.meta)
        _native.rasterize_batch_runs(s_us, e_us, meta, self.firsts[track_of], counts, ratio, offs, caps, lens, data, sample_rate,
                                     float(start_seconds))
        return data, offs, lens, np.maximum(2 * counts, 2).astype(np.int32)


def rasterize_vectors(tracks, track_of, ratio, sample_rate: int = 100, start_seconds: float = 0):
    """``TrackSet(tracks).rasterize(track_of, ratio)``."""
    return TrackSet(tracks).rasterize(track_of, ratio, sample_rate, start_seconds)


def pairs_from_intervals(records, ratios: Sequence[float], sample_rate: int = 100, start_seconds: float = 0,
                         lists: bool = False) -> DeviceBatch:
    """Bit-packed DeviceBatch straight from subtitle interval lists: ``records`` is a list of
    (reference_track, candidate_track), a track being (start_us, end_us, is_metadata) arrays
    (``subtitle_raster.subtitle_records``).  The reference track is rasterised as it is, the candidate track once per
    framerate ratio (``SubtitleScaler`` + ``SubtitleSpeechTransformer``: times scaled by the ratio, amplitude
    min(1/ratio, 1)) -- all of it by ONE ``ffs_rasterize_batch_bits`` call that writes into the batch buffer (interval
    arithmetic on the device; no per-vector tensors, no pack copy).  Same vectors as ``pack_pairs`` over
    ``subtitle_raster.rasterize_candidates``.  ``lists``: the vectors as boundary lists (FFS_DTYPE_RUNS,
    ``ffs_rasterize_batch_runs``) -- no bitmap at all, and the solve needs no pass over one either; needs
    ``start_seconds <= 0``."""
    ratios = [float(r) for r in ratios]
    n_pairs, n_vec = len(records), 1 + len(ratios)
    tracks = [t for rec in records for t in rec]  # track 2p: pair p's reference, 2p + 1: its candidates
    track_of = np.tile(np.array([0] + [1] * len(ratios)), (n_pairs, 1)) + 2 * np.arange(n_pairs)[:, None]
    ratio = np.tile(np.array([1.0] + ratios), (n_pairs, 1))
    hi = np.minimum(1.0 / ratio, 1.0)
    if lists:
        if start_seconds > 0:
            raise ValueError("boundary lists need start_seconds <= 0 (negative start samples wrap around in Python slices)")
        data, offs, lens, bounds = TrackSet(tracks).rasterize_runs(track_of.ravel(), ratio.ravel(), sample_rate, start_seconds)
        return DeviceBatch(data, offs.reshape(n_pairs, n_vec), lens.reshape(n_pairs, n_vec), np.zeros_like(hi), hi,
                           _native.FFS_DTYPE_RUNS, None, bounds.reshape(n_pairs, n_vec))
    data, offs, lens = rasterize_vectors(tracks, track_of.ravel(), ratio.ravel(), sample_rate, start_seconds)
    return DeviceBatch(data, offs.reshape(n_pairs, n_vec), lens.reshape(n_pairs, n_vec), np.zeros_like(hi), hi,
                       _native.FFS_DTYPE_U1)


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of ceil(n/world) items per rank: [lo, hi)."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def pairs_in_flight_for(n_pairs: int, ceiling: int = 512) -> int:
    """Pairs per sweep of the transform kernels for a call of ``n_pairs`` problems: all of them up to 512 (measured
    plateau: 512 and 1024 pairs per sweep run at the same rate, profiles/r02_ab_experiments.json; fewer, larger sweeps
    have fewer launch tails), which also bounds the workspace at ~30 GiB for seven 2 h candidates."""
    return max(1, min(int(n_pairs), int(ceiling)))


class BatchAligner:
    """MaxScoreAligner(FFTAligner, None, sample_rate, max_offset_seconds) over a DeviceBatch."""

    def __init__(self, n_fft: int, n_cand: int, max_offset_samples: Optional[int] = 6000,
                 pairs_in_flight: int = 4, device: Optional[int] = None, streams: int = 1,
                 algorithm: Optional[str] = None) -> None:
        """``streams`` > 1: the pairs of a call are split into that many contiguous parts, each solved by its own
        plan (own workspace, ``pairs_in_flight`` shared between them) on its own HIP stream -- concurrent sub-batches
        fill each other's launch tails (+2-3 % solves/s at two streams, profiles/overlap_experiment.py).  The caller's
        current stream orders the whole call as before."""
        self.torch = _native.require_gpu()
        self.streams = max(1, int(streams))
        per_plan = max(1, (pairs_in_flight + self.streams - 1) // self.streams)
        self.plans = [_native.Plan(n_fft, per_plan, max(n_cand, 1), device) for _ in range(self.streams)]
        self.plan = self.plans[0]
        if algorithm is not None:  # "auto" | "fft" | "runs" (Plan.set_algorithm); None: the library default / FFS_ALGORITHM
            for p in self.plans:
                p.set_algorithm(algorithm)
        self._side = [self.torch.cuda.Stream(device=device) for _ in range(self.streams)] if self.streams > 1 else []
        self.n_cand = n_cand
        self.max_offset_samples = max_offset_samples

    def close(self) -> None:
        for p in self.plans:
            p.close()

    def solve_async(self, batch: DeviceBatch, pair_lo: int = 0, pair_hi: Optional[int] = None,
                    cand_out=None, pair_out=None):
        """Enqueue the solve of pairs [pair_lo, pair_hi) on the current stream; returns the two
        result tensors (uint8 views of ffs_cand_result / ffs_pair_result arrays)."""
        torch = self.torch
        pair_hi = batch.n_pairs if pair_hi is None else pair_hi
        n = pair_hi - pair_lo
        if cand_out is None:
            cand_out = torch.empty(max(n, 1) * self.n_cand * 24, dtype=torch.uint8, device=batch.data.device)
        if pair_out is None:
            pair_out = torch.empty(max(n, 1) * 24, dtype=torch.uint8, device=batch.data.device)
        def run(plan, lo, hi):
            sl = slice(lo, hi)
            ptrs = (batch.data.data_ptr() + batch.offs[sl]).astype(np.uint64)
            c0, p0 = (lo - pair_lo) * self.n_cand * 24, (lo - pair_lo) * 24
            dt = batch.call_dtype
            if batch.bounds is not None and isinstance(dt, (int, np.integer)):
                dt = (dt, dt)  # (the bounds travel through the per-vector-type entry point)
            plan.align_batch(hi - lo, self.n_cand, dt, ptrs.ravel(), batch.lens[sl].ravel(),
                             batch.lo[sl].ravel(), batch.hi[sl].ravel(), self.max_offset_samples,
                             self.max_offset_samples, cand_out[c0:], pair_out[p0:],
                             vec_max_boundaries=None if batch.bounds is None else batch.bounds[sl].ravel())

        if n > 0 and (self.streams == 1 or n < 2 * self.streams):
            run(self.plan, pair_lo, pair_hi)
        elif n > 0:
            main = torch.cuda.current_stream()
            start = torch.cuda.Event()
            start.record(main)
            for i, (plan, side) in enumerate(zip(self.plans, self._side)):
                lo, hi = shard_bounds(n, i, self.streams)
                if hi <= lo:
                    continue
                side.wait_event(start)
                with torch.cuda.stream(side):
                    run(plan, pair_lo + lo, pair_lo + hi)
                    done = torch.cuda.Event()
                    done.record(side)
                main.wait_event(done)
            for t in (batch.data, cand_out, pair_out):
                for side in self._side:
                    t.record_stream(side)  # the caching allocator must not recycle them under the side streams
        return cand_out, pair_out

    def solve(self, batch: DeviceBatch, pair_lo: int = 0, pair_hi: Optional[int] = None):
        cand_out, pair_out = self.solve_async(batch, pair_lo, pair_hi)
        pair_hi = batch.n_pairs if pair_hi is None else pair_hi
        n = pair_hi - pair_lo
        cres = cand_out.cpu().numpy().view(_native.CAND_RESULT_DTYPE)[: n * self.n_cand].reshape(n, self.n_cand)
        pres = pair_out.cpu().numpy().view(_native.PAIR_RESULT_DTYPE)[:n]
        return cres, pres


def gather_pair_results(local, n_total: int, world: int, group=None, comm: "Optional[_native.Comm]" = None):
    """All-gather per-pair results (one 24-byte record per pair) from every rank -- the only
    collective on the path.  ``local`` is a uint8 tensor holding this rank's ceil(n/world) records
    (zero-padded); returns a uint8 tensor with all n_total records, in pair order.  With ``comm`` the
    gather is the library's own ``ffs_gather_results`` (RCCL C API); otherwise torch.distributed's
    all_gather_into_tensor on ``group`` (RCCL on GPUs, gloo in the CPU tests)."""
    import torch

    per = (n_total + world - 1) // world
    assert local.numel() == per * 24
    if comm is not None:
        out = comm.gather_pair_results(local)
    else:
        import torch.distributed as dist

        out = torch.empty(world * per * 24, dtype=torch.uint8, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
    return out[: n_total * 24]


def make_comm(rank: int, world: int, group=None) -> "_native.Comm":
    """``ffs_comm_create`` bootstrapped over an initialised torch.distributed group: rank 0 creates the 128-byte
    RCCL unique id and broadcasts it (public API only; a fresh id per call, so communicators can be created
    repeatedly in one process group), everyone joins."""
    import torch.distributed as dist

    box = [_native.Comm.unique_id() if rank == 0 else None]
    # `rank` is the caller's rank WITHIN `group`; broadcast's `src` is a global rank
    src = 0 if group is None else dist.get_global_rank(group, 0)
    dist.broadcast_object_list(box, src=src, group=group)
    return _native.Comm(rank, world, bytes(box[0]))
