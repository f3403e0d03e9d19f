"""Batched, device-resident driver for many (reference, candidates) alignment problems.

This is the throughput path behind the headline metric (seven-ratio MaxScoreAligner solves per
second): all activity vectors of a batch live back to back in one uint8 HBM buffer, one call into
``ffs_align_batch`` solves every problem, and across GPUs the problems are sharded by pair (no
data exchange during solves) with one all-gather of the 24-byte results over RCCL.
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from .synth import PairSpec


@dataclass
class DeviceBatch:
    """Packed two-level activity vectors in HBM.  Row p of the descriptor arrays holds pair p's
    reference followed by its candidates."""

    data: "object"  # torch.uint8 CUDA tensor
    offs: np.ndarray  # [n_pairs, 1+n_cand] byte offsets into data
    lens: np.ndarray  # [n_pairs, 1+n_cand]
    lo: np.ndarray
    hi: np.ndarray

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
        return DeviceBatch(self.data, pick(self.offs), pick(self.lens), pick(self.lo), pick(self.hi))

    def required_fft_length(self, max_offset_samples: Optional[int] = None) -> int:
        """Plan length for the whole batch: the reference's N = 2^ceil(log2(R+S)) without a lag window,
        the alias-free (possibly shorter) length with one (``_native.plan_length``)."""
        n = 2
        for p in range(self.n_pairs):
            for j in range(1, self.offs.shape[1]):
                n = max(n, _native.plan_length(int(self.lens[p, 0]), int(self.lens[p, j]), max_offset_samples))
        return n


def pack_pairs(pairs) -> DeviceBatch:
    """DeviceBatch from HBM-resident two-level vectors: ``pairs`` is a list of
    (reference, [candidates]) whose items expose ``.bits`` (uint8 CUDA tensor), ``.lo`` and ``.hi``
    (e.g. ``subtitle_raster.DeviceRaster``).  The vectors are copied device-to-device into one
    buffer at 64-byte aligned offsets."""
    torch = _native.require_gpu()
    n_pairs, n_vec = len(pairs), 1 + len(pairs[0][1])
    flat = []
    for ref, cands in pairs:
        if len(cands) != n_vec - 1:
            raise ValueError("all pairs need the same number of candidates")
        flat.append(ref)
        flat.extend(cands)
    lens = np.array([int(v.bits.numel()) for v in flat], dtype=np.int64).reshape(n_pairs, n_vec)
    padded = (lens + 63) // 64 * 64
    offs = np.concatenate([[0], np.cumsum(padded.ravel())[:-1]]).reshape(n_pairs, n_vec).astype(np.int64)
    data = torch.zeros(int(padded.sum()), dtype=torch.uint8, device=flat[0].bits.device)
    for v, o in zip(flat, offs.ravel()):
        data[int(o): int(o) + int(v.bits.numel())] = v.bits
    lo = np.array([v.lo for v in flat], dtype=np.float64).reshape(n_pairs, n_vec)
    hi = np.array([v.hi for v in flat], dtype=np.float64).reshape(n_pairs, n_vec)
    return DeviceBatch(data, offs, lens, lo, hi)


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of ceil(n/world) items per rank: [lo, hi)."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def build_device_batch(specs: Sequence[PairSpec], device=None, chunk_pairs: int = 32) -> DeviceBatch:
    """Rasterise interval lists straight into HBM with torch ops (index_add + cumsum)."""
    torch = _native.require_gpu()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    n_pairs = len(specs)
    n_vec = 1 + len(specs[0].cand_len)
    lens = np.zeros((n_pairs, n_vec), dtype=np.int64)
    lo = np.zeros((n_pairs, n_vec), dtype=np.float64)
    hi = np.ones((n_pairs, n_vec), dtype=np.float64)
    for p, sp in enumerate(specs):
        lens[p, 0] = sp.ref_len
        lens[p, 1:] = sp.cand_len
        hi[p, 1:] = sp.cand_amp
    padded = (lens + 63) // 64 * 64
    offs = np.concatenate([[0], np.cumsum(padded.ravel())[:-1]]).reshape(n_pairs, n_vec).astype(np.int64)
    total = int(padded.sum())
    data = torch.zeros(total, dtype=torch.uint8, device=device)
    for p0 in range(0, n_pairs, chunk_pairs):
        p1 = min(p0 + chunk_pairs, n_pairs)
        base = int(offs[p0, 0])
        end = int(offs[p1 - 1, -1] + padded[p1 - 1, -1])
        starts, ends = [], []
        for p in range(p0, p1):
            sp = specs[p]
            for v, (s, e) in enumerate([(sp.ref_starts, sp.ref_ends)] + list(zip(sp.cand_starts, sp.cand_ends))):
                n = int(lens[p, v])
                o = int(offs[p, v]) - base
                starts.append(np.clip(s, 0, n) + o)
                ends.append(np.clip(e, 0, n) + o)
        starts = torch.from_numpy(np.concatenate(starts)).to(device)
        ends = torch.from_numpy(np.concatenate(ends)).to(device)
        delta = torch.zeros(end - base + 1, dtype=torch.int32, device=device)
        delta.index_add_(0, starts, torch.ones_like(starts, dtype=torch.int32))
        delta.index_add_(0, ends, -torch.ones_like(ends, dtype=torch.int32))
        data[base:end] = (torch.cumsum(delta[:-1], 0) > 0).to(torch.uint8)
        del delta
    return DeviceBatch(data, offs, lens, lo, hi)


class BatchAligner:
    """MaxScoreAligner(FFTAligner, None, sample_rate, max_offset_seconds) over a DeviceBatch."""

    def __init__(self, n_fft: int, n_cand: int, max_offset_samples: Optional[int] = 6000,
                 pairs_in_flight: int = 4, device: Optional[int] = None) -> None:
        self.torch = _native.require_gpu()
        self.plan = _native.Plan(n_fft, pairs_in_flight, max(n_cand, 1), device)
        self.n_cand = n_cand
        self.max_offset_samples = max_offset_samples

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
        if n > 0:
            sl = slice(pair_lo, pair_hi)
            ptrs = (batch.data.data_ptr() + batch.offs[sl]).astype(np.uint64)
            self.plan.align_batch(n, self.n_cand, _native.FFS_DTYPE_U8, ptrs.ravel(), batch.lens[sl].ravel(),
                                  batch.lo[sl].ravel(), batch.hi[sl].ravel(), self.max_offset_samples,
                                  self.max_offset_samples, cand_out, pair_out)
        return cand_out, pair_out

    def solve(self, batch: DeviceBatch, pair_lo: int = 0, pair_hi: Optional[int] = None):
        cand_out, pair_out = self.solve_async(batch, pair_lo, pair_hi)
        pair_hi = batch.n_pairs if pair_hi is None else pair_hi
        n = pair_hi - pair_lo
        cres = cand_out.cpu().numpy().view(_native.CAND_RESULT_DTYPE)[: n * self.n_cand].reshape(n, self.n_cand)
        pres = pair_out.cpu().numpy().view(_native.PAIR_RESULT_DTYPE)[:n]
        return cres, pres


def gather_pair_results(local, n_total: int, world: int, group=None):
    """All-gather per-pair results (one 24-byte record per pair) from every rank -- the only
    collective on the path.  ``local`` is a uint8 tensor holding this rank's ceil(n/world) records
    (zero-padded); returns a uint8 tensor with all n_total records, in pair order."""
    import torch
    import torch.distributed as dist

    per = (n_total + world - 1) // world
    assert local.numel() == per * 24
    out = torch.empty(world * per * 24, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out[: n_total * 24]
