"""Seeded synthetic workloads for the alignment hot path.

Speech-like activity vectors at 100 Hz and the subtitle tracks derived from them, rasterised with
the reference's own conventions (SubtitleScaler + SubtitleSpeechTransformer,
ffsubsync/subtitle_transformers.py:35-47, speech_transformers.py:957-980): start =
int(round(t0*sr)), end = start + int(round(dur*sr)), length int(max_end*sr)+2, amplitude
min(1/ratio, 1).  Run lists come from ``numpy.random.RandomState(seed)`` on the host; only the
rasterisation runs where the data is needed (numpy, or torch on the GPU for the benchmark batch).
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from ffsubsync_amd.constants import SAMPLE_RATE, candidate_ratios


@dataclass
class PairSpec:
    """Interval lists (frame indices) of one (reference, candidates) problem."""

    seed: int
    ref_len: int
    ref_starts: np.ndarray
    ref_ends: np.ndarray
    cand_len: List[int]
    cand_starts: List[np.ndarray]
    cand_ends: List[np.ndarray]
    cand_amp: List[float]
    ratios: List[float]
    true_offset_samples: int
    true_ratio_index: int


def _speech_runs(rng: np.random.RandomState, duration_s: float, run_scale: float = 1.0):
    """Alternate gaps U[0.2, 8] s and speech runs U[0.5, 6] s (times ``run_scale``) until the duration is used up."""
    n_max = int(duration_s / (0.7 * run_scale)) + 8
    gaps = rng.uniform(0.2, 8.0, n_max) * run_scale
    runs = rng.uniform(0.5, 6.0, n_max) * run_scale
    ends = np.cumsum(gaps + runs)
    starts = ends - runs
    keep = ends < duration_s
    return starts[keep], ends[keep]


def make_pair_spec(seed: int, duration_s: float = 7200.0, ratios: Optional[Sequence[float]] = None,
                   max_true_offset_s: float = 55.0, sample_rate: int = SAMPLE_RATE, run_scale: float = 1.0) -> PairSpec:
    """``run_scale`` < 1 shortens gaps, runs and the edge jitter alike: 1 / run_scale times as many runs per vector (a
    flickering frame-level detector instead of subtitle-like activity); 1.0 is the benchmark workload."""
    ratios = list(candidate_ratios() if ratios is None else ratios)
    rng = np.random.RandomState(seed)
    t0, t1 = _speech_runs(rng, duration_s, run_scale)
    ref_len = int(round(duration_s * sample_rate))
    rs = np.rint(t0 * sample_rate).astype(np.int64)
    re = np.minimum(rs + np.rint((t1 - t0) * sample_rate).astype(np.int64), ref_len)
    true_offset_s = float(np.round(rng.uniform(-max_true_offset_s, max_true_offset_s), 2))
    true_idx = int(rng.randint(len(ratios)))
    true_ratio = ratios[true_idx]
    keep = rng.rand(t0.size) > 0.15
    js = rng.uniform(-0.1, 0.1, t0.size) * run_scale
    je = rng.uniform(-0.1, 0.1, t0.size) * run_scale
    # subtitle clock: t_ref = t_sub * true_ratio + true_offset
    s0 = (t0 + js - true_offset_s) / true_ratio
    s1 = (t1 + je - true_offset_s) / true_ratio
    ok = keep & (s0 >= 0.0) & (s1 > s0)
    s0, s1 = s0[ok], s1[ok]
    cand_len, cand_starts, cand_ends, cand_amp = [], [], [], []
    for ratio in ratios:
        a0, a1 = s0 * ratio, s1 * ratio  # SubtitleScaler (subtitle_transformers.py:35-47)
        n = int(a1.max() * sample_rate) + 2 if a1.size else 2  # speech_transformers.py:962
        st = np.rint(a0 * sample_rate).astype(np.int64)
        en = np.minimum(st + np.rint((a1 - a0) * sample_rate).astype(np.int64), n)
        cand_len.append(n)
        cand_starts.append(st)
        cand_ends.append(en)
        cand_amp.append(min(1.0 / ratio, 1.0))  # speech_transformers.py:977
    return PairSpec(seed, ref_len, rs, re, cand_len, cand_starts, cand_ends, cand_amp, ratios,
                    int(round(true_offset_s * sample_rate)), true_idx)


def rasterize(n: int, starts: np.ndarray, ends: np.ndarray) -> np.ndarray:
    """0/1 uint8 vector with [start, end) set for every interval (union on overlap)."""
    delta = np.zeros(n + 1, dtype=np.int32)
    np.add.at(delta, np.clip(starts, 0, n), 1)
    np.add.at(delta, np.clip(ends, 0, n), -1)
    return (np.cumsum(delta[:-1]) > 0).astype(np.uint8)


def pair_arrays(spec: PairSpec):
    """(ref01 uint8, [cand01 uint8 ...]) on the host."""
    ref = rasterize(spec.ref_len, spec.ref_starts, spec.ref_ends)
    cands = [rasterize(n, s, e) for n, s, e in zip(spec.cand_len, spec.cand_starts, spec.cand_ends)]
    return ref, cands


def pair_float_arrays(spec: PairSpec):
    """The float64 arrays the reference pipeline would hand the aligner: reference 0/1,
    candidate j in {0, min(1/ratio_j, 1)}."""
    ref, cands = pair_arrays(spec)
    return ref.astype(float), [c.astype(float) * a for c, a in zip(cands, spec.cand_amp)]


def fused_reference(spec: PairSpec, w_silero: float = 0.6, w_webrtc: float = 0.4) -> np.ndarray:
    """The reference of ``spec`` as the `weighted` fused detector would hand it over
    (ffsubsync/speech_transformers.py:290-293: 0.6 * silero + 0.4 * webrtc of two 0/1 label vectors): the spec's own
    speech runs stand in for one detector, the same runs with every edge moved by up to +-15 frames and a tenth of them
    missed for the other -- two detectors that mostly agree.  float64, levels {0, 0.4, 0.6, 1}; seeded by the spec."""
    rng = np.random.RandomState(spec.seed + 77001)
    n = spec.ref_starts.size
    keep = rng.rand(n) > 0.1
    js, je = rng.randint(-15, 16, n), rng.randint(-15, 16, n)
    a = rasterize(spec.ref_len, spec.ref_starts, spec.ref_ends).astype(np.float64)
    st, en = (spec.ref_starts + js)[keep], (spec.ref_ends + je)[keep]
    ok = en > st
    b = rasterize(spec.ref_len, st[ok], en[ok]).astype(np.float64)
    return w_silero * a + w_webrtc * b


def simple_pair(n_ref: int, n_sub: int, offset: int, seed: int = 0, density: float = 0.4, flip: float = 0.05):
    """A random 0/1 reference and a noisy copy of a window of it, so that the best offset is
    ``offset``: sub[i] ~ ref[i + offset].  (BASELINE config 1: n=60000, offset=+3720.)"""
    rng = np.random.RandomState(seed)
    # piecewise-constant 'speech' so neighbouring lags are not independent
    seg = np.maximum(1, rng.geometric(1.0 / 150.0, size=n_ref // 50 + 16))
    vals = (rng.rand(seg.size) < density).astype(np.uint8)
    ref = np.repeat(vals, seg)[:n_ref]
    if ref.size < n_ref:
        ref = np.concatenate([ref, np.zeros(n_ref - ref.size, np.uint8)])
    idx = np.arange(n_sub) + offset
    ok = (idx >= 0) & (idx < n_ref)
    sub = np.zeros(n_sub, np.uint8)
    sub[ok] = ref[idx[ok]]
    noise = rng.rand(n_sub) < flip
    sub = np.where(noise, 1 - sub, sub).astype(np.uint8)
    return ref, sub


def make_subtitle_records(seed: int, duration_s: float = 5400.0, mean_gap_s: float = 3.0, mean_dur_s: float = 2.7):
    """Seeded subtitle file as (start_us, end_us, is_metadata) with millisecond stamps (srt
    resolution): alternating gaps U[0.2, 2*mean_gap] and lines U[0.4, 2*mean_dur] up to the duration."""
    rng = np.random.RandomState(seed)
    n_max = int(duration_s / 0.6) + 8
    gaps = rng.uniform(0.2, 2.0 * mean_gap_s, n_max)
    durs = rng.uniform(0.4, 2.0 * mean_dur_s, n_max)
    ends = np.cumsum(gaps + durs)
    keep = ends < duration_s
    ends, durs = ends[keep], durs[keep]
    start_us = np.rint((ends - durs) * 1e3).astype(np.int64) * 1000
    end_us = np.rint(ends * 1e3).astype(np.int64) * 1000
    return start_us, end_us, np.zeros(start_us.size, dtype=np.uint8)


def build_device_batch(specs: Sequence[PairSpec], device=None, chunk_pairs: int = 32, packed: bool = True):
    """The specs' vectors rasterised straight into HBM with torch ops (index_add + cumsum) as one
    ``ffsubsync_amd.batch.DeviceBatch``: bit-packed (FFS_DTYPE_U1, the library's native format; the bytes
    of each chunk of pairs are packed as soon as they are rasterised) unless ``packed`` is False (0/1 bytes)."""
    import torch

    from ffsubsync_amd import _native
    from ffsubsync_amd.batch import DeviceBatch

    _native.require_gpu()
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
    data = torch.zeros(total // 8 if packed else total, dtype=torch.uint8, device=device)
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
        chunk = (torch.cumsum(delta[:-1], 0) > 0).to(torch.uint8)
        del delta
        if packed:  # vector offsets are multiples of 64 bytes = 512 samples = 16 words
            _native.pack_bits(chunk, out=data[base // 8: end // 8].view(torch.int32))
        else:
            data[base:end] = chunk
        del chunk
    if packed:
        return DeviceBatch(data, offs // 8, lens, lo, hi, _native.FFS_DTYPE_U1)
    return DeviceBatch(data, offs, lens, lo, hi, _native.FFS_DTYPE_U8)


def build_fused_batch(specs: Sequence[PairSpec], device=None):
    """The specs as a mixed-type ``DeviceBatch``: candidates bit-packed (FFS_DTYPE_U1) exactly as ``build_device_batch``
    lays them out, references the four-level float64 vectors of ``fused_reference`` (FFS_DTYPE_F64) appended to the same
    buffer -- what a pipeline with the weighted fused VAD hands the aligner."""
    import torch

    from ffsubsync_amd import _native
    from ffsubsync_amd.batch import DeviceBatch

    db = build_device_batch(specs, device=device, packed=True)
    refs = [fused_reference(sp) for sp in specs]
    sizes = np.array([(r.size * 8 + 63) // 64 * 64 for r in refs], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    host = np.zeros(int(sizes.sum()), dtype=np.uint8)
    for r, o in zip(refs, starts):
        host[o:o + r.size * 8] = r.view(np.uint8)
    base = (db.data.numel() + 63) // 64 * 64
    data = torch.zeros(base + host.size, dtype=torch.uint8, device=db.data.device)
    data[: db.data.numel()] = db.data
    data[base:] = torch.from_numpy(host).to(db.data.device)
    offs, lo, hi = db.offs.copy(), db.lo.copy(), db.hi.copy()
    offs[:, 0] = base + starts
    lo[:, 0], hi[:, 0] = 0.0, 1.0
    return DeviceBatch(data, offs, db.lens, lo, hi, _native.FFS_DTYPE_U1, ref_dtype=_native.FFS_DTYPE_F64)
