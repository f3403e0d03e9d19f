"""Deterministic inputs shared by tests/golden/make_golden.py (which runs the UNMODIFIED reference
on them, in the build container) and by the parity tests (which run the oracle and the HIP path
on the same inputs and compare with the committed expectations)."""
import hashlib

import numpy as np

from workloads import synth
from ffsubsync_amd.constants import FRAMERATE_RATIOS, candidate_ratios

SR = 100


def scaled(sub, sf):
    """Array-domain emulation of SubtitleScaler (same construction as the reference's
    tests/test_multi_segment.py:113-120): out[k] = sub[round(k / sf)]."""
    out = np.zeros(int(len(sub) * sf) + 2)
    k = np.arange(len(out))
    src = np.round(k / sf).astype(int)
    ok = src < len(sub)
    out[k[ok]] = sub[src[ok]]
    return out


def sparse_case(true_scale, true_shift):
    """reference tests/test_multi_segment.py:136-153 -- RandomState(13), n_sub=24000."""
    rng = np.random.RandomState(13)
    n_sub = 24000
    n_ref = int(true_scale * n_sub + abs(true_shift) * SR) + 2000
    ref_full = (rng.rand(n_ref) > 0.6).astype(float)
    m = np.arange(n_sub)
    idx = np.round(true_scale * m + true_shift * SR).astype(int)
    sub = np.zeros(n_sub)
    ok = (idx >= 0) & (idx < n_ref)
    sub[m[ok]] = ref_full[idx[ok]]
    cands = [scaled(sub, sf) for sf in candidate_ratios()]
    return ref_full, cands


def digest(arrays):
    h = hashlib.sha1()
    for a in arrays:
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def build_cases(include_large=True):
    """name -> dict(ref=array|str, cands=[array|str...], max_offset=int|None).
    Every candidate is solved with FFTAligner(max_offset_samples=max_offset) and the set with
    MaxScoreAligner(FFTAligner(max_offset_samples=max_offset))."""
    cases = {}
    # 1. reference KATs (tests/test_alignment.py:7-14): fit_transform(s2, s1)
    for i, (s1, s2) in enumerate([("111001", "11001"), ("1001", "1001"), ("10010", "01001")]):
        cases["kat%d" % i] = dict(ref=s2, cands=[s1], max_offset=None)
    # 2. sparse scale/shift recovery (tests/test_multi_segment.py:136-167)
    for i, (sc, sh) in enumerate([(1.0, 5.0), (1.0, -8.0), (25.0 / 24.0, 3.0), (24.0 / 25.0, -2.0)]):
        ref, cands = sparse_case(sc, sh)
        cases["sparse%d" % i] = dict(ref=ref, cands=cands, max_offset=60 * SR)
    # 3. BASELINE config 1: 10 min pair, +37.2 s
    ref, sub = synth.simple_pair(60000, 60000, 3720, seed=1)
    cases["config1_none"] = dict(ref=ref, cands=[sub], max_offset=None)
    cases["config1_6000"] = dict(ref=ref, cands=[sub], max_offset=6000)
    # 4. lag-window semantics (aligners.py:31-43)
    ref, sub = synth.simple_pair(5000, 4800, -57, seed=2)
    cases["mask100"] = dict(ref=ref, cands=[sub], max_offset=100)
    cases["mask40_truth_outside"] = dict(ref=ref, cands=[sub], max_offset=40)
    ref, sub = synth.simple_pair(3000, 3000, 11, seed=3)
    cases["mask_negative_index"] = dict(ref=ref, cands=[sub], max_offset=6000)
    cases["mask_all"] = dict(ref=ref, cands=[sub], max_offset=0)
    ref, sub = synth.simple_pair(700, 300, 123, seed=9, flip=0.02)
    cases["short_direct"] = dict(ref=ref, cands=[sub], max_offset=None)
    cases["short_direct_w"] = dict(ref=ref, cands=[sub], max_offset=150)
    # 5. production-shaped two-level candidates (amplitude 1/ratio), 10 minutes
    spec = synth.make_pair_spec(5, duration_s=600.0)
    ref, cands = synth.pair_float_arrays(spec)
    cases["pipeline_10min"] = dict(ref=ref, cands=cands, max_offset=6000)
    # 6. non-two-level float inputs ("weighted" fused VAD levels, speech_transformers.py:290-293)
    rng = np.random.RandomState(6)
    lv = np.array([0.0, 0.4, 0.6, 1.0])
    ref = np.repeat(lv[rng.randint(0, 4, 400)], 25)
    sub = np.concatenate([np.zeros(37), ref[:8000]]) * 0.96
    sub2 = sub[5:].copy()
    sub2[1000:1600] = 0.0  # a distinctly worse second candidate (no score tie between candidates)
    cases["float_levels"] = dict(ref=ref, cands=[sub, sub2], max_offset=None)
    # 7. non-default non_speech_label (reference values {-1, 1} before the 2x-1 map)
    ref, sub = synth.simple_pair(20000, 18000, -250, seed=7)
    cases["label_minus1"] = dict(ref=2.0 * ref - 1.0, cands=[sub.astype(float)], max_offset=None)
    if include_large:
        # 8. headline shape: 2 h @ 100 Hz, 7 ratios, max_offset 6000
        for seed in (0, 1):
            spec = synth.make_pair_spec(seed)
            ref, cands = synth.pair_float_arrays(spec)
            cases["pipeline_2h_seed%d" % seed] = dict(ref=ref, cands=cands, max_offset=6000)
        spec = synth.make_pair_spec(0)
        ref, cands = synth.pair_float_arrays(spec)
        cases["single_2h_none"] = dict(ref=ref, cands=[cands[spec.true_ratio_index]], max_offset=None)
    return cases


class ScaledPipe:
    """Stand-in for the subtitle pipeline a gss callable returns: fit_transform(srtin) gives the
    track rescaled by `ratio` (array domain)."""

    def __init__(self, sub, ratio):
        self.sub, self.ratio = sub, ratio

    def fit_transform(self, *_):
        return scaled(self.sub, self.ratio)

    fit = fit_transform


def gss_case():
    ref, cands = sparse_case(25.0 / 24.0, 3.0)
    rng = np.random.RandomState(13)
    n_sub = 24000
    n_ref = int(25.0 / 24.0 * n_sub + 3.0 * SR) + 2000
    ref_full = (rng.rand(n_ref) > 0.6).astype(float)
    m = np.arange(n_sub)
    idx = np.round(25.0 / 24.0 * m + 3.0 * SR).astype(int)
    sub = np.zeros(n_sub)
    ok = (idx >= 0) & (idx < n_ref)
    sub[m[ok]] = ref_full[idx[ok]]
    return ref_full, sub
