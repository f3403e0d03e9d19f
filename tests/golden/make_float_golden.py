"""Generate tests/golden/float_golden.json: the UNMODIFIED reference aligner on 2 h x 7-ratio problems whose
REFERENCE vector is four-level -- what the `weighted` fused VAD emits ({0, 0.4, 0.6, 1},
/root/reference/ffsubsync/speech_transformers.py:290-293) -- and whose candidates are the subtitle rasters with
amplitude min(1/ratio, 1) (:977).  MaxScoreAligner(FFTAligner, None, 100, 60), i.e. the production window, plus
FFTAligner(None) on the true-ratio candidate.  Inputs are regenerated from the seed on the GPU box
(workloads.synth.make_pair_spec + fused_reference); only the reference's answers are committed.

Runs only in the build container (needs /root/reference).

    python tests/golden/make_float_golden.py [n_pairs=12] [procs=4]
"""
import json
import logging
import multiprocessing as mp
import os
import sys

sys.dont_write_bytecode = True  # importing the reference must not write __pycache__ into /root/reference
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

SEED0 = 5000  # float problems use their own seed range


def _reference():
    if "ffsubsync.aligners" not in sys.modules:
        pkg = types.ModuleType("ffsubsync")
        pkg.__path__ = ["/root/reference/ffsubsync"]
        sys.modules["ffsubsync"] = pkg
        logging.disable(logging.INFO)
    from ffsubsync.aligners import FFTAligner, MaxScoreAligner

    return FFTAligner, MaxScoreAligner


def fnum(x):
    x = float(x)
    return "-inf" if x == float("-inf") else repr(x)


def problem(seed):
    """(reference float64 four-level, [seven candidate float64 arrays], spec) of one float problem."""
    from workloads import synth

    spec = synth.make_pair_spec(seed)
    _, cands = synth.pair_float_arrays(spec)
    return synth.fused_reference(spec), cands, spec


def _solve(seed):
    FFTAligner, MaxScoreAligner = _reference()
    ref, cands, spec = problem(seed)
    assert sorted(set(np.unique(ref).tolist())) == [0.0, 0.4, 0.6, 1.0], np.unique(ref)
    msa = MaxScoreAligner(FFTAligner, None, 100, 60)
    (score, offset), winner = msa.fit_transform(ref, list(cands))
    idx = next(i for i, c in enumerate(cands) if c is winner)
    per = [[fnum(s), int(o)] for (s, o), _ in msa._scores]
    gaps = []  # top-2 gap of every candidate's masked `convolve` (same arithmetic as aligners.py:55-78)
    for c in cands:
        al = FFTAligner(6000)
        r_, s_ = 2 * np.asarray(ref, dtype=float) - 1, 2 * np.asarray(c, dtype=float) - 1
        n_ = int(2 ** np.ceil(np.log2(len(r_) + len(s_))))
        conv = np.real(np.fft.ifft(np.fft.fft(np.append(np.zeros(n_ - len(s_)), s_)) *
                                   np.fft.fft(np.flip(np.append(r_, np.zeros(n_ - len(r_))), 0))))
        m = al._eliminate_extreme_offsets_from_solutions(conv, s_)
        top = np.partition(m[np.isfinite(m)], -2)[-2:]
        gaps.append(float(top[1] - top[0]))
    s_none, o_none = FFTAligner(None).fit_transform(ref, cands[spec.true_ratio_index], get_score=True)
    return {"seed": seed, "index": idx, "offset": int(offset), "score": fnum(score), "per_candidate": per,
            "per_candidate_top2_gap": [round(g, 9) for g in gaps], "single_none": [fnum(s_none), int(o_none)],
            "true_ratio_index": spec.true_ratio_index, "true_offset_samples": spec.true_offset_samples}


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_solve, range(SEED0, SEED0 + n_pairs), chunksize=1)
    out = {
        "_generator": "tests/golden/make_float_golden.py",
        "_reference": "smacke/ffsubsync @ /root/reference (v0.5.0), unmodified aligners.py",
        "_numpy": np.__version__,
        "_workload": "reference = workloads.synth.fused_reference(make_pair_spec(seed)) (levels 0/0.4/0.6/1), candidates = "
                     "pair_float_arrays(spec)[1] (levels 0 / min(1/ratio, 1)); MaxScoreAligner(FFTAligner, None, 100, 60)",
        "pairs": res,
    }
    with open(os.path.join(HERE, "float_golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"), sort_keys=True)
    print("wrote %d pairs; winner gaps: %s" % (len(res), [r["per_candidate_top2_gap"][r["index"]] for r in res]))


if __name__ == "__main__":
    main()
