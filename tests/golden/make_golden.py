"""Generate tests/golden/aligner_golden.json by running the UNMODIFIED reference aligner
(/root/reference/ffsubsync/aligners.py, golden_section_search.py, sklearn_shim.py) on the inputs
of tests/golden_cases.py.  Runs only in the build container (needs /root/reference); the JSON it
writes is committed and is what travels to the GPU box.

    python tests/golden/make_golden.py
"""
import json
import logging
import os
import sys

sys.dont_write_bytecode = True  # importing the reference must not write __pycache__ into /root/reference
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# import the reference modules without executing ffsubsync/__init__.py (needs ffmpeg/srt/pysubs2)
_pkg = types.ModuleType("ffsubsync")
_pkg.__path__ = ["/root/reference/ffsubsync"]
sys.modules["ffsubsync"] = _pkg
logging.disable(logging.INFO)
from ffsubsync.aligners import FailedToFindAlignmentException, FFTAligner, MaxScoreAligner  # noqa: E402
from ffsubsync.golden_section_search import gss  # noqa: E402

import golden_cases  # noqa: E402


def fnum(x):
    x = float(x)
    return "-inf" if x == float("-inf") else repr(x)


def main():
    out = {"_generator": "tests/golden/make_golden.py", "_reference": "smacke/ffsubsync @ /root/reference (v0.5.0)",
           "_numpy": np.__version__, "cases": {}}
    for name, c in golden_cases.build_cases().items():
        per = []
        for cand in c["cands"]:
            score, offset = FFTAligner(max_offset_samples=c["max_offset"]).fit_transform(c["ref"], cand, get_score=True)
            # uniqueness of the winner (offsets are only defined when the top-2 gap > 0.5)
            al = FFTAligner(max_offset_samples=c["max_offset"])
            r_, s_ = [list(map(int, x)) if isinstance(x, str) else x for x in (c["ref"], cand)]
            r_, s_ = 2 * np.array(r_).astype(float) - 1, 2 * np.array(s_).astype(float) - 1
            n_ = int(2 ** np.ceil(np.log2(len(r_) + len(s_))))
            conv = np.real(np.fft.ifft(np.fft.fft(np.append(np.zeros(n_ - len(s_)), s_)) *
                                       np.fft.fft(np.flip(np.append(r_, np.zeros(n_ - len(r_))), 0))))
            m = al._eliminate_extreme_offsets_from_solutions(conv, s_)
            fin = np.sort(m[np.isfinite(m)])
            gap = float(fin[-1] - fin[-2]) if fin.size >= 2 else float("inf")
            assert gap > 0.5, (name, gap)
            per.append({"score": fnum(score), "offset": int(offset), "top2_gap": repr(gap)})
        entry = {"max_offset": c["max_offset"], "per_candidate": per,
                 "digest": golden_cases.digest([np.array(list(map(int, c["ref"]))) if isinstance(c["ref"], str) else c["ref"]]
                                                 + [np.array(list(map(int, s))) if isinstance(s, str) else s for s in c["cands"]])}
        try:
            (score, offset), winner = MaxScoreAligner(FFTAligner(max_offset_samples=c["max_offset"])).fit_transform(
                c["ref"], list(c["cands"]))
            # MaxScoreAligner built from an instance has max_offset_samples None: no filtering
            idx = next(i for i, s in enumerate(c["cands"]) if s is winner)
            entry["best_unfiltered"] = {"score": fnum(score), "offset": int(offset), "index": idx}
        except FailedToFindAlignmentException as e:
            entry["best_unfiltered"] = {"raises": str(e)[:40]}
        if c["max_offset"] is not None and c["max_offset"] % SR100 == 0:
            try:
                (score, offset), winner = MaxScoreAligner(FFTAligner, None, 100, c["max_offset"] // 100).fit_transform(
                    c["ref"], list(c["cands"]))
                idx = next(i for i, s in enumerate(c["cands"]) if s is winner)
                entry["best_filtered"] = {"score": fnum(score), "offset": int(offset), "index": idx}
            except FailedToFindAlignmentException as e:
                entry["best_filtered"] = {"raises": str(e)[:40]}
        out["cases"][name] = entry
        print(name, entry["per_candidate"][:2], entry.get("best_filtered", entry["best_unfiltered"]))

    # empty inputs (reference tests/test_alignment.py:17-27)
    empties = []
    for r, s in [([], [1, 0, 1]), ([1, 0, 1], []), ([], [])]:
        try:
            FFTAligner().fit(np.array(r), np.array(s))
            empties.append("no-raise")
        except FailedToFindAlignmentException as e:
            empties.append(str(e))
    out["empty_messages"] = empties

    # golden-section search (aligners.py:111-129 + golden_section_search.py)
    ref, sub = golden_cases.gss_case()
    trace = []

    def maker(ratio):
        trace.append(float(ratio))
        return golden_cases.ScaledPipe(sub, ratio)

    msa = MaxScoreAligner(FFTAligner(max_offset_samples=6000))
    msa.fit(ref, [maker])
    (score, offset), pipe = msa.transform()
    out["gss"] = {"ratios": [repr(t) for t in trace], "final_ratio": repr(pipe.ratio), "score": fnum(score),
                  "offset": int(offset), "n_scores": len(msa._scores)}
    # (a one-argument objective crashes the reference at golden_section_search.py:69, so use two)
    g = gss(lambda x, last: (x - 2) ** 2, 1, 5, 1e-5)
    out["gss_doc_example"] = [repr(g[0]), repr(g[1])]
    print("gss", out["gss"]["final_ratio"], out["gss"]["score"], out["gss"]["offset"], len(trace))

    with open(os.path.join(HERE, "aligner_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


SR100 = 100
if __name__ == "__main__":
    main()
