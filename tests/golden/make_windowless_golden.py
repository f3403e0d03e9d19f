"""Generate tests/golden/windowless_golden.json: the UNMODIFIED reference aligner
(/root/reference/ffsubsync/aligners.py) run WITHOUT a lag window on the benchmark's own inputs --
``MaxScoreAligner(FFTAligner())`` (the default constructor, aligners.py:25-29; a base-aligner INSTANCE, so
aligners.py:105-106 keeps its ``max_offset_samples=None`` and aligners.py:156 filters nothing) over the seven
framerate-ratio candidates of seeds 0..N-1 of workloads.synth.make_pair_spec (2 h @ 100 Hz).

Every lag of every wrong-ratio candidate is eligible here, so this is where plateau ties and the zero-overlap
rule bite; the round-5 library moved exactly this configuration from the transforms onto 118 lag tiles of the
run-boundary path.  Per candidate the generator also records the top-2 gap of the reference's own ``convolve``
(an offset is only defined where it exceeds 0.5) and, to let the checker tell a plateau from a bug, the number
of lags within 0.5 of the maximum together with the smallest / largest such offset.

Runs only in the build container (needs /root/reference).  The JSON is committed; tests/test_gpu_runs.py
(test_seven_ratios_without_a_window) and bench.py's `windowless` leg compare the device with it.

    python tests/golden/make_windowless_golden.py [n_pairs=64] [procs=8]
"""
import json
import multiprocessing as mp
import os
import sys

sys.dont_write_bytecode = True  # importing the reference must not write __pycache__ into /root/reference
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_headline_golden import _reference, fnum  # noqa: E402  (the stub-package import of the reference)


def _solve(seed):
    from workloads import synth

    FFTAligner, MaxScoreAligner = _reference()
    spec = synth.make_pair_spec(seed)
    ref, cands = synth.pair_float_arrays(spec)
    t0 = time.perf_counter()
    msa = MaxScoreAligner(FFTAligner())
    assert msa.max_offset_samples is None and msa.base_aligner.max_offset_samples is None
    (score, offset), winner = msa.fit_transform(ref, list(cands))
    dt7 = time.perf_counter() - t0
    idx = next(i for i, c in enumerate(cands) if c is winner)
    per = [[fnum(s), int(o)] for (s, o), _ in msa._scores]
    gaps, plateaus = [], []
    for c in cands:
        # the reference's own arithmetic (aligners.py:55-74), kept to look at the runner-up lags
        r_, s_ = 2 * np.asarray(ref, dtype=float) - 1, 2 * np.asarray(c, dtype=float) - 1
        n_ = int(2 ** np.ceil(np.log2(len(r_) + len(s_))))
        conv = np.real(np.fft.ifft(np.fft.fft(np.append(np.zeros(n_ - len(s_)), s_)) *
                                   np.fft.fft(np.flip(np.append(r_, np.zeros(n_ - len(r_))), 0))))
        top = np.partition(conv, -2)[-2:]
        gaps.append(float(top[1] - top[0]))
        near = np.flatnonzero(conv >= top[1] - 0.5)
        offs = n_ - 1 - near - len(s_)  # aligners.py:47
        plateaus.append([int(near.size), int(offs.min()), int(offs.max())])
    return {
        "seed": seed,
        "index": idx, "offset": int(offset), "score": fnum(score),
        "per_candidate": per,
        "per_candidate_top2_gap": [round(g, 6) for g in gaps],
        "per_candidate_plateau": plateaus,
        "true_ratio_index": spec.true_ratio_index,
        "true_offset_samples": spec.true_offset_samples,
        "seconds_seven_ratio": dt7,
    }


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(procs) as pool:
        res = pool.map(_solve, range(n_pairs), chunksize=1)
    wall = time.perf_counter() - t0
    assert [r["seed"] for r in res] == list(range(n_pairs))
    out = {
        "_generator": "tests/golden/make_windowless_golden.py",
        "_reference": "smacke/ffsubsync @ /root/reference (v0.5.0), unmodified aligners.py",
        "_numpy": np.__version__,
        "_workload": "workloads.synth.make_pair_spec(seed), duration 7200 s, MaxScoreAligner(FFTAligner()) -- no lag window",
        "_mean_seconds_seven_ratio": float(np.mean([r["seconds_seven_ratio"] for r in res])),
        "pairs": [{k: v for k, v in r.items() if not k.startswith("seconds")} for r in res],
    }
    with open(os.path.join(HERE, "windowless_golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"), sort_keys=True)
    print("wrote %d pairs in %.0f s (%.2f s per seven-ratio solve inside the workers)"
          % (n_pairs, wall, out["_mean_seconds_seven_ratio"]))


if __name__ == "__main__":
    main()
