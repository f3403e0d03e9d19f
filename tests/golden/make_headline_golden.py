"""Generate tests/golden/headline_golden.json: the UNMODIFIED reference aligner
(/root/reference/ffsubsync/aligners.py) run on the benchmark's own inputs -- BASELINE configs[2]
(seeds 0..N-1 of workloads.synth.make_pair_spec: 2 h @ 100 Hz, seven framerate-ratio candidates,
MaxScoreAligner(FFTAligner, None, 100, 60)) and configs[1] (the true-ratio candidate alone, with
FFTAligner(None) and FFTAligner(6000)).

Runs only in the build container (needs /root/reference).  The JSON it writes is committed; the
`-m gpu` test tests/test_gpu_headline.py and bench.py compare the timed batch with it.  The same
run is timed and written to profiles/r04_cpu_reference_baseline.json (the unmodified
reference, one process and a pool, on this container's cores).

    python tests/golden/make_headline_golden.py [n_pairs=1024] [procs=8]
"""
import json
import logging
import multiprocessing as mp
import os
import sys

sys.dont_write_bytecode = True  # importing the reference must not write __pycache__ into /root/reference
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _reference():
    """Import the reference modules without executing ffsubsync/__init__.py (needs ffmpeg/srt/pysubs2)."""
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


def _solve(seed):
    """One bench problem through the reference: seven-ratio solve + the two single-ratio solves."""
    from workloads import synth

    FFTAligner, MaxScoreAligner = _reference()
    spec = synth.make_pair_spec(seed)
    ref, cands = synth.pair_float_arrays(spec)
    t0 = time.perf_counter()
    msa = MaxScoreAligner(FFTAligner, None, 100, 60)
    (score, offset), winner = msa.fit_transform(ref, list(cands))
    dt7 = time.perf_counter() - t0
    idx = next(i for i, c in enumerate(cands) if c is winner)
    per = [[fnum(s), int(o)] for (s, o), _ in msa._scores]
    # Top-2 gap of every candidate's masked `convolve` (an offset is only defined when the gap
    # exceeds 0.5 -- among exactly tied lags the reference's pick is its own fp64 FFT rounding noise, which
    # happens on the plateaus of wrong-ratio candidates).  Same arithmetic as aligners.py:55-78.
    gaps = []
    for c in cands:
        al = FFTAligner(6000)
        r_, s_ = 2 * np.asarray(ref, dtype=float) - 1, 2 * np.asarray(c, dtype=float) - 1
        n_ = int(2 ** np.ceil(np.log2(len(r_) + len(s_))))
        conv = np.real(np.fft.ifft(np.fft.fft(np.append(np.zeros(n_ - len(s_)), s_)) *
                                   np.fft.fft(np.flip(np.append(r_, np.zeros(n_ - len(r_))), 0))))
        m = al._eliminate_extreme_offsets_from_solutions(conv, s_)
        top = np.partition(m[np.isfinite(m)], -2)[-2:]
        gaps.append(float(top[1] - top[0]))
    tc = cands[spec.true_ratio_index]
    t0 = time.perf_counter()
    s_none, o_none = FFTAligner(None).fit_transform(ref, tc, get_score=True)
    dt1 = time.perf_counter() - t0
    s_6000, o_6000 = FFTAligner(6000).fit_transform(ref, tc, get_score=True)
    return {
        "seed": seed,
        "index": idx, "offset": int(offset), "score": fnum(score),
        "per_candidate": per,
        "per_candidate_top2_gap": [round(g, 6) for g in gaps],
        "single_none": [fnum(s_none), int(o_none)],
        "single_6000": [fnum(s_6000), int(o_6000)],
        "true_ratio_index": spec.true_ratio_index,
        "true_offset_samples": spec.true_offset_samples,
        "seconds_seven_ratio": dt7, "seconds_single": dt1,
    }


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    # (i) one process, 16 pairs, best of 3
    one = []
    for rep in range(3):
        t0 = time.perf_counter()
        first = [_solve(s) for s in range(16)]
        one.append(sum(r["seconds_seven_ratio"] for r in first))
        print("1 process, 16 pairs, run %d: %.1f s of solves" % (rep, one[-1]), flush=True)
    # (ii) a pool over all pairs (also produces the goldens)
    ctx = mp.get_context("fork")
    with ctx.Pool(procs) as pool:
        pool.map(_solve, range(procs))  # start workers, import numpy
        pool_t = []
        for rep in range(3 if n_pairs <= 32 else 1):
            t0 = time.perf_counter()
            res = pool.map(_solve, range(n_pairs), chunksize=1)
            pool_t.append(time.perf_counter() - t0)
    assert [r["seed"] for r in res] == list(range(n_pairs))
    for a, b in zip(first, res):
        assert {k: v for k, v in a.items() if not k.startswith("seconds")} == \
               {k: v for k, v in b.items() if not k.startswith("seconds")}, "reference is not deterministic?"
    out = {
        "_generator": "tests/golden/make_headline_golden.py",
        "_reference": "smacke/ffsubsync @ /root/reference (v0.5.0), unmodified aligners.py",
        "_numpy": np.__version__,
        "_workload": "workloads.synth.make_pair_spec(seed), duration 7200 s, MaxScoreAligner(FFTAligner, None, 100, 60)",
        "pairs": [{k: v for k, v in r.items() if not k.startswith("seconds")} for r in res],
    }
    with open(os.path.join(HERE, "headline_golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"), sort_keys=True)
    base = {
        "what": "UNMODIFIED reference MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform on bench seeds "
                "(2 h @ 100 Hz x 7 ratios), numpy %s pocketfft, input generation excluded" % np.__version__,
        "host": "build container, %d vCPUs (%s)" % (os.cpu_count() or 0, _cpu_model()),
        "one_process": {"pairs": 16, "runs_s": one, "best_solves_per_s": 16 / min(one), "cores": 1},
        # every worker is busy for the whole map, so the pool's rate is procs / (mean seconds per seven-ratio solve
        # measured inside the busy workers); the map's wall-clock also covers the extra single-ratio / gap solves
        "pool": {"processes": procs, "pairs": n_pairs, "map_wall_s": pool_t,
                 "mean_seven_ratio_solve_s": float(np.mean([r["seconds_seven_ratio"] for r in res])),
                 "best_solves_per_s": procs / float(np.mean([r["seconds_seven_ratio"] for r in res])), "cores": procs},
        "single_ratio_none_mean_s": float(np.mean([r["seconds_single"] for r in res])),
    }
    with open(os.path.join(ROOT, "profiles", "r04_cpu_reference_baseline.json"), "w") as f:
        json.dump(base, f, indent=1)
    print(json.dumps(base))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
