"""TEST INFRASTRUCTURE ONLY -- multi-process timing of the CPU restatement (CPU baseline (ii)).

Run as a separate process by bench.py's `cpu_baseline` leg (never imported by the product path):

    python -m oracle.cpu_parallel_baseline <procs> <pairs_per_proc> [duration_s]

forks `procs` workers (no GPU runtime is initialised in this process), each solving its own seeded
2 h x 7-ratio problems with oracle.aligners_oracle (numpy complex128, single-threaded pocketfft --
aligners.py:50-167), and prints one JSON line with the aggregate solves/s.
"""
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _work(args):
    seed0, n, duration = args
    from workloads import synth
    from oracle import aligners_oracle as orc

    problems = []
    for seed in range(seed0, seed0 + n):  # input generation is not timed
        spec = synth.make_pair_spec(seed, duration_s=duration)
        problems.append((spec,) + tuple(synth.pair_float_arrays(spec)))
    ok = 0
    t0 = time.perf_counter()
    for spec, ref, cands in problems:
        (score, offset), idx = orc.max_score_align(ref, cands, 6000)
        ok += int(idx == spec.true_ratio_index)
    return time.perf_counter() - t0, ok


def main():
    procs, per = int(sys.argv[1]), int(sys.argv[2])
    duration = float(sys.argv[3]) if len(sys.argv) > 3 else 7200.0
    ctx = mp.get_context("fork")
    with ctx.Pool(procs) as pool:
        pool.map(_work, [(10_000 + i, 0, duration) for i in range(procs)])  # start the workers, import numpy
        res = pool.map(_work, [(20_000 + i * per, per, duration) for i in range(procs)], chunksize=1)
    slowest = max(t for t, _ in res)  # the workers solve concurrently; the batch is done when the slowest is
    print(json.dumps({
        "value": procs * per / slowest,
        "cores": procs,
        "solves": procs * per,
        "slowest_worker_s": slowest,
        "mean_solve_s": sum(t for t, _ in res) / (procs * per),
        "recovered": int(sum(ok for _, ok in res)),
    }))


if __name__ == "__main__":
    main()
