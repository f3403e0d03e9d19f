#!/usr/bin/env python
"""Same-box A/B of mid-pass variants: per-kernel us/pair (plan profile) + throughput + golden check.

    python profiles/mid_ab.py '{"FFS_DISABLE_HALF_LAST": "1"}' '{}' ...      (one JSON env dict per variant)

A variant is an environment: one of the five run-time knobs, `FFS_LIBRARY_PATH` pointing at another build
(`make variant NAME=x DEFS=-D...`), or -- with the lab build, `make lab` -- the timing-only section switches.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ffsubsync_amd import batch  # noqa: E402
from workloads import synth  # noqa: E402

P = int(os.environ.get("AB_PAIRS", "1024"))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "headline_golden.json")))["pairs"]
specs = [synth.make_pair_spec(i) for i in range(P)]
db = synth.build_device_batch(specs)
for arg in sys.argv[1:]:
    env = json.loads(arg)
    for k, v in env.items():
        os.environ[k] = v
    al = batch.BatchAligner(db.required_fft_length(6000), 7, 6000, pairs_in_flight=512)
    cres, pres = al.solve(db)
    ok = sum(int(pres[i]["best_cand"]) == g["index"] and int(pres[i]["offset"]) == g["offset"] for i, g in enumerate(GOLD[:P]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        al.solve_async(db)
    torch.cuda.synchronize()
    rate = 4 * P / (time.perf_counter() - t0)
    al.plan.profile(True)
    for _ in range(3):
        al.solve_async(db)
    torch.cuda.synchronize()
    kt = al.plan.profile_read()
    al.plan.close()
    for k in env:
        del os.environ[k]
    print(json.dumps({"env": env, "solves_per_s": round(rate, 1), "golden": "%d/%d" % (ok, min(P, len(GOLD))),
                      "us_per_pair": {k: round(1e3 * ms / (3 * P), 3) for k, (ms, n) in kt.items() if n}}), flush=True)
