#!/usr/bin/env python
"""Windowless solves (max_offset_samples=None): time per pair at the two admissible plan lengths for 2 h
pairs -- 3*2^19 (radix-3 columns, N1 = 384) and the reference's 2^21 (N1 = 512) -- single-ratio and
seven-ratio, with the per-kernel HIP-event split.  Decides which length ffs_plan_length should prefer.

    python profiles/none_path_lengths.py [pairs=1024]
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

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
specs = [synth.make_pair_spec(i) for i in range(P)]
db = synth.build_device_batch(specs)
sdb = db.select_candidates([sp.true_ratio_index for sp in specs])
out = {}
for label, the_db, cands in (("single_ratio", sdb, 1), ("seven_ratio", db, 7)):
    for n in (3 << 19, 1 << 21):
        al = batch.BatchAligner(n, cands, None, pairs_in_flight=512)
        al.solve_async(the_db)
        torch.cuda.synchronize()
        al.plan.profile(True)
        t0 = time.perf_counter()
        for _ in range(3):
            c, p = al.solve_async(the_db)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        kt = al.plan.profile_read()
        out["%s_n%d" % (label, n)] = {"us_per_pair": 1e6 * dt / P, "solves_per_s": P / dt,
                                      "kernels_us_per_pair": {k: 1e3 * ms / (3 * P) for k, (ms, cnt) in kt.items() if cnt}}
        al.plan.close()
print(json.dumps(out, indent=1))
