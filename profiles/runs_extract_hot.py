"""Timing-only experiment: k_runs_extract when every vector of the call is the SAME memory (all reads become L2 hits) vs
the real batch -- tells memory-system time from in-kernel time.  Results of the hot run are meaningless."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ffsubsync_amd import _native, batch
from workloads import synth

n = 2048
specs = [synth.make_pair_spec(s) for s in range(n)]
db = synth.build_device_batch(specs)
out = {}
for label in ("real", "hot"):
    if label == "hot":
        db.offs[:] = db.offs[0:1]      # every pair reads pair 0's vectors
        db.lens[:] = db.lens[0:1]
    al = batch.BatchAligner(db.required_fft_length(6000), 7, 6000, pairs_in_flight=512, algorithm="auto")
    for _ in range(2):
        al.solve_async(db, 0, n)
    torch.cuda.synchronize()
    al.plan.profile(True)
    for _ in range(5):
        al.solve_async(db, 0, n)
    torch.cuda.synchronize()
    prof = al.plan.profile_read()
    out[label] = {k: 1e3 * v[0] / (n * 5) for k, v in prof.items() if v[1]}
    al.close()
print(json.dumps(out))
