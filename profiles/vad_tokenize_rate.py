#!/usr/bin/env python
"""Token smoothing of one 90-minute file (540 000 frames): the workgroup-per-chunk kernel (100 s chunks, the reference's
buffer) against the one-thread-per-chunk state machine, which chunks above 28 672 frames still take (here: 30 000-frame
chunks -- the round-2 environment switch that forced it is gone).

    python profiles/vad_tokenize_rate.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ffsubsync_amd import _native  # noqa: E402

rng = np.random.RandomState(5)
n = 540000
runs = rng.geometric(1.0 / 60, size=n // 30)
valid = np.repeat(rng.rand(runs.size) < 0.5, runs)[:n].astype(np.float32)
dev = torch.from_numpy(valid).cuda()
out = {}
ref = None
for label, chunk in (("scan_100s_chunks", 10000), ("serial_300s_chunks", 30000)):
    res = {}
    for what, frames in (("one_chunk", chunk), ("whole_file", n)):
        x = dev[:frames]
        got = _native.vad_tokenize(x, chunk, 20, 500, 25, 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            got = _native.vad_tokenize(x, chunk, 20, 500, 25, 0.0)
        torch.cuda.synchronize()
        res[what + "_ms"] = 1e3 * (time.perf_counter() - t0) / 20
    out[label] = res
print(json.dumps(out))
