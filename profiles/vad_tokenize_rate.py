#!/usr/bin/env python
"""Token smoothing of one 90-minute file (540 000 frames, 100 s chunks): the scan kernel against the one-thread-per-
chunk state machine (FFS_VAD_TOKENIZE_SERIAL=1).

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
for label, env in (("scan", None), ("serial", "1")):
    if env:
        os.environ["FFS_VAD_TOKENIZE_SERIAL"] = env
    res = {}
    for what, frames in (("one_100s_buffer", 10000), ("whole_file_54_chunks", n)):
        x = dev[:frames]
        got = _native.vad_tokenize(x, 10000, 20, 500, 25, 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            got = _native.vad_tokenize(x, 10000, 20, 500, 25, 0.0)
        torch.cuda.synchronize()
        res[what + "_ms"] = 1e3 * (time.perf_counter() - t0) / 20
        if frames == n:
            if ref is None:
                ref = got.clone()
            res["equal_to_scan"] = bool(torch.equal(ref, got))
    out[label] = res
    os.environ.pop("FFS_VAD_TOKENIZE_SERIAL", None)
print(json.dumps(out))
