#!/usr/bin/env python
"""Measured HBM ceilings on this box with plain torch kernels: write-only (fill), read-only (sum),
copy (read + write).  Context for the pass-A roofline: pass A is a write-only kernel."""
import json

import torch

n = 1 << 30  # bytes
x = torch.empty(n // 4, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


out = {}
t = timed(lambda: x.fill_(1.0))
out["fill_GBps"] = n / t / 1e9
t = timed(lambda: x.zero_())
out["memset_GBps"] = n / t / 1e9
t = timed(lambda: y.copy_(x))
out["copy_read_plus_write_GBps"] = 2 * n / t / 1e9
t = timed(lambda: x.sum())
out["sum_read_GBps"] = n / t / 1e9
print(json.dumps(out))
