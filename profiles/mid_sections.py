#!/usr/bin/env python
"""Section experiment of the block-segmented mid pass (results are WRONG in the debug modes: timing only).

    FFS_MID_DEBUG=0  the real kernel
    FFS_MID_DEBUG=1  no row transforms: loads, multiply-accumulate, stores only (memory system alone)
    FFS_MID_DEBUG=2  every pair works on pair 0's buffers: all traffic is L2 hits (compute + LDS + issue alone)
    FFS_MID_DEBUG=3  both (launch + instruction overhead floor)

The switches exist in the LAB build only (`make -C ffsubsync_amd/csrc lab` -> libffsalign_lab.so; the product library
ignores them).  Run once per mode:
    FFS_LIBRARY_PATH=$PWD/ffsubsync_amd/libffsalign_lab.so FFS_MID_DEBUG=k python profiles/mid_sections.py
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

P = 1024
specs = [synth.make_pair_spec(i) for i in range(P)]
db = synth.build_device_batch(specs)
al = batch.BatchAligner(db.required_fft_length(6000), 7, 6000, pairs_in_flight=512)
al.solve_async(db)
torch.cuda.synchronize()
al.plan.profile(True)
for _ in range(5):
    al.solve_async(db)
torch.cuda.synchronize()
kt = al.plan.profile_read()
print(json.dumps({"FFS_MID_DEBUG": os.environ.get("FFS_MID_DEBUG", "0"),
                  "us_per_pair": {k: 1e3 * ms / (5 * P) for k, (ms, n) in kt.items() if n}}))
