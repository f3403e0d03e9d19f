#!/usr/bin/env python
"""k_vad_energy alone: one 90-min 48 kHz s16le file resident in HBM, HIP-event timing, GB/s -- fp32 labels and the
bit-packed form (ffs_vad_energy_bits)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ffsubsync_amd import _native  # noqa: E402

n = 259200000
g = torch.Generator(device="cuda")
g.manual_seed(1)
pcm = (torch.randn(n, device="cuda", generator=g) * 3000).to(torch.int16)
out = {}


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


words = torch.zeros((n // 480 + 31) // 32, dtype=torch.int32, device="cuda")
for name, fn in (("labels_f32", lambda: _native.vad_energy(pcm, 480, 50.0, 0.0)),
                 ("bits", lambda: _native.vad_energy_bits(pcm, 480, 50.0, out=words))):
    ms = min(timed(fn) for _ in range(3))
    out[name] = {"ms_per_file": ms, "GBps": 2 * n / ms / 1e6, "frac_of_8TBps": 2 * n / ms / 1e6 / 8000}
lab = _native.vad_energy(pcm, 480, 50.0, 0.0)
out["speech_frames"] = int((lab > 0).sum())
out["bits_equal_labels"] = bool(torch.equal(_native.unpack_bits(words, lab.numel()), (lab > 0).to(torch.uint8)))
print(json.dumps(out))
