#!/usr/bin/env python
"""k_vad_energy alone: one 90-min 48 kHz s16le file resident in HBM, HIP-event timing, GB/s."""
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
for _ in range(3):
    lab = _native.vad_energy(pcm, 480, 50.0, 0.0)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    lab = _native.vad_energy(pcm, 480, 50.0, 0.0)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print("k_vad_energy: %.4f ms per file, %.0f GB/s, speech frames %d of %d" % (ms, 2 * n / ms / 1e6, int((lab > 0).sum()), lab.numel()))
