"""Per-step cost of the batched golden-section search (GPU box): python profiles/gss_profile.py [files]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ffsubsync_amd import _native, batch
from ffsubsync_amd.batch_gss import fit_gss_batch
from ffsubsync_amd.subtitle_raster import DeviceRaster
from workloads import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
specs = [synth.make_pair_spec(s) for s in range(n)]
refs_t = batch.TrackSet([(sp.ref_starts * 10000, sp.ref_ends * 10000, None) for sp in specs])
data, offs, lens = refs_t.rasterize(np.arange(n), np.ones(n))
refs = [DeviceRaster(data[int(o): int(o) + (int(l) + 31) // 32 * 4].view(torch.int32), 0.0, 1.0, int(l)) for o, l in zip(offs, lens)]
i1 = [i for i, r in enumerate(specs[0].ratios) if r == 1.0][0]
recs = [(sp.cand_starts[i1] * 10000, sp.cand_ends[i1] * 10000, None) for sp in specs]
fit_gss_batch(refs[:8], recs[:8], max_offset_samples=6000)
fit_gss_batch(refs, recs, max_offset_samples=6000)
torch.cuda.synchronize()
out = {}
for label, st in (("plain", {}), ("timed_steps", {"time_steps": True})):
    t0 = time.perf_counter()
    fit_gss_batch(refs, recs, max_offset_samples=6000, stats=st)
    el = time.perf_counter() - t0
    out[label] = {"files_per_s": n / el, "total_us": 1e6 * el, "us_per_step": 1e6 * el / st["steps"], "stats": {k: v for k, v in st.items() if k != "time_steps"}}
print(json.dumps(out))
