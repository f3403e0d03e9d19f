"""Where does a SMALL step go?  python profiles/small_step_profile.py [pairs=128] [steps=300]  (GPU box)
One rank's share of configs[3] at 8 GPUs (128 pairs x 7 ratios from bits): wall time per step (no profiling), the library's
own host-side sections (FFS_HOST_TIMING=1: printed when the plan is destroyed), then per-kernel HIP-event times."""
import json, os, sys, time
os.environ["FFS_HOST_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ffsubsync_amd import _native, batch
from workloads import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
specs = [synth.make_pair_spec(s) for s in range(n)]
db = synth.build_device_batch(specs)
n_fft = db.required_fft_length(6000)
out = {}
for label, dbx in (("bits", db), ("lists", None)):
    if dbx is None:
        dbx = db.to_runs(cap=8192)
        torch.cuda.synchronize()
        n_l = dbx.data.view(torch.int32).reshape(-1, int(dbx.offs.ravel()[1]) // 4)[:, 0].cpu().numpy().reshape(dbx.offs.shape)
        dbx.bounds = (n_l + 2).astype(np.int32)
    al = batch.BatchAligner(n_fft, 7, 6000, pairs_in_flight=n)
    cand_out = torch.empty(n * 7 * 24, dtype=torch.uint8, device="cuda")
    pair_out = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
    for _ in range(20):
        al.solve_async(dbx, 0, n, cand_out, pair_out)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        al.solve_async(dbx, 0, n, cand_out, pair_out)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    al.plan.profile(True)
    for _ in range(50):
        al.solve_async(dbx, 0, n, cand_out, pair_out)
    torch.cuda.synchronize()
    prof = al.plan.profile_read()
    out[label] = {"us_per_step_wall": 1e6 * wall / steps, "us_per_step_host_in_call": 1e6 * host / steps,
                  "kernels_us_per_step": {k: 1e3 * v[0] / 50 for k, v in prof.items() if v[1]}}
    al.close()   # (prints the host-timing sections to stderr)
print(json.dumps(out))
