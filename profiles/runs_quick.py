"""Quick timing of the run-boundary path: python profiles/runs_quick.py [pairs] [algorithm] [max_off|none] (GPU box)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ffsubsync_amd import _native, batch
from workloads import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
algo = sys.argv[2] if len(sys.argv) > 2 else "auto"
mo = None if len(sys.argv) > 3 and sys.argv[3] == "none" else (int(sys.argv[3]) if len(sys.argv) > 3 else 6000)
pif = int(sys.argv[4]) if len(sys.argv) > 4 else 512
specs = [synth.make_pair_spec(s) for s in range(n)]
db = synth.build_device_batch(specs)
n_fft = db.required_fft_length(mo)
streams = int(sys.argv[5]) if len(sys.argv) > 5 else 1
al = batch.BatchAligner(n_fft, 7, mo, pairs_in_flight=pif, algorithm=algo, streams=streams)
cand_out = torch.empty(n * 7 * 24, dtype=torch.uint8, device="cuda")
pair_out = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
for _ in range(2):
    al.solve_async(db, 0, n, cand_out, pair_out)
torch.cuda.synchronize()
[p_.profile(streams == 1) for p_ in al.plans]
reps = 5
t0 = time.perf_counter()
host = 0.0
for _ in range(reps):
    h0 = time.perf_counter()
    al.solve_async(db, 0, n, cand_out, pair_out)
    host += time.perf_counter() - h0
torch.cuda.synchronize()
el = time.perf_counter() - t0
prof = al.plan.profile_read()
pres = pair_out.cpu().numpy().view(_native.PAIR_RESULT_DTYPE)[:n]
ok = sum(int(pres[i]["best_cand"]) == sp.true_ratio_index and abs(int(pres[i]["offset"]) - sp.true_offset_samples) <= 30
         for i, sp in enumerate(specs))
print(json.dumps({"pairs": n, "algorithm": algo, "max_offset": mo, "n_fft": n_fft, "solves_per_s": n * reps / el,
                  "us_per_pair": 1e6 * el / (n * reps), "host_us_per_pair": 1e6 * host / (n * reps), "ground_truth": "%d/%d" % (ok, n),
                  "stats": al.plan.runs_stats(),
                  "kernels_us_per_pair": {k: 1e3 * v[0] / (n * reps) for k, v in prof.items() if v[1]},
                  "launches": {k: v[1] for k, v in prof.items() if v[1]}}))
