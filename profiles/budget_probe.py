"""Where is the break-even between the run-boundary path and the transforms?  (GPU box)
    python profiles/budget_probe.py
Windowless seven-ratio solves (every lag searched) and windowed solves on denser vectors (run_scale 1/8, 1/6: 13 600 / 10 200
boundaries per vector), each with FFS_ALGO_RUNS (never the transforms), FFS_ALGO_FFT and FFS_ALGO_AUTO: solves/s."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ffsubsync_amd import _native, batch
from workloads import synth

out = {}
for label, n, mo, scale in (("windowless", 1024, None, 1.0), ("windowed_run_scale_1_6", 512, 6000, 1 / 6.0),
                            ("windowed_run_scale_1_8", 512, 6000, 1 / 8.0), ("windowed_run_scale_1_12", 512, 6000, 1 / 12.0)):
    specs = [synth.make_pair_spec(s, run_scale=scale) for s in range(n)]
    db = synth.build_device_batch(specs)
    n_fft = db.required_fft_length(mo)
    row = {}
    ref = None
    for algo in ("runs", "fft", "auto"):
        al = batch.BatchAligner(n_fft, 7, mo, pairs_in_flight=512, algorithm=algo)
        co = torch.empty(n * 7 * 24, dtype=torch.uint8, device="cuda")
        po = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
        al.solve_async(db, 0, n, co, po)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            al.solve_async(db, 0, n, co, po)
        torch.cuda.synchronize()
        row[algo] = round(n * reps / (time.perf_counter() - t0))
        pres = po.cpu().numpy().view(_native.PAIR_RESULT_DTYPE)[:n]
        key = (pres["best_cand"].copy(), pres["offset"].copy())
        if ref is None:
            ref = key
        row[algo + "_same_answers"] = bool(np.array_equal(key[0], ref[0]) and np.array_equal(key[1], ref[1]))
        row[algo + "_stats"] = al.plan.runs_stats()
        al.close()
    row["boundaries_per_reference"] = int(np.mean([2 * len(sp.ref_starts) for sp in specs[:32]]))
    out[label] = row
    print(label, json.dumps(row), flush=True)
print(json.dumps(out))
