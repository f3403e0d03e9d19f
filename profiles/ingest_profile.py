"""Where the time of the boundary-list ingest stream goes (GPU box): python profiles/ingest_profile.py [pairs]
Times, for one batch of `pairs` x 8 vectors from interval lists: TrackSet construction, TrackSet.rasterize_runs (host
only / with device sync), BatchAligner.solve_async on the lists (host only / with sync), and the per-step cost of the
batched golden-section search."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ffsubsync_amd import _native, batch
from ffsubsync_amd.constants import candidate_ratios
from workloads import synth

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ratios = candidate_ratios()
recs = []
for f in range(n_pairs):
    rng = np.random.RandomState(9000 + f)
    s_us, e_us, meta = synth.make_subtitle_records(9000 + f, duration_s=120 * 60 * 0.95)
    idx, shift_us = int(rng.randint(7)), int(rng.randint(-40, 40)) * 1_000_000
    r_s = np.maximum(np.rint(s_us * ratios[idx]).astype(np.int64) + shift_us, 0)
    r_e = np.maximum(np.rint(e_us * ratios[idx]).astype(np.int64) + shift_us, 0)
    recs.append(((r_s, r_e, meta), (s_us, e_us, meta)))
tracks = [t for rec in recs for t in rec]
track_of = (np.tile(np.array([0] + [1] * 7), (n_pairs, 1)) + 2 * np.arange(n_pairs)[:, None]).ravel()
ratio = np.tile(np.array([1.0] + list(ratios)), (n_pairs, 1)).ravel()
hi = np.minimum(1.0 / ratio, 1.0).reshape(n_pairs, 8)


def t(fn, reps=10, sync=True):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    if sync:
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    return 1e6 * dt


out = {"pairs": n_pairs, "subtitles_per_track": float(np.mean([len(t_[0]) for t_ in tracks]))}
out["trackset_build_us"] = t(lambda: batch.TrackSet(tracks), sync=False)
ts_host = batch.TrackSet(tracks)
ts_dev = batch.TrackSet(tracks).to_device()
out["rasterize_runs_host_tables_us"] = {"host_only": t(lambda: ts_host.rasterize_runs(track_of, ratio), sync=False),
                                        "with_sync": t(lambda: ts_host.rasterize_runs(track_of, ratio))}
out["rasterize_runs_device_tables_us"] = {"host_only": t(lambda: ts_dev.rasterize_runs(track_of, ratio), sync=False),
                                          "with_sync": t(lambda: ts_dev.rasterize_runs(track_of, ratio))}
ts_pin = batch.TrackSet(tracks).pin()
out["rasterize_runs_pinned_tables_us"] = {"host_only": t(lambda: ts_pin.rasterize_runs(track_of, ratio), sync=False),
                                          "with_sync": t(lambda: ts_pin.rasterize_runs(track_of, ratio))}
out["rasterize_bits_us"] = {"with_sync": t(lambda: ts_host.rasterize(track_of, ratio))}
data, offs, lens, bounds = ts_dev.rasterize_runs(track_of, ratio)
db = batch.DeviceBatch(data, offs.reshape(n_pairs, 8), lens.reshape(n_pairs, 8), np.zeros_like(hi), hi, _native.FFS_DTYPE_RUNS, None,
                       bounds.reshape(n_pairs, 8))
al = batch.BatchAligner(batch.pairs_from_intervals(recs[:4], ratios).required_fft_length(6000), 7, 6000, pairs_in_flight=min(256, n_pairs))
co = torch.empty(n_pairs * 7 * 24, dtype=torch.uint8, device="cuda")
po = torch.empty(n_pairs * 24, dtype=torch.uint8, device="cuda")
out["solve_lists_us"] = {"host_only": t(lambda: al.solve_async(db, 0, n_pairs, co, po), sync=False),
                         "with_sync": t(lambda: al.solve_async(db, 0, n_pairs, co, po))}
db_nb = batch.DeviceBatch(data, db.offs, db.lens, db.lo, db.hi, _native.FFS_DTYPE_RUNS)
out["solve_lists_without_bounds_us"] = {"with_sync": t(lambda: al.solve_async(db_nb, 0, n_pairs, co, po))}
al.plan.profile(True)
for _ in range(5):
    al.solve_async(db, 0, n_pairs, co, po)
torch.cuda.synchronize()
out["solve_lists_kernels_us"] = {k: 1e3 * v[0] / v[1] for k, v in al.plan.profile_read().items() if v[1]}
al.close()
print(json.dumps(out))

# the stream as bench.py runs it (resident tracks), with and without warm-up of the allocator
al = batch.BatchAligner(batch.pairs_from_intervals(recs[:4], ratios).required_fft_length(6000), 7, 6000, pairs_in_flight=min(256, n_pairs))
outs = [(torch.empty(n_pairs * 7 * 24, dtype=torch.uint8, device="cuda"), torch.empty(n_pairs * 24, dtype=torch.uint8, device="cuda")) for _ in range(2)]
keep = []
def run(k):
    data, offs, lens, bounds = ts_dev.rasterize_runs(track_of, ratio)
    db = batch.DeviceBatch(data, offs.reshape(n_pairs, 8), lens.reshape(n_pairs, 8), np.zeros_like(hi), hi, _native.FFS_DTYPE_RUNS, None, bounds.reshape(n_pairs, 8))
    keep.append(db)
    al.solve_async(db, 0, n_pairs, outs[k % 2][0], outs[k % 2][1])
    if len(keep) > 2:
        keep.pop(0)
stream = {}
for label, warm in (("first", 1), ("after_8_warm_batches", 8)):
    for k in range(warm):
        run(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for k in range(16):
        run(k)
        marks.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stream[label] = {"us_per_batch": 1e6 * dt / 16, "pairs_per_s": 16 * n_pairs / dt, "host_marks_us": [round(1e6 * m) for m in marks]}
print(json.dumps({"stream_resident_tracks": stream}))
al.close()
