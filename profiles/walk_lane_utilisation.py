"""Host-only: how full are the ds_add_u32 instructions of k_runs_corr's walk on the benchmark pairs?
    python profiles/walk_lane_utilisation.py [pairs]
Replays the kernel's index arithmetic (csrc/ffs_runs.h, step 2) in numpy: one candidate boundary per lane, 64 consecutive
ones per wave task, a task issues two atomics per step for as many steps as its slowest lane needs.  Prints the useful
lane-adds, the atomic wave-instructions issued, the same with the lanes sorted by trip count, and the flat ideal
(coincidences / 64)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import runs_model as rm
from workloads import synth

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W = 12001  # lags of the +-60 s window
useful = issued = issued_sorted = flat = 0
trips = []
for seed in range(n_pairs):
    sp = synth.make_pair_spec(seed)
    ref, cands = synth.pair_arrays(sp)
    q, _ = rm.boundaries(ref)
    ends = np.concatenate([q, [2 ** 40] * 3])[1::2]
    for c in cands:
        p, _ = rm.boundaries(c)
        x, wlim = p - 6000, W - 2
        lb = np.searchsorted(q, x, side="left")
        # steps of a lane: runs (entry pairs) from the one that contains or follows x to the first whose end lies beyond the window
        it = np.maximum(np.searchsorted(ends, x + wlim, side="right") - (lb & ~1) // 2 + 1, 1)
        trips.append(it)
        useful += int((np.searchsorted(q, x + wlim, side="right") - lb).sum())
        for arr, acc in ((it, "issued"), (np.sort(it), "issued_sorted")):
            n = sum(int(arr[t0:t0 + 64].max()) * 2 for t0 in range(0, arr.size, 64))
            if acc == "issued":
                issued += n
            else:
                issued_sorted += n
        flat += -(-int((np.searchsorted(q, x + wlim, side="right") - lb).sum()) // 64)
t = np.concatenate(trips)
print(json.dumps({"pairs": n_pairs, "useful_lane_adds_per_pair": useful / n_pairs,
                  "atomic_wave_instructions_per_pair": issued / n_pairs, "lanes_per_atomic": useful / issued,
                  "lane_utilisation": useful / issued / 64,
                  "sorted_by_trip_count": {"atomic_wave_instructions_per_pair": issued_sorted / n_pairs,
                                           "lane_utilisation": useful / issued_sorted / 64},
                  "flat_ideal_wave_instructions_per_pair": flat / n_pairs,
                  "steps_per_lane": {"mean": float(t.mean()), "std": float(t.std()), "max": int(t.max())}}))
