import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np, torch
from workloads import synth
from ffsubsync_amd.aligners import FFTAligner, MaxScoreAligner
spec = synth.make_pair_spec(0)
ref, cands = synth.pair_float_arrays(spec)
for _ in range(3):
    MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform(ref, list(cands))
def run():
    for _ in range(50):
        MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform(ref, list(cands))
t0=time.perf_counter(); run(); print("ms per solve", (time.perf_counter()-t0)/50*1e3)
cProfile.run("run()", "/tmp/d.prof")
pstats.Stats("/tmp/d.prof").sort_stats("cumtime").print_stats(22)
