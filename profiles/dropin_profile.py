"""cProfile of the drop-in solve, one 2 h x 7-ratio problem per call: python profiles/dropin_profile.py [host|device]
(host: float64 arrays as an unpatched pipeline hands them over; device: DeviceRaster inputs with their boundary lists)."""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np, torch
from workloads import synth
from ffsubsync_amd.aligners import FFTAligner, MaxScoreAligner
from ffsubsync_amd.subtitle_raster import DeviceRaster
spec = synth.make_pair_spec(0)
ref, cands = synth.pair_float_arrays(spec)
if len(sys.argv) > 1 and sys.argv[1] == "device":
    ref, cands = DeviceRaster.from_host(ref), [DeviceRaster.from_host(c) for c in cands]
for _ in range(3):
    MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform(ref, list(cands))
def run(n=200):
    for _ in range(n):
        MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform(ref, list(cands))
t0=time.perf_counter(); run(); print("ms per solve", (time.perf_counter()-t0)/200*1e3)
cProfile.run("run(100)", "/tmp/d.prof")
pstats.Stats("/tmp/d.prof").sort_stats("cumtime").print_stats(28)
