import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from workloads import synth
from ffsubsync_amd.aligners import FFTAligner, MaxScoreAligner
from ffsubsync_amd.subtitle_raster import DeviceRaster
spec = synth.make_pair_spec(0)
ref, cands = synth.pair_float_arrays(spec)
for _ in range(2):
    MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform(ref, list(cands))
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    (s, o), w = MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform(ref, list(cands))
    ts.append(time.perf_counter() - t0)
print("host float64 arrays -> result: median %.2f ms (min %.2f)" % (1e3 * np.median(ts), 1e3 * min(ts)), s, o)
# HBM-resident inputs
ref01, c01 = synth.pair_arrays(spec)
from ffsubsync_amd import _native
pk = lambda x: _native.pack_bits(torch.from_numpy(x).cuda())
dref = DeviceRaster(pk(ref01), 0.0, 1.0, ref01.size)  # bit-packed, as the rasteriser / VAD leave them in HBM
dc = [DeviceRaster(pk(c), 0.0, a, c.size) for c, a in zip(c01, spec.cand_amp)]
for _ in range(2):
    MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform(dref, list(dc))
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    (s, o), w = MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform(dref, list(dc))
    ts.append(time.perf_counter() - t0)
print("HBM-resident bit-packed vectors -> result: median %.2f ms (min %.2f)" % (1e3 * np.median(ts), 1e3 * min(ts)), s, o)
