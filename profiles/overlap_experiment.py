"""Concurrent sub-batches on several HIP streams (one plan per stream, the batch split between them): solves/s.

    python profiles/overlap_experiment.py [pairs=1024] [STREAMSxPAIRS_IN_FLIGHT ...]
"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from ffsubsync_amd import batch, _native
from workloads import synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
specs = [synth.make_pair_spec(i) for i in range(P)]
db = synth.build_device_batch(specs)
n = db.required_fft_length(6000)
def run(nstreams, pif, steps=4):
    als = [batch.BatchAligner(n, 7, 6000, pairs_in_flight=pif) for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    outs = [(torch.empty(P*7*24, dtype=torch.uint8, device='cuda'), torch.empty(P*24, dtype=torch.uint8, device='cuda')) for _ in range(nstreams)]
    per = P // nstreams
    def step():
        for k in range(nstreams):
            with torch.cuda.stream(streams[k]):
                als[k].solve_async(db, k*per, (k+1)*per, outs[k][0], outs[k][1])
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for a in als: a.plan.close()
    return P*steps/dt
CASES = [(1,64),(2,32),(2,64),(4,16),(4,32)] if len(sys.argv) <= 2 else [tuple(int(x) for x in a.split('x')) for a in sys.argv[2:]]
for ns, pif in CASES:  # streams x pairs-in-flight, e.g. 1x512 2x256 2x512
    print(ns, pif, round(run(ns,pif)), flush=True)
