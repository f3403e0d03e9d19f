"""N > 1 path on CPU: two gloo ranks shard the pairs by rank (no data exchange during solves) and
all-gather the 24-byte per-pair records, exactly as bench.py / batch.gather_pair_results do over
RCCL.  The per-rank 'solver' here is the CPU oracle (test infrastructure) so the test runs
without a GPU; what is under test is the sharding + gather logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ffsubsync_amd import _native
from workloads import synth
from ffsubsync_amd.batch import gather_pair_results, shard_bounds
from oracle import aligners_oracle as orc

N_PAIRS = 5  # deliberately not divisible by the world size
WORLD = 2


def _solve_local(seeds):
    out = np.zeros(len(seeds), dtype=_native.PAIR_RESULT_DTYPE)
    for i, seed in enumerate(seeds):
        spec = synth.make_pair_spec(seed, duration_s=120.0)
        ref, cands = synth.pair_float_arrays(spec)
        (score, offset), idx = orc.max_score_align(ref, cands, 6000)
        out[i] = (score, offset, idx, 0)
    return out


def _worker(rank, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    lo, hi = shard_bounds(N_PAIRS, rank, WORLD)
    per = (N_PAIRS + WORLD - 1) // WORLD
    local = np.zeros(per, dtype=_native.PAIR_RESULT_DTYPE)
    local[: hi - lo] = _solve_local(list(range(lo, hi)))
    buf = torch.from_numpy(local.view(np.uint8).copy())
    allr = gather_pair_results(buf, N_PAIRS, WORLD)
    ret[rank] = allr.numpy().tobytes()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_matches_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(port, ret), nprocs=WORLD, join=True)
    single = _solve_local(list(range(N_PAIRS)))
    for rank in range(WORLD):
        got = np.frombuffer(ret[rank], dtype=_native.PAIR_RESULT_DTYPE)
        assert got.size == N_PAIRS
        assert np.array_equal(got["offset"], single["offset"])
        assert np.array_equal(got["best_cand"], single["best_cand"])
        assert np.array_equal(got["score"], single["score"])
