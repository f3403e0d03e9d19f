"""Worker of tests/test_gpu_round3.py::test_two_ranks_on_one_gpu_*: launched under torch.distributed.run with two
ranks that SHARE cuda:0 (one-GPU box).  Each rank solves its shard_bounds slice of the bench pairs with BatchAligner
(the HIP path), the 24-byte records are all-gathered through a gloo group (RCCL refuses two ranks on one device), and
rank 0 writes all records to the file given on the command line."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffsubsync_amd import _native, batch  # noqa: E402
from workloads import synth  # noqa: E402


def main():
    out_path, n_pairs = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = batch.shard_bounds(n_pairs, rank, world)
    per = (n_pairs + world - 1) // world
    specs = [synth.make_pair_spec(s) for s in range(lo, hi)]
    local = torch.zeros(per * 24, dtype=torch.uint8)
    if specs:
        db = synth.build_device_batch(specs, packed=True)
        al = batch.BatchAligner(db.required_fft_length(6000), 7, 6000, pairs_in_flight=4)
        _, pair_out = al.solve_async(db)
        torch.cuda.synchronize()
        local[: (hi - lo) * 24] = pair_out[: (hi - lo) * 24].cpu()
        al.close()
    allr = batch.gather_pair_results(local, n_pairs, world)
    every = [None] * world
    dist.all_gather_object(every, (rank, torch.cuda.current_device(), hi - lo))
    if rank == 0:
        np.save(out_path, allr.numpy())
        print("RANKS", every, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
