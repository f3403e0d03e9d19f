"""GPU tests of the pieces added around the aligner core: bit-packed vectors, the zero-overlap rule,
per-candidate pool quotas, the device scatter of the multi-segment reference, the bulk .npz feeder,
stream ordering on one plan and the one-rank RCCL gather."""
import io
import os
import threading

import numpy as np
import pytest

from oracle import aligners_oracle as orc
from oracle import vad_oracle as vo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def torch():
    import torch as t

    assert t.cuda.is_available()
    return t


def test_pack_bits_matches_numpy_packbits(torch):
    from ffsubsync_amd import _native

    rng = np.random.RandomState(1)
    for n in (1, 31, 32, 33, 63, 4096, 100003, 720000):
        x = (rng.rand(n) < 0.4).astype(np.uint8) * rng.randint(1, 255, n).astype(np.uint8)  # any non-zero byte is a 1
        want = np.packbits(x != 0, bitorder="little")
        want = np.concatenate([want, np.zeros(-want.size % 4, np.uint8)])
        got = _native.pack_bits(torch.from_numpy(x).cuda()).view(torch.uint8).cpu().numpy()
        assert np.array_equal(got, want), n
        # unaligned source, float source with a threshold
        buf = torch.from_numpy(np.concatenate([[7], x]).astype(np.uint8)).cuda()
        assert np.array_equal(_native.pack_bits(buf[1:]).view(torch.uint8).cpu().numpy(), want)
        f = np.where(x != 0, 1.0, rng.choice([0.0, -0.5, 0.25], n)).astype(np.float32)
        assert np.array_equal(_native.pack_bits(torch.from_numpy(f).cuda(), 0.5).view(torch.uint8).cpu().numpy(), want)
        assert np.array_equal(_native.unpack_bits(_native.pack_bits(torch.from_numpy(x).cuda()), n).cpu().numpy(), x != 0)


@pytest.mark.parametrize("name", ["a", "b", "late_start"])
def test_bit_packed_rasters_match_reference(torch, name):
    from ffsubsync_amd import _native

    gold = np.load(os.path.join(HERE, "golden", "raster_golden.npz"))
    s, e, m = gold[name + "_start_us"], gold[name + "_end_us"], gold[name + "_meta"]
    ss = int(gold[name + "_start_seconds"])
    for j, r in enumerate(float(x) for x in gold["ratios"]):
        words, n = _native.rasterize_subtitles(s, e, m, r, 100, ss, packed=True)
        want = gold["%s_r%d" % (name, j)]
        assert n == want.size
        assert np.array_equal(np.unpackbits(words.view(torch.uint8).cpu().numpy(), bitorder="little")[:n], want)


def test_lags_without_overlap_follow_the_zero_rule(torch):
    """Lags with an empty overlap are exactly 0 in the reference's `convolve` (fp64 noise ~1e-10 around 0).
    When every real lag scores below 0 the maximum is one of them; the device reports score 0.0 at the
    largest such lag (np.argmax's first k).  The reference's own pick among those noise-level ties is not
    defined, so only the score is compared with the oracle."""
    from ffsubsync_amd.aligners import FFTAligner

    rng = np.random.RandomState(3)
    for R, S in ((5000, 3000), (3000, 5000), (700, 300)):  # FFT path twice, direct kernel once
        ref = np.ones(R)
        sub = np.zeros(S)  # +1 against -1 everywhere: every overlapping lag is negative
        score, offset = FFTAligner(None).fit_transform(ref, sub, get_score=True)
        o_score, _ = orc.fft_align(ref, sub, None)
        n_ref = orc.fft_length(R, S)
        assert float(score) == 0.0 and abs(float(o_score)) < 1e-6
        assert offset == n_ref - 1 - S and offset >= R  # k = 0: the first maximum
        # a window that keeps real lags only changes nothing about them
        sub2 = (rng.rand(S) < 0.5).astype(float)
        ref2 = (rng.rand(R) < 0.5).astype(float)
        got = FFTAligner(None).fit_transform(ref2, sub2, get_score=True)
        exp = orc.fft_align(ref2, sub2, None)
        assert got[1] == exp[1] and float(got[0]) == pytest.approx(float(exp[0]), rel=1e-9)
    # Python negative-slice window of short inputs (R = S = 3000, max 6000 -> lags [-3000, -2192]): the
    # lag -3000 has no overlap; it only wins when everything else is negative
    ref, sub = np.ones(3000), np.zeros(3000)
    score, offset = FFTAligner(6000).fit_transform(ref, sub, get_score=True)
    assert float(score) == 0.0 and offset == -3000


def test_one_degenerate_pair_does_not_change_its_neighbours(torch):
    """Per-candidate pool quotas: a silent reference (thousands of exactly tied lags in every candidate)
    shares a batch with ordinary pairs; the ordinary pairs' records are those of a batch without it, and the
    degenerate pair still gets the exact answer."""
    from ffsubsync_amd.aligners import _Vec, solve_pairs
    from workloads import synth

    rng = np.random.RandomState(11)
    normal = []
    for i in range(3):
        spec = synth.make_pair_spec(40 + i, duration_s=300.0)
        ref, cands = synth.pair_float_arrays(spec)
        normal.append((_Vec(ref), [_Vec(c) for c in cands]))
    silent_ref = np.zeros(30000)
    subs = [0.4 * (rng.rand(20000 + 100 * j) < 0.3) for j in range(7)]
    silent = (_Vec(silent_ref), [_Vec(s) for s in subs])
    alone_c, alone_p = solve_pairs(normal, 6000, 6000)
    mixed_c, mixed_p = solve_pairs(normal[:1] + [silent] + normal[1:], 6000, 6000)
    keep = [0, 2, 3]
    assert np.array_equal(mixed_p[keep], alone_p)
    for f in ("score", "offset", "flags"):
        assert np.array_equal(mixed_c[keep][f], alone_c[f])
    assert not (mixed_c["flags"] & 2).any()
    for j, s in enumerate(subs):
        exp = orc.fft_align(silent_ref, s, 6000)
        conv, S = orc.convolve_full(silent_ref, s)
        m = orc.mask_extreme_offsets(conv, S, 6000)
        k = int(np.argmax(m >= m.max() - 1e-6))
        assert int(mixed_c[1, j]["offset"]) == len(m) - 1 - k - S
        assert float(mixed_c[1, j]["score"]) == pytest.approx(float(exp[0]), rel=1e-9)


def test_device_scatter_of_the_multi_segment_reference(torch):
    """assemble_sparse_reference == the reference's host loop (speech_transformers.py:871-890), with the
    window labels produced by the GPU detector from four threads at once."""
    from ffsubsync_amd.speech_transformers import assemble_sparse_reference, detect_device

    total_s, win = 600, 30
    pcm, _ = vo.synth_pcm(480 * 100 * total_s, seed=21)
    full = vo.detect_fast(pcm)
    starts = [0, 114, 228, 342, 456, 570, 585]  # the last window is clipped at the end, the last two overlap
    dev_pcm = torch.from_numpy(pcm).cuda()
    labels = [None] * len(starts)

    def work(i):
        torch.cuda.set_device(0)
        a, b = starts[i] * 48000, min((starts[i] + win) * 48000, pcm.size)
        labels[i] = detect_device(dev_pcm[a:b], 100, 48000, 0.0)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(starts))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for s, lab in zip(starts, labels):
        assert np.array_equal(lab.cpu().numpy(), full[s * 100: s * 100 + lab.numel()])
    labels[-1] = 1.0 - labels[-1]  # make the order of the two overlapping windows matter: the later one wins
    duration = float(total_s) - 2.0  # ... and let the vector end inside the last two windows (clipping)
    sparse = assemble_sparse_reference(labels, starts, duration, 100).cpu().numpy()
    want = np.zeros(int(duration * 100) + 2)
    for s, lab in zip(starts, labels):  # the reference's loop, speech_transformers.py:886-890
        seg = lab.cpu().numpy()
        begin = int(s * 100)
        end = min(begin + len(seg), len(want))
        if end > begin:
            want[begin:end] = seg[: end - begin]
    assert sparse.dtype == np.float32 and np.array_equal(sparse, want.astype(np.float32)) and sparse.sum() > 0
    with pytest.raises(ValueError, match="Unable to detect speech in any sampled segment"):
        assemble_sparse_reference([torch.zeros(100, device="cuda")], [3], 60.0, 100)


def test_bulk_npz_feeder_and_aligner(torch, tmp_path):
    """load_speech_batch: .npz/.npy -> bit-packed HBM vectors (DeserializeSpeechTransformer semantics), usable
    as the reference of a solve."""
    from ffsubsync_amd.aligners import FFTAligner
    from ffsubsync_amd.speech_transformers import DeserializeSpeechTransformer, load_speech_batch, serialize_speech
    from workloads import synth

    files, hosts = [], []
    for i in range(3):
        ref, sub = synth.simple_pair(50000 + 37 * i, 42000, 250 + i, seed=60 + i)
        speech = np.where(ref > 0, 1.0, [0.0, 0.3, 0.7][i])  # serialized labels below 1 become non_speech_label
        path = str(tmp_path / ("ref%d.npz" % i))
        serialize_speech(path, speech)
        files.append(path)
        hosts.append((sub, DeserializeSpeechTransformer(-0.25).fit(path).transform()))
    np.save(str(tmp_path / "w.npy"), np.array([0.0, 1.0, 1.5, 1.0]))
    rasters = load_speech_batch(files + [str(tmp_path / "w.npy")], non_speech_label=-0.25)
    assert isinstance(rasters[3], np.ndarray) and rasters[3].tolist() == [-0.25, 1.0, 1.5, 1.0]  # not two-level
    for r, (sub, host) in zip(rasters[:3], hosts):
        assert r.packed and len(r) == host.size and (r.lo, r.hi) == (-0.25, 1.0)
        assert np.array_equal(np.asarray(r), host)
        got = FFTAligner(6000).fit_transform(r, sub, get_score=True)
        exp = orc.fft_align(host, sub, 6000)
        assert got[1] == exp[1] and float(got[0]) == pytest.approx(float(exp[0]), rel=1e-9)


def test_chunked_detector_loop_over_a_pipe(torch):
    """PCMSpeechTransformer with the auditok-style GPU detector fed by a pipe-like object: 100 s buffers
    (speech_transformers.py:683-685), one tokenizer pass per buffer, monotonic progress, float64 labels."""
    from ffsubsync_amd.speech_transformers import PCMSpeechTransformer

    pcm, _ = vo.synth_pcm(480 * 23000 + 11, seed=12)

    class Pipe:
        def __init__(self, raw):
            self.raw, self.pos, self.reads = raw, 0, []

        def read(self, n):
            self.reads.append(n)
            blob = self.raw[self.pos:self.pos + n]
            self.pos += len(blob)
            return blob

    seen = []
    pipe = Pipe(pcm.tobytes())
    t = PCMSpeechTransformer("subs_then_auditok", 100, 48000, 0.0, progress_handler=seen.append)
    assert t.fit(pipe) is t
    assert set(pipe.reads) == {960 * 10000}
    want = np.concatenate([vo.tokenize_chunk(vo.detect_fast(pcm[o:o + 4800000]) > 0.5, 0.0)
                           for o in range(0, pcm.size, 4800000)])
    assert np.array_equal(t.transform(), want) and t.video_speech_results_.dtype == np.float64
    assert len(seen) == 3 and seen == sorted(seen)
    with pytest.raises(ValueError, match="unknown vad"):
        PCMSpeechTransformer("nonsense", 100, 48000, 0.0).fit(io.BytesIO(b"\0" * 9600))
    with pytest.raises(ValueError, match="Unable to detect speech"):
        PCMSpeechTransformer("energy", 100, 48000, 0.0).fit(io.BytesIO(b""))


def test_calls_on_two_streams_share_one_plan(torch):
    """A plan's workspace is reused by every call; calls issued on different streams are ordered by the
    library, so alternating streams gives the same records as one stream."""
    from ffsubsync_amd import batch
    from workloads import synth

    specs = [synth.make_pair_spec(800 + i, duration_s=1800.0) for i in range(6)]
    db = synth.build_device_batch(specs)
    al = batch.BatchAligner(db.required_fft_length(6000), 7, 6000, pairs_in_flight=2)
    _, want = al.solve(db)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(6):
        with torch.cuda.stream(s1 if rep % 2 == 0 else s2):
            outs.append(al.solve_async(db, rep % 3 * 2, rep % 3 * 2 + 2))
    torch.cuda.synchronize()
    for rep, (_, pair_out) in enumerate(outs):
        got = pair_out.cpu().numpy().view(want.dtype)[:2]
        assert np.array_equal(got, want[rep % 3 * 2: rep % 3 * 2 + 2])
    al.plan.close()


def test_one_rank_rccl_gather_and_sharded_solve(torch):
    """The N > 1 path of bench.py with one rank, end to end on the GPU: torch.distributed 'nccl' (= RCCL)
    process group, ffs_comm_create bootstrapped through its store, the shard solved with the HIP path,
    ffs_gather_results (RCCL C API) == torch's all_gather_into_tensor == the local records."""
    import torch.distributed as dist

    from ffsubsync_amd import _native, batch
    from workloads import synth

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        comm = batch.make_comm(0, 1)
        specs = [synth.make_pair_spec(900 + i, duration_s=900.0) for i in range(5)]
        db = synth.build_device_batch(specs)
        lo, hi = batch.shard_bounds(len(specs), 0, 1)
        al = batch.BatchAligner(db.required_fft_length(6000), 7, 6000, pairs_in_flight=4)
        _, pair_out = al.solve_async(db, lo, hi)
        mine = batch.gather_pair_results(pair_out, len(specs), 1, comm=comm)
        theirs = batch.gather_pair_results(pair_out, len(specs), 1)
        torch.cuda.synchronize()
        assert torch.equal(mine, pair_out) and torch.equal(theirs, pair_out)
        pres = mine.cpu().numpy().view(_native.PAIR_RESULT_DTYPE)
        for p, sp in enumerate(specs):
            assert int(pres[p]["best_cand"]) == sp.true_ratio_index
        comm.close()
        al.plan.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_cand", [7, 5, 3, 1])
def test_half_rows_of_a_single_candidate_slot_equal_all_rows(torch, monkeypatch, n_cand):
    """HALF_LAST: with an odd candidate count the last packed transform holds one real candidate and only rows
    0..N1/2 of it are stored / transformed / read (the last pass rebuilds the rest by conjugation).  Records
    must be identical to FFS_DISABLE_HALF_LAST=1 on every pipeline: block-segmented, single transform with
    the pruned and with the full last pass, and a windowless solve (full last pass)."""
    from ffsubsync_amd import batch
    from workloads import synth

    specs = [synth.make_pair_spec(1200 + i, duration_s=d) for i, d in enumerate((7200.0, 7000.0, 5400.0))]
    db7 = synth.build_device_batch(specs)
    cols = list(range(n_cand))
    pick = lambda a: np.ascontiguousarray(a[:, [0] + [1 + j for j in cols]])
    db = batch.DeviceBatch(db7.data, pick(db7.offs), pick(db7.lens), pick(db7.lo), pick(db7.hi), db7.dtype)

    def solve(max_offset, env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        al = batch.BatchAligner(db.required_fft_length(max_offset), n_cand, max_offset, pairs_in_flight=2)
        out = al.solve(db)
        al.plan.close()
        for k in env:
            monkeypatch.delenv(k)
        return out

    for max_offset, extra in ((6000, {}), (6000, {"FFS_DISABLE_SEGMENTED": "1"}),
                              (6000, {"FFS_DISABLE_SEGMENTED": "1", "FFS_DISABLE_PRUNED_PASS_C": "1"}), (None, {})):
        half = solve(max_offset, dict(extra))
        full = solve(max_offset, dict(extra, FFS_DISABLE_HALF_LAST="1"))
        for f in ("score", "offset", "flags"):
            assert np.array_equal(half[0][f], full[0][f]), (n_cand, max_offset, extra, f)
        assert np.array_equal(half[1], full[1])
        assert np.abs(half[0]["score_f32"].astype(np.float64) - half[0]["score"]).max() < 0.5
        if n_cand == 7:
            for p, sp in enumerate(specs):
                assert int(half[1][p]["best_cand"]) == sp.true_ratio_index


@pytest.mark.parametrize("n_cand", [7, 8, 4, 2])
def test_mid_pass_variants_give_identical_records(torch, monkeypatch, n_cand):
    """The default block-segmented mid pass (k_mid_seg_pipe: row loads issued one item ahead, LDS-only
    barriers, mirror-row pairs on one XCD) against the plain two-slot k_mid_seg with rows in index order,
    with and without the half last slot, for even and odd candidate counts."""
    from ffsubsync_amd import batch
    from workloads import synth

    ratios = None
    if n_cand == 8:  # the inferred-length eighth candidate of subtitle references (ffsubsync.py:205-222)
        from ffsubsync_amd.constants import candidate_ratios

        ratios = candidate_ratios() + [1.0213]
    specs = [synth.make_pair_spec(1300 + i, duration_s=d, ratios=ratios) for i, d in enumerate((7200.0, 6800.0, 4000.0))]
    db8 = synth.build_device_batch(specs)
    pick = lambda a: np.ascontiguousarray(a[:, : 1 + n_cand])
    db = batch.DeviceBatch(db8.data, pick(db8.offs), pick(db8.lens), pick(db8.lo), pick(db8.hi), db8.dtype)

    def solve(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        al = batch.BatchAligner(db.required_fft_length(6000), n_cand, 6000, pairs_in_flight=2)
        out = al.solve(db)
        al.plan.close()
        for k in env:
            monkeypatch.delenv(k)
        return out

    base = solve({"FFS_DISABLE_SEGMENTED": "1"})  # one transform of the full length: k_mid instead of the segmented kernels
    for env in ({}, {"FFS_DISABLE_HALF_LAST": "1"}):  # k_mid_seg_one (<= 4 packed slots) / k_mid_seg_pipe (more)
        got = solve(env)
        for f in ("score", "offset", "flags"):
            assert np.array_equal(base[0][f], got[0][f]), (env, f)
        assert np.array_equal(base[1], got[1])
        assert np.abs(got[0]["score_f32"].astype(np.float64) - got[0]["score"]).max() < 0.5
    for i, sp in enumerate(specs):  # and they are the right answers (when the true ratio is among the candidates)
        assert sp.true_ratio_index >= n_cand or int(base[1][i]["best_cand"]) == sp.true_ratio_index


def test_float64_inputs_are_rescored_from_the_callers_samples(torch):
    """FFS_DTYPE_F64: sample values that fp32 cannot represent (0.1, 0.3, 0.7 ...) -- the transforms nominate
    in fp32, but the reported score is the fp64 sum over the caller's own float64 samples: it agrees with the
    reference arithmetic to fp64 rounding (1e-12 relative here), not merely to the 1e-5 contract."""
    from ffsubsync_amd import _native
    from ffsubsync_amd.aligners import FFTAligner, _Vec, solve_pairs

    rng = np.random.RandomState(17)
    lv = np.array([0.0, 0.1, 0.3, 0.7, 1.0])
    ref = np.repeat(lv[rng.randint(0, 5, 900)], 23)
    sub = np.concatenate([np.zeros(211), ref[:15000]]) * 0.9
    assert not _Vec(ref).two_level
    for mo in (None, 6000, 300):
        got = FFTAligner(mo).fit_transform(ref, sub, get_score=True)
        exp = orc.fft_align(ref, sub, mo)
        assert got[1] == exp[1] == -211
        assert float(got[0]) == pytest.approx(float(exp[0]), rel=1e-12)
    # the same through the raw ABI with fp32 copies: still inside the contract, but only to fp32 input rounding
    d = lambda x, t: torch.from_numpy(np.ascontiguousarray(x, dtype=t)).cuda()
    n_fft = _native.plan_length(ref.size, sub.size, None)
    plan = _native.Plan(n_fft, 1, 1)
    cand_out = torch.empty(24, dtype=torch.uint8, device="cuda")
    pair_out = torch.empty(24, dtype=torch.uint8, device="cuda")
    scores = {}
    for name, dt, t in (("f32", _native.FFS_DTYPE_F32, np.float32), ("f64", _native.FFS_DTYPE_F64, np.float64)):
        r_dev, s_dev = d(ref, t), d(sub, t)
        plan.align_batch(1, 1, dt, np.array([r_dev.data_ptr(), s_dev.data_ptr()], np.uint64), np.array([ref.size, sub.size]),
                         np.array([0.0, 0.0]), np.array([1.0, 0.9]), None, None, cand_out, pair_out)
        res = cand_out.cpu().numpy().view(_native.CAND_RESULT_DTYPE)[0]
        assert int(res["offset"]) == -211
        scores[name] = float(res["score"])
    exact = float(orc.fft_align(ref, sub, None)[0])
    assert abs(scores["f64"] - exact) <= 1e-12 * abs(exact)
    assert 1e-12 * abs(exact) < abs(scores["f32"] - exact) <= 1e-5 * abs(exact)
    plan.close()


@pytest.mark.parametrize("case", ["n192_window", "n384_none", "n768_none", "n512_none", "n512_reference_length"])
def test_radix3_columns_in_registers_give_identical_records(torch, monkeypatch, case):
    """Plans with 3*2^k columns (N1 = 192 / 384 / 768) and with 512-row columns (N = 2^21): k_pass_a3 / k_pass_c3
    keep the three (two) sub-transforms of a column in one thread (last radix step in registers).  Same records as
    the power-of-two plan of the next length up (FFS_DISABLE_RADIX3=1; none for the 2^21 cases, which are compared
    with the CPU oracle), with and without the half slot and the pruned last pass."""
    from ffsubsync_amd import batch
    from workloads import synth

    dur = {"n192_window": 7200.0, "n384_none": 7000.0, "n768_none": 14000.0, "n512_none": 9200.0,
           "n512_reference_length": 7200.0}[case]
    specs = [synth.make_pair_spec(2100 + i, duration_s=dur - 300.0 * i) for i in range(2)]
    db = synth.build_device_batch(specs)
    max_offset = 6000 if case in ("n192_window", "n512_reference_length") else None
    n_fft = db.required_fft_length(max_offset, reference_length=(case == "n512_reference_length"))
    assert n_fft == {"n192_window": 786432, "n384_none": 1572864, "n768_none": 3145728, "n512_none": 2097152,
                     "n512_reference_length": 2097152}[case]

    def solve(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        al = batch.BatchAligner(n_fft, 7, max_offset, pairs_in_flight=2)
        out = al.solve(db)
        al.plan.close()
        for k in env:
            monkeypatch.delenv(k)
        return out

    seg_off = {"FFS_DISABLE_SEGMENTED": "1"}  # the windowed case would otherwise run block-segmented (power-of-two columns)
    if n_fft % 3 == 0:  # baseline: the power-of-two plan of the same problem
        monkeypatch.setenv("FFS_DISABLE_RADIX3", "1")
        n_pow2 = db.required_fft_length(max_offset)
        monkeypatch.delenv("FFS_DISABLE_RADIX3")
        assert n_pow2 & (n_pow2 - 1) == 0 and n_pow2 > n_fft
        al = batch.BatchAligner(n_pow2, 7, max_offset, pairs_in_flight=2)
        base = al.solve(db)
        al.plan.close()
    else:
        base = solve(dict(seg_off))
        from oracle import aligners_oracle as orc

        for i, sp in enumerate(specs):
            ref, cands = synth.pair_float_arrays(sp)
            (score, offset), idx = orc.max_score_align(ref, cands, max_offset)
            assert (int(base[1][i]["best_cand"]), int(base[1][i]["offset"])) == (idx, offset)
            assert abs(float(base[1][i]["score"]) - score) <= 1e-5 * abs(score)
    for env in (dict(seg_off), dict(seg_off, FFS_DISABLE_HALF_LAST="1"), dict(seg_off, FFS_DISABLE_PRUNED_PASS_C="1")):
        got = solve(env)
        for f in ("score", "offset", "flags"):
            assert np.array_equal(base[0][f], got[0][f]), (env, f)
        assert np.array_equal(base[1], got[1])
        assert np.abs(got[0]["score_f32"].astype(np.float64) - got[0]["score"]).max() < 1.0
    for i, sp in enumerate(specs):  # and they are the right answers
        assert int(base[1][i]["best_cand"]) == sp.true_ratio_index
        assert abs(int(base[1][i]["offset"]) - sp.true_offset_samples) <= 30



def test_pass_a_prefetch_blocks_change_nothing_but_time(torch, monkeypatch):
    """The input prefetch blocks of pass A (eight extra workgroups per grid row that touch the bit-packed vectors of a
    transform a few rows ahead) only warm L2: records with and without them are identical, window or not."""
    from ffsubsync_amd import batch
    from workloads import synth

    specs = [synth.make_pair_spec(2200 + i, duration_s=d) for i, d in enumerate((7200.0, 6999.7, 3611.3))]
    db = synth.build_device_batch(specs)
    for max_offset in (6000, None):
        out = []
        for pf in ("0", "12", "200"):
            monkeypatch.setenv("FFS_PASS_A_PREFETCH", pf)
            al = batch.BatchAligner(db.required_fft_length(max_offset), 7, max_offset, pairs_in_flight=2)
            out.append(al.solve(db))
            al.plan.close()
        monkeypatch.delenv("FFS_PASS_A_PREFETCH")
        for got in out[1:]:
            for f in ("score", "offset", "flags", "score_f32"):
                assert np.array_equal(out[0][0][f], got[0][f]), (max_offset, f)
            assert np.array_equal(out[0][1], got[1])


def test_scan_tokenizer_equals_serial_kernel_and_restatement(torch, monkeypatch):
    """ffs_vad_tokenize runs one workgroup per chunk (k_vad_tokenize_scan: bit words and islands) for chunks of up to
    28672 frames and max_length >= min_length, the one-thread-per-chunk state machine (k_vad_tokenize) otherwise (the
    30011-frame chunks below, and the max_length < min_length case).  Both against the Python restatement: reference
    parameters and degenerate ones (no tolerated silence, one-frame tokens, silence longer than a token), several
    labels, chunk lengths around the 64-frame word and the segment size of the prefix sum, chunks too long for the
    workgroup kernel."""
    from ffsubsync_amd import _native

    rng = np.random.RandomState(21)
    cases = [(20, 500, 25), (3, 10, 2), (5, 5, 1), (1, 7, 0), (4, 40, 30), (2, 9, 9), (1, 1, 0), (0, 3, -1), (2, 70, 100), (9, 4, 2)]
    for trial in range(int(os.environ.get("FFS_TOK_TRIALS", "20"))):
        n = int(rng.choice([1, 63, 64, 65, 255, 256, 257, 3000, 10000, 20480, 28672, 30011]))
        p_on = rng.choice([0.02, 0.2, 0.6, 0.95])
        runs = rng.geometric(1.0 / rng.choice([1, 3, 15, 80, 700]), size=n + 4)
        valid = np.repeat(rng.rand(runs.size) < p_on, runs)[:n]
        dev = torch.from_numpy(valid.astype(np.float32)).cuda()
        mn, mx, msil = cases[trial % len(cases)]
        for label in (0.0, 0.25, -1.0):
            for chunk in (10000, 997, 20480, 25000, 28672, 30011):
                tok = lambda c: vo._Tokenizer(mn, mx, msil).tokenize(c)
                want = []
                for o in range(0, n, chunk):
                    c = valid[o:o + chunk]
                    marks = np.zeros(c.size + 1)
                    for s, e in tok(c):
                        marks[s] = 1.0
                        marks[e + 1] = label - 1.0
                    want.append(np.clip(np.cumsum(marks)[:-1], 0.0, 1.0))
                want = np.concatenate(want)
                got = _native.vad_tokenize(dev, chunk, mn, mx, msil, label).cpu().numpy().astype(float)
                assert np.array_equal(got, want), (trial, n, (mn, mx, msil), label, chunk)


@pytest.mark.parametrize("n_cand", [1, 2])
def test_one_slot_mid_kernel_gives_identical_records(torch, monkeypatch, n_cand):
    """Solves with one packed slot (every FFTAligner.fit: one or two candidates) run the block-segmented mid pass with
    ONE accumulator row and conj(R)/N in registers (k_mid_seg_one<.., 1>: three blocks per CU).  Records identical to
    the single-transform pipeline's."""
    from ffsubsync_amd import batch
    from workloads import synth

    specs = [synth.make_pair_spec(2300 + i, duration_s=d) for i, d in enumerate((7200.0, 6800.0, 4000.0))]
    db8 = synth.build_device_batch(specs)
    pick = lambda a: np.ascontiguousarray(a[:, : 1 + n_cand])
    db = batch.DeviceBatch(db8.data, pick(db8.offs), pick(db8.lens), pick(db8.lo), pick(db8.hi), db8.dtype)
    out = []
    for env in ("1", None):  # single-transform pipeline (k_mid) vs the block-segmented one-slot kernel
        if env is not None:
            monkeypatch.setenv("FFS_DISABLE_SEGMENTED", env)
        al = batch.BatchAligner(db.required_fft_length(6000), n_cand, 6000, pairs_in_flight=2)
        out.append(al.solve(db))
        al.plan.close()
        if env is not None:
            monkeypatch.delenv("FFS_DISABLE_SEGMENTED")
    for f in ("score", "offset", "flags"):
        assert np.array_equal(out[0][0][f], out[1][0][f]), f
    assert np.array_equal(out[0][1], out[1][1])
    assert np.abs(out[1][0]["score_f32"].astype(np.float64) - out[1][0]["score"]).max() < 0.5


def test_two_streams_give_identical_records(torch):
    """BatchAligner(streams=2): the pairs of a call split over two plans on two HIP streams, ordered by the caller's
    current stream -- same records as one stream, also when the call is issued on a non-default stream and the
    results are consumed right away on it."""
    from ffsubsync_amd import batch
    from workloads import synth

    specs = [synth.make_pair_spec(2400 + i, duration_s=1800.0 + 37.0 * i) for i in range(11)]
    db = synth.build_device_batch(specs)
    n_fft = db.required_fft_length(6000)
    one = batch.BatchAligner(n_fft, 7, 6000, pairs_in_flight=4)
    want = one.solve(db)
    one.close()
    two = batch.BatchAligner(n_fft, 7, 6000, pairs_in_flight=4, streams=2)
    got = two.solve(db)
    for f in ("score", "offset", "flags", "score_f32"):
        assert np.array_equal(want[0][f], got[0][f]), f
    assert np.array_equal(want[1], got[1])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            c, p = two.solve_async(db, 2, 9)
            snap = p.clone()  # consumed on the caller's stream right behind the call
    side.synchronize()
    assert np.array_equal(snap.cpu().numpy().view(got[1].dtype)[:7], want[1][2:9])
    two.close()
