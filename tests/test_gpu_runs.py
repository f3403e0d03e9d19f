"""The run-boundary path (csrc/ffs_runs.h: exact correlation of run-length-coded activity vectors, FFS_ALGORITHM=auto)
against the transform path, the CPU oracle and the unmodified reference's goldens.  Through the C ABI; need a real MI355X.

"Identical records" = every field of ffs_cand_result / ffs_pair_result except score_f32 (the fp32 transform's value at
the winning lag, which only the transform path has; the run-boundary path stores the exact score rounded to fp32).
"""
import json
import os

import numpy as np
import pytest

import golden_cases
from oracle import aligners_oracle as orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
HEAD = json.load(open(os.path.join(HERE, "golden", "headline_golden.json")))["pairs"]
GOLD = json.load(open(os.path.join(HERE, "golden", "aligner_golden.json")))
SMALL = golden_cases.build_cases(include_large=False)


@pytest.fixture(scope="module")
def torch():
    import torch as t

    assert t.cuda.is_available()
    return t


def _same_records(a, b):
    ca, pa = a
    cb, pb = b
    for f in ("score", "offset", "flags"):
        assert np.array_equal(ca[f], cb[f]), (f, np.argwhere(ca[f] != cb[f])[:5])
    assert np.array_equal(pa, pb)


def _solve(db, n_fft, max_off, algorithm, n_cand=7, pairs_in_flight=64):
    from ffsubsync_amd import batch

    al = batch.BatchAligner(n_fft, n_cand, max_off, pairs_in_flight=pairs_in_flight, algorithm=algorithm)
    out = al.solve(db)
    stats = al.plan.runs_stats()
    al.close()
    return out, stats


@pytest.fixture(scope="module")
def headline(torch):
    from workloads import synth

    n = int(os.environ.get("FFS_RUNS_PAIRS", "128"))
    specs = [synth.make_pair_spec(g["seed"]) for g in HEAD[:n]]
    db = synth.build_device_batch(specs)
    yield specs, db, HEAD[:n]
    del db
    torch.cuda.empty_cache()


def test_headline_pairs_equal_the_reference_and_the_transform_path(headline):
    """configs[2] at the timed shape: auto takes the run-boundary path for every sub-batch; records identical to the
    transform path and to the unmodified reference's goldens (same rule as tests/test_gpu_headline.py)."""
    import test_gpu_headline as th

    specs, db, gold = headline
    n_fft = db.required_fft_length(6000)
    (c_auto, p_auto), st_auto = _solve(db, n_fft, 6000, "auto")
    (c_fft, p_fft), st_fft = _solve(db, n_fft, 6000, "fft")
    assert st_auto[0] == 1 and st_auto[1] == 2 and st_auto[2] == 0  # one call, two sub-batches, none through the transforms
    assert st_fft == (0, 0, 0)
    _same_records((c_auto, p_auto), (c_fft, p_fft))
    th._check_seven(p_auto, c_auto, gold)
    # the exact score is what both paths report; the run-boundary path stores it (rounded) as score_f32 too
    assert np.array_equal(c_auto["score_f32"], c_auto["score"].astype(np.float32))


@pytest.mark.parametrize("max_off", [None, 6000, 150000])
def test_single_ratio_and_wide_windows(headline, max_off):
    """FFTAligner() / FFTAligner(6000) on the true-ratio candidate (configs[1]) and a window of 25 tiles: windows wider
    than one tile (12 288 lags) are cut into tiles whose results k_runs_pick combines."""
    specs, db, gold = headline
    from workloads import synth

    db32 = synth.build_device_batch(specs[:32])
    one = db32.select_candidates([sp.true_ratio_index for sp in specs[:32]])
    n_fft = one.required_fft_length(max_off)
    # "runs": no coincidence budget -- this test is about the tiles, not about the choice
    a, st = _solve(one, n_fft, max_off, "runs", n_cand=1, pairs_in_flight=32)
    b, _ = _solve(one, n_fft, max_off, "fft", n_cand=1, pairs_in_flight=32)
    assert st[2] == 0
    _same_records(a, b)
    key = {None: "single_none", 6000: "single_6000"}.get(max_off)
    if key:
        for i, g in enumerate(gold[:32]):
            assert int(a[0][i, 0]["offset"]) == g[key][1]
            assert float(a[0][i, 0]["score"]) == pytest.approx(float(g[key][0]), rel=1e-5)


def test_seven_ratios_without_a_window(headline):
    """max_offset_samples=None, seven ratios: 118 tiles per candidate on the run-boundary path (where round 5 moved this
    configuration), the transforms and `auto` -- identical records, and all of them equal to what the UNMODIFIED reference's
    ``MaxScoreAligner(FFTAligner())`` (aligners.py:25-29 default constructor, no filter at :156) returns for the same bench
    seeds (tests/golden/windowless_golden.json, make_windowless_golden.py): winner bit-identical, every score within 1e-5,
    per-candidate offsets bit-identical wherever the reference's top-2 gap exceeds 0.5 and inside the reference's own
    plateau of near-maximal lags otherwise."""
    from workloads import golden_check, synth

    wl = golden_check.load("windowless_golden")
    n = int(os.environ.get("FFS_WINDOWLESS_PAIRS", "64"))
    seeds = sorted(wl)[:n]
    assert len(seeds) >= min(n, 64)
    dbn = synth.build_device_batch([synth.make_pair_spec(s) for s in seeds])
    n_fft = dbn.required_fft_length(None)
    a, st = _solve(dbn, n_fft, None, "runs", pairs_in_flight=32)
    assert st[2] == 0
    b, st_b = _solve(dbn, n_fft, None, "fft", pairs_in_flight=32)
    assert st_b == (0, 0, 0)
    _same_records(a, b)
    # auto: ~6.6 M boundary coincidences per candidate against a budget of twelve per transform point and slot (10.8 M):
    # whatever the library picks, the records are the same
    c, st_auto = _solve(dbn, n_fft, None, "auto", pairs_in_flight=32)
    assert st_auto[0] == 1
    _same_records(c, b)
    ok, total, first = golden_check.matching(wl, seeds, a[1], a[0])
    assert total == len(seeds) and ok == total, first
    ties = sum(golden_check.count_ties(wl[s]) for s in seeds)
    assert ties < 0.2 * 7 * len(seeds)  # the plateau rule stays an exception
    # the drop-in classes on pair 0's host arrays, constructed exactly as the golden's generator constructs the reference's
    from ffsubsync_amd.aligners import FFTAligner, MaxScoreAligner

    ref, cands = synth.pair_float_arrays(synth.make_pair_spec(seeds[0]))
    msa = MaxScoreAligner(FFTAligner())
    assert msa.max_offset_samples is None
    (score, offset), winner = msa.fit_transform(ref, list(cands))
    g0 = wl[seeds[0]]
    assert winner is cands[g0["index"]] and int(offset) == g0["offset"]
    assert float(score) == pytest.approx(float(g0["score"]), rel=1e-5)


def test_plan_owned_lists_grow_with_the_data(torch):
    """Round 6: the plan's boundary lists (vectors that arrive as bits) start at 4 096 entries per vector and the call is
    solved again with four times the room when a vector has more (at most twice, up to 32 768): vectors with ~6 800 and
    ~13 600 boundaries take the run-boundary path on the FIRST call exactly as vectors with ~1 700 do, records identical to
    the transforms'; the plan keeps the longer stride (the second call extracts once)."""
    from ffsubsync_amd import batch
    from workloads import synth

    for scale in (0.25, 0.125):
        db = synth.build_device_batch([synth.make_pair_spec(7100 + i, run_scale=scale) for i in range(6)])
        n_fft = db.required_fft_length(6000)
        want, _ = _solve(db, n_fft, 6000, "fft", pairs_in_flight=6)
        al = batch.BatchAligner(n_fft, 7, 6000, pairs_in_flight=6, algorithm="runs")
        for call in range(2):
            got = al.solve(db)
            assert al.plan.runs_stats() == (call + 1, call + 1, 0)  # run-boundary path, nothing through the transforms
            _same_records(got, want)
            # (boundaries of the call's 48 vectors: every list is longer than the first stride)
            assert al.plan.runs_boundaries_last_call() > 48 * 4096 * (1 if scale == 0.25 else 2)
        al.close()


@pytest.mark.parametrize("name", sorted(SMALL))
def test_golden_cases_through_the_dropin_classes(torch, name, monkeypatch):
    """Every reference golden (KATs, lag-window semantics incl. Python negative slices and all-masked windows, float
    inputs, short inputs) through the drop-in classes with the run-boundary path enabled."""
    import test_gpu_parity as tp

    monkeypatch.setenv("FFS_ALGORITHM", "auto")
    tp.test_golden_cases_through_the_dropin_classes(torch, name)


def test_large_goldens_and_gss_trace(torch, monkeypatch):
    import test_gpu_parity as tp

    monkeypatch.setenv("FFS_ALGORITHM", "auto")
    tp.test_headline_shape_2h_seven_ratios(torch)
    tp.test_gss_through_the_dropin(torch)
    tp.test_extreme_densities_stay_exact(torch)
    tp.test_exact_tie_rule_is_first_maximum_in_k(torch)


def _pair(rng, R, S, run_mean, dens, n_cand, amp_choices=(1.0, 0.96, 0.999)):
    from ffsubsync_amd.aligners import _Vec

    seg = np.maximum(1, rng.geometric(1.0 / run_mean, size=R))
    ref01 = np.repeat(rng.rand(seg.size) < dens, seg)[:R]
    lo_hi_ref = [(0.0, 1.0), (-1.0, 1.0), (0.25, 1.0)][rng.randint(3)]
    ref = np.where(ref01, lo_hi_ref[1], lo_hi_ref[0])
    cands = []
    for _ in range(n_cand):
        off = int(rng.randint(-S // 2, R // 2 + 1))
        idx = np.arange(S) + off
        ok = (idx >= 0) & (idx < R)
        c01 = np.zeros(S, bool)
        c01[ok] = ref01[idx[ok]]
        c01 ^= rng.rand(S) < rng.choice([0.0, 0.002, 0.05])
        cands.append(c01 * amp_choices[rng.randint(len(amp_choices))])
    return ref, cands, (_Vec(ref), [_Vec(c) for c in cands])


def test_randomised_problems_both_paths_and_oracle(torch, monkeypatch):
    """Fuzz over lengths (incl. multiples of 32 and vectors whose last sample is set), run lengths from 2 (lists longer
    than the LDS staging area, read from global memory) to 2000, densities, levels, windows (None, narrow, wider than
    the data, Python-negative-slice) and candidate counts: run-boundary records == transform records, and both against
    the oracle (offset whenever its top-2 gap exceeds 0.5, score always)."""
    from ffsubsync_amd.aligners import solve_pairs

    trials = int(os.environ.get("FFS_FUZZ_TRIALS", "60"))
    rng = np.random.RandomState(4242)
    for trial in range(trials):
        R = int(rng.choice([rng.randint(4200, 9000), rng.randint(9000, 70000), 32 * rng.randint(200, 2000)]))
        S = int(max(2100, R * rng.uniform(0.4, 1.5)))
        if trial % 6 == 0:
            S = S // 32 * 32
        run_mean = int(rng.choice([2, 5, 50, 400, 2000]))
        n_cand = int(rng.choice([1, 2, 3, 7]))
        ref, cands, pv = _pair(rng, R, S, run_mean, rng.choice([0.05, 0.4, 0.9]), n_cand)
        if trial % 9 == 0:
            cands[0] = cands[0] * 0  # a silent candidate
            pv = (pv[0], [type(pv[0])(c) for c in cands])
        mo = [None, None, 6000, int(rng.randint(0, 3 * R)), int(rng.randint(1, 200))][rng.randint(5)]
        monkeypatch.setenv("FFS_ALGORITHM", "runs")
        c_r, p_r = solve_pairs([pv], mo, mo)
        monkeypatch.setenv("FFS_ALGORITHM", "fft")
        c_f, p_f = solve_pairs([pv], mo, mo)
        _same_records((c_r, p_r), (c_f, p_f))
        for j, c in enumerate(cands):
            conv, S_ = orc.convolve_full(ref, c)
            m = orc.mask_extreme_offsets(conv, S_, mo)
            k = int(np.argmax(m))
            s_o, o_o = m[k], len(m) - 1 - k - S_
            got_s, got_o = float(c_r[0, j]["score"]), int(c_r[0, j]["offset"])
            if not np.isfinite(s_o):
                assert got_s == -np.inf and got_o == o_o, (trial, j)
                continue
            assert got_s == pytest.approx(s_o, rel=1e-9, abs=1e-6), (trial, j, R, S, mo)
            fin = np.sort(m[np.isfinite(m)])
            if fin.size < 2 or fin[-1] - fin[-2] > 0.5:
                assert got_o == o_o, (trial, j, R, S, mo, got_o, o_o)


def test_dense_vectors_fall_back_to_the_transforms_per_sub_batch(torch):
    """Four 20-minute pairs, two per sub-batch; pair 2's reference is random bits (60 000 boundaries: its list is
    truncated): auto solves sub-batch 0 by run boundaries and sub-batch 1 through the transforms, records identical to
    the all-transform solve; with a zero coincidence budget everything goes through the transforms."""
    from ffsubsync_amd import _native, batch
    from ffsubsync_amd.batch import DeviceBatch

    rng = np.random.RandomState(5)
    R = S = 120000
    vecs, lens = [], []
    for p in range(4):
        dense = p == 2
        ref = (rng.rand(R) < 0.5) if dense else np.repeat(rng.rand(R // 300 + 1) < 0.4, 300)[:R]
        off = int(rng.randint(-3000, 3000))
        idx = np.arange(S) + off
        ok = (idx >= 0) & (idx < R)
        c = np.zeros(S, bool)
        c[ok] = ref[idx[ok]]
        c2 = np.roll(c, 777)
        vecs += [ref, c, c2]
        lens.append([R, S, S])
    sizes = [(v.size + 7) // 8 for v in vecs]
    offs = np.zeros(len(vecs), np.int64)
    tot = 0
    for i, s in enumerate(sizes):
        offs[i] = tot
        tot += (s + 63) // 64 * 64
    host = np.zeros(tot, np.uint8)
    for v, o in zip(vecs, offs):
        pk = np.packbits(v.astype(np.uint8), bitorder="little")
        host[o:o + pk.size] = pk
    data = torch.from_numpy(host).cuda()
    shape = (4, 3)
    db = DeviceBatch(data, offs.reshape(shape), np.array(lens, np.int64), np.zeros(shape), np.ones(shape), _native.FFS_DTYPE_U1)
    n_fft = db.required_fft_length(6000)
    a, st = _solve(db, n_fft, 6000, "auto", n_cand=2, pairs_in_flight=2)
    b, _ = _solve(db, n_fft, 6000, "fft", n_cand=2, pairs_in_flight=2)
    assert st == (1, 2, 1)
    _same_records(a, b)
    os.environ["FFS_RUNS_BUDGET"] = "0"
    try:
        c, st0 = _solve(db, n_fft, 6000, "auto", n_cand=2, pairs_in_flight=2)
    finally:
        del os.environ["FFS_RUNS_BUDGET"]
    assert st0 == (1, 2, 2)
    _same_records(c, b)
    # "runs": no budget, but truncated lists still go through the transforms
    d, st1 = _solve(db, n_fft, 6000, "runs", n_cand=2, pairs_in_flight=2)
    assert st1 == (1, 2, 1)
    _same_records(d, b)


def test_host_batch_entry_equals_one_call_per_problem(torch):
    """aligners.solve_host_batch (many files' float64 arrays -> threaded packing -> one upload -> one ffs_align_batch)
    against MaxScoreAligner.fit_transform one problem at a time; a three-level reference in the batch sends that call
    down the float64 path, with the same answers."""
    from ffsubsync_amd.aligners import FFTAligner, MaxScoreAligner, solve_host_batch
    from workloads import synth

    specs = [synth.make_pair_spec(300 + i, duration_s=1800.0) for i in range(5)]
    problems = [synth.pair_float_arrays(sp) for sp in specs]
    cres, pres = solve_host_batch(problems, 6000, 6000)
    for i, (ref, cands) in enumerate(problems):
        cands = list(cands)
        (score, offset), winner = MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform(ref, cands)
        assert cands[int(pres[i]["best_cand"])] is winner and int(pres[i]["offset"]) == offset
        assert float(pres[i]["score"]) == float(score)
        assert int(pres[i]["best_cand"]) == specs[i].true_ratio_index
    ref3 = problems[0][0].copy()
    ref3[::1000] = 0.5
    c3, p3 = solve_host_batch([(ref3, problems[0][1])] + problems[1:], 6000, 6000)
    assert int(p3[0]["best_cand"]) == specs[0].true_ratio_index
    for f in ("best_cand", "offset"):
        assert np.array_equal(p3[f][1:], pres[f][1:])
    assert np.allclose(p3["score"][1:], pres["score"][1:], rtol=1e-12)


def _block_batch(torch, seed, n_pairs, n_cand=7, R=4608):
    """n_pairs short problems (runs of 64 samples, candidates = shifted copies) as one bit-packed DeviceBatch."""
    from ffsubsync_amd import _native
    from ffsubsync_amd.batch import DeviceBatch

    rng = np.random.RandomState(seed)
    base = np.repeat(rng.rand(n_pairs, R // 64) < 0.4, 64, axis=1)
    shifts = rng.randint(-300, 300, size=(n_pairs, n_cand))
    words_per = R // 32
    stride_b = (words_per * 4 + 63) // 64 * 64
    host = np.zeros((n_pairs, 1 + n_cand, stride_b), np.uint8)
    host[:, 0, : words_per * 4] = np.packbits(base, axis=1, bitorder="little")
    for j in range(n_cand):
        idx = (np.arange(R)[None, :] + shifts[:, j:j + 1]) % R
        c = np.take_along_axis(base, idx, axis=1)
        c[:, : 50 * j] ^= True if j == 3 else False
        host[:, 1 + j, : words_per * 4] = np.packbits(c, axis=1, bitorder="little")
    data = torch.from_numpy(host.reshape(-1)).cuda()
    offs = (np.arange(n_pairs * (1 + n_cand), dtype=np.int64) * stride_b).reshape(n_pairs, 1 + n_cand)
    lens = np.full((n_pairs, 1 + n_cand), R, np.int64)
    return DeviceBatch(data, offs, lens, np.zeros(offs.shape), np.ones(offs.shape), _native.FFS_DTYPE_U1)


def test_calls_beyond_65536_vectors_are_split(torch):
    """8200 seven-candidate problems = 65 600 vectors in one ffs_align_batch call: the run-boundary path solves them as two
    consecutive sub-calls (boundary-list workspace bounded at 65 536 vectors); records equal the transform path's."""
    db = _block_batch(torch, 11, 8200)
    n_fft = db.required_fft_length(600)
    a, st = _solve(db, n_fft, 600, "auto", pairs_in_flight=512)
    b, _ = _solve(db, n_fft, 600, "fft", pairs_in_flight=512)
    assert st[0] == 2 and st[2] == 0  # two sub-calls, nothing through the transforms
    _same_records(a, b)


def test_large_calls_alternate_descriptor_blocks_and_mix_with_small_ones(torch):
    """Calls of more than 4096 candidates upload their descriptors on the plan's copy stream into alternating device blocks
    (the next call's vector table goes up while this call's correlation runs); small calls keep the single upload on the
    call's stream.  Six calls queued back to back on ONE plan without a synchronisation in between -- three different
    large batches, a small one in the middle, one call on a side stream, a growing call last (the descriptor blocks are
    reallocated) -- every call's records equal the transform path's: a descriptor that arrived late, or in the block a
    running kernel still reads, would show as another call's answers."""
    from ffsubsync_amd import _native, batch

    sets = {"A": _block_batch(torch, 21, 700), "B": _block_batch(torch, 22, 700), "C": _block_batch(torch, 23, 640),
            "S": _block_batch(torch, 24, 40), "G": _block_batch(torch, 25, 1500)}
    n_fft = sets["A"].required_fft_length(600)
    want = {k: _solve(db, n_fft, 600, "fft", pairs_in_flight=128)[0] for k, db in sets.items()}
    al = batch.BatchAligner(n_fft, 7, 600, pairs_in_flight=128, algorithm="auto")
    side = torch.cuda.Stream()
    order = ["A", "B", "S", "C", "A", "B", "G", "A"]
    outs = []
    for i, k in enumerate(order):
        n = sets[k].offs.shape[0]
        co = torch.empty(n * 7 * 24, dtype=torch.uint8, device="cuda")
        po = torch.empty(n * 24, dtype=torch.uint8, device="cuda")
        if i == 4:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                al.solve_async(sets[k], 0, n, co, po)
            torch.cuda.current_stream().wait_stream(side)
        else:
            al.solve_async(sets[k], 0, n, co, po)
        outs.append((k, n, co, po))
    torch.cuda.synchronize()
    calls, _, fft_chunks = al.plan.runs_stats()
    assert calls == len(order) and fft_chunks == 0
    for k, n, co, po in outs:
        got = (co.cpu().numpy().view(_native.CAND_RESULT_DTYPE).reshape(n, 7), po.cpu().numpy().view(_native.PAIR_RESULT_DTYPE))
        _same_records(got, want[k])
    al.close()


def test_byte_inputs_are_packed_on_the_device_and_take_the_same_path(headline):
    """FFS_DTYPE_U8 batches (one 0/1 byte per frame, the north star's literal input format): auto packs every vector to
    bits in one launch and continues as FFS_DTYPE_U1 -- same records as the bit-packed batch and as the byte transform
    kernels (FFS_ALGO_FFT), run-boundary path taken."""
    from workloads import synth

    specs, db, gold = headline
    n = 24
    db8 = synth.build_device_batch(specs[:n], packed=False)
    db1 = synth.build_device_batch(specs[:n])
    n_fft = db8.required_fft_length(6000)
    a, st = _solve(db8, n_fft, 6000, "auto", pairs_in_flight=8)
    b, _ = _solve(db1, n_fft, 6000, "auto", pairs_in_flight=8)
    c, st_c = _solve(db8, n_fft, 6000, "fft", pairs_in_flight=8)
    assert st == (1, 3, 0) and st_c == (0, 0, 0)
    _same_records(a, b)
    _same_records(a, c)
