"""Parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle
and against the committed outputs of the unmodified reference.  Need a real MI355X."""
import json
import os

import numpy as np
import pytest

import golden_cases
from oracle import aligners_oracle as orc
from oracle import vad_oracle as vo

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "aligner_golden.json")))
SMALL = golden_cases.build_cases(include_large=False)
SCORE_RTOL = 1e-5  # north-star tolerance for float correlation scores


@pytest.fixture(scope="module")
def torch():
    import torch as t

    assert t.cuda.is_available()
    return t


def _direct_corr(ref_pm, s_pm, N):
    a = np.zeros(N)
    a[: len(s_pm)] = s_pm
    b = np.zeros(N)
    b[: len(ref_pm)] = ref_pm
    return np.real(np.fft.ifft(np.conj(np.fft.fft(a)) * np.fft.fft(b)))


@pytest.mark.parametrize("log2n", [12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24])
def test_raw_correlation_u8_matches_fp64(torch, log2n):
    """Every (N1, N2) kernel instantiation: full fp32 correlation vs numpy complex128."""
    from ffsubsync_amd import _native

    N = 1 << log2n
    rng = np.random.RandomState(log2n)
    R, Sa, Sb = N // 2 - 5, N // 2 - 77, N // 3 + 1
    ref = (rng.rand(R) < 0.35).astype(np.uint8)
    a = (rng.rand(Sa) < 0.35).astype(np.uint8)
    b = (rng.rand(Sb) < 0.6).astype(np.uint8)
    plan = _native.Plan(N, 1, 2)
    d = lambda x: torch.from_numpy(x).cuda()
    out_a, out_b = plan.correlate_full(_native.FFS_DTYPE_U8, d(ref), (0, 1), d(a), (0, 1), d(b), (0.0, 0.96))
    torch.cuda.synchronize()
    ea = _direct_corr(2.0 * ref - 1, 2.0 * a - 1, N)
    eb = _direct_corr(2.0 * ref - 1, 2.0 * (0.96 * b) - 1, N)
    tol = 1.0 * 6e-8 * log2n * np.sqrt(R * Sa)  # the data-independent part of the nominee margin
    err_a = np.abs(out_a.cpu().numpy() - ea).max()
    err_b = np.abs(out_b.cpu().numpy() - eb).max()
    print("N=2^%d max abs err a=%.4g b=%.4g (margin %.4g)" % (log2n, err_a, err_b, tol))
    assert err_a < tol / 8 and err_b < tol / 8  # fp32 error must stay well inside the margin
    # the same vectors bit-packed (FFS_DTYPE_U1): identical transform inputs, hence identical outputs
    pk = lambda x: _native.pack_bits(d(x))
    bit_a, bit_b = plan.correlate_full(_native.FFS_DTYPE_U1, pk(ref), (0, 1), pk(a), (0, 1), pk(b), (0.0, 0.96),
                                       lens=(R, Sa, Sb))
    if log2n != 21:
        assert torch.equal(bit_a, out_a) and torch.equal(bit_b, out_b)
    else:  # N1 = 512: bit-packed inputs take k_pass_a3<2, 256> (two sub-transforms per thread) -- same values, other rounding
        assert np.abs(bit_a.cpu().numpy() - ea).max() < tol / 8 and np.abs(bit_b.cpu().numpy() - eb).max() < tol / 8
    plan.close()


@pytest.mark.parametrize("n", [3 << 12, 3 << 13, 3 << 14, 3 << 15, 3 << 16, 3 << 17, 3 << 18, 3 << 19, 3 << 20])
def test_raw_correlation_three_times_power_of_two(torch, n):
    """Transform lengths 3*2^k (one radix-3 step in the column passes): N1 = 48 with N2 = 256..4096,
    then N1 = 96..768 with N2 = 4096; u8 and f32 inputs."""
    from ffsubsync_amd import _native

    rng = np.random.RandomState(n % 1009)
    R, Sa, Sb = n // 2 - 5, n // 2 - 77, n // 3 + 1
    ref = (rng.rand(R) < 0.35).astype(np.uint8)
    a = (rng.rand(Sa) < 0.35).astype(np.uint8)
    b = (rng.rand(Sb) < 0.6).astype(np.uint8)
    plan = _native.Plan(n, 1, 2)
    d = lambda x: torch.from_numpy(x).cuda()
    out_a, out_b = plan.correlate_full(_native.FFS_DTYPE_U8, d(ref), (0, 1), d(a), (0, 1), d(b), (0.0, 0.96))
    torch.cuda.synchronize()
    ea = _direct_corr(2.0 * ref - 1, 2.0 * a - 1, n)
    eb = _direct_corr(2.0 * ref - 1, 2.0 * (0.96 * b) - 1, n)
    tol = 1.0 * 6e-8 * np.log2(n) * np.sqrt(R * Sa)
    err_a = np.abs(out_a.cpu().numpy() - ea).max()
    err_b = np.abs(out_b.cpu().numpy() - eb).max()
    print("N=%d max abs err a=%.4g b=%.4g (margin %.4g)" % (n, err_a, err_b, tol))
    assert err_a < tol / 8 and err_b < tol / 8
    pk = lambda x: _native.pack_bits(d(x))
    bit_a, bit_b = plan.correlate_full(_native.FFS_DTYPE_U1, pk(ref), (0, 1), pk(a), (0, 1), pk(b), (0.0, 0.96),
                                       lens=(R, Sa, Sb))
    if n < 3 << 18:
        assert torch.equal(bit_a, out_a) and torch.equal(bit_b, out_b)
    else:  # N1 >= 192: bit-packed inputs take k_pass_a3 (radix-3 step in registers) -- same values, other rounding
        assert np.abs(bit_a.cpu().numpy() - ea).max() < tol / 8 and np.abs(bit_b.cpu().numpy() - eb).max() < tol / 8
        assert not torch.equal(bit_a, out_a)  # i.e. the three-sub-transforms-per-thread kernel did run
    if n <= 3 << 16:
        fa = rng.rand(Sa).astype(np.float32)
        out_f, _ = plan.correlate_full(_native.FFS_DTYPE_F32, d(ref.astype(np.float32)), (0, 1), d(fa), (0, 1))
        ef = _direct_corr(2.0 * ref - 1, 2.0 * fa.astype(float) - 1, n)
        assert np.abs(out_f.cpu().numpy() - ef).max() < tol / 4
    plan.close()


def test_three_times_power_of_two_plans_give_identical_records(torch, monkeypatch):
    """ffs_plan_length may pick 3*2^k; results must equal those of the power-of-two plan and of the
    reference-length plan, with the pruned and the full last pass."""
    from ffsubsync_amd import _native, batch
    from workloads import synth

    assert _native.plan_length(720000, 750751, 6000) == 3 << 18
    specs = [synth.make_pair_spec(500 + i, duration_s=d) for i, d in enumerate((7200.0, 6900.0, 3500.0, 1700.0))]
    for group in (specs[:2], specs[2:3], specs[3:]):
        db = synth.build_device_batch(group)
        n3 = db.required_fft_length(6000)
        assert n3 % 3 == 0, n3
        n2 = 1 << int(np.ceil(np.log2(n3)))
        n_full = db.required_fft_length(6000, reference_length=True)
        a = batch.BatchAligner(n3, 7, max_offset_samples=6000, pairs_in_flight=2).solve(db)
        b = batch.BatchAligner(n2, 7, max_offset_samples=6000, pairs_in_flight=2).solve(db)
        c = batch.BatchAligner(n_full, 7, max_offset_samples=6000, pairs_in_flight=2).solve(db)
        monkeypatch.setenv("FFS_DISABLE_PRUNED_PASS_C", "1")
        e = batch.BatchAligner(n3, 7, max_offset_samples=6000, pairs_in_flight=2).solve(db)
        monkeypatch.delenv("FFS_DISABLE_PRUNED_PASS_C")
        for other in (b, c, e):
            assert np.array_equal(a[0]["offset"], other[0]["offset"]) and np.array_equal(a[0]["score"], other[0]["score"])
            assert np.array_equal(a[1], other[1])
        for p, sp in enumerate(group):
            assert a[1][p]["best_cand"] == sp.true_ratio_index
        # the fp32 chain itself is healthy (exact re-evaluation would hide a damaged transform as long as
        # the true peak is still nominated): fp32 value at the winning lag vs the exact score
        for res in (a, b, c, e):
            assert np.abs(res[0]["score_f32"].astype(np.float64) - res[0]["score"]).max() < 0.5


def test_raw_correlation_f32_single_candidate(torch):
    from ffsubsync_amd import _native

    N = 1 << 16
    rng = np.random.RandomState(3)
    ref = rng.rand(30000).astype(np.float32)
    a = rng.rand(29000).astype(np.float32)
    plan = _native.Plan(N, 1, 2)
    out_a, out_b = plan.correlate_full(_native.FFS_DTYPE_F32, torch.from_numpy(ref).cuda(), (0, 1),
                                       torch.from_numpy(a).cuda(), (0, 1))
    assert out_b is None
    e = _direct_corr(2.0 * ref.astype(float) - 1, 2.0 * a.astype(float) - 1, N)
    assert np.abs(out_a.cpu().numpy() - e).max() < 0.05
    plan.close()


def _check_case(name, c, g, FFTAligner, MaxScoreAligner, FailedToFindAlignmentException):
    for cand, exp in zip(c["cands"], g["per_candidate"]):
        al = FFTAligner(max_offset_samples=c["max_offset"])
        score, offset = al.fit_transform(c["ref"], cand, get_score=True)
        assert offset == exp["offset"], (name, score, offset, exp)
        assert isinstance(offset, int) and al.best_offset_ == offset
        if exp["score"] == "-inf":
            assert score == -np.inf
        else:
            assert score == pytest.approx(float(exp["score"]), rel=SCORE_RTOL, abs=1e-6)
    cands = list(c["cands"])
    (score, offset), winner = MaxScoreAligner(FFTAligner(max_offset_samples=c["max_offset"])).fit_transform(c["ref"], cands)
    bu = g["best_unfiltered"]
    assert winner is cands[bu["index"]] and offset == bu["offset"]
    if "best_filtered" in g:
        bf = g["best_filtered"]
        msa = MaxScoreAligner(FFTAligner, None, 100, c["max_offset"] // 100)
        if "raises" in bf:
            with pytest.raises(FailedToFindAlignmentException, match="Synchronization failed"):
                msa.fit_transform(c["ref"], cands)
        else:
            (score, offset), winner = msa.fit_transform(c["ref"], cands)
            assert winner is cands[bf["index"]] and offset == bf["offset"]
            assert score == pytest.approx(float(bf["score"]), rel=SCORE_RTOL)


@pytest.mark.parametrize("name", sorted(SMALL))
def test_golden_cases_through_the_dropin_classes(torch, name):
    from ffsubsync_amd.aligners import FailedToFindAlignmentException, FFTAligner, MaxScoreAligner

    _check_case(name, SMALL[name], GOLD["cases"][name], FFTAligner, MaxScoreAligner, FailedToFindAlignmentException)


def test_reference_kats_and_call_styles(torch):
    """reference tests/test_alignment.py:7-27, verbatim call styles."""
    from ffsubsync_amd.aligners import FailedToFindAlignmentException, FFTAligner, MaxScoreAligner

    for s1, s2, true_offset in [("111001", "11001", -1), ("1001", "1001", 0), ("10010", "01001", 1)]:
        assert FFTAligner().fit_transform(s2, s1) == true_offset
        assert MaxScoreAligner(FFTAligner).fit_transform(s2, s1)[0][1] == true_offset
        assert MaxScoreAligner(FFTAligner()).fit_transform(s2, s1)[0][1] == true_offset
    for r, s in [(np.array([]), np.array([1, 0, 1])), (np.array([1, 0, 1]), np.array([])), (np.array([]), np.array([]))]:
        with pytest.raises(FailedToFindAlignmentException, match="empty speech data"):
            FFTAligner().fit(r, s)


def test_headline_shape_2h_seven_ratios(torch):
    """BASELINE configs 2/3 shape: 2 h @ 100 Hz, N = 2^21, 7 ratios, max_offset 6000 -- offsets and
    winner bit-identical to the reference, scores within 1e-5."""
    from ffsubsync_amd.aligners import FailedToFindAlignmentException, FFTAligner, MaxScoreAligner

    large = {k: v for k, v in golden_cases.build_cases(include_large=True).items() if k not in SMALL}
    for name, c in large.items():
        _check_case(name, c, GOLD["cases"][name], FFTAligner, MaxScoreAligner, FailedToFindAlignmentException)


def test_scores_are_exact_integers_for_binary_inputs(torch):
    from ffsubsync_amd.aligners import FFTAligner

    c = SMALL["config1_6000"]
    score, offset = FFTAligner(6000).fit_transform(c["ref"], c["cands"][0], get_score=True)
    assert offset == 3720 and float(score) == 50498.0


def test_gss_through_the_dropin(torch):
    from ffsubsync_amd.aligners import FFTAligner, MaxScoreAligner

    ref, sub = golden_cases.gss_case()
    ratios = []

    def maker(r):
        ratios.append(r)
        return golden_cases.ScaledPipe(sub, r)

    msa = MaxScoreAligner(FFTAligner(max_offset_samples=6000))
    msa.fit(ref, [maker])
    (score, offset), pipe = msa.transform()
    g = GOLD["gss"]
    assert [repr(r) for r in ratios] == g["ratios"]
    assert repr(pipe.ratio) == g["final_ratio"] and offset == g["offset"]
    assert score == pytest.approx(float(g["score"]), rel=SCORE_RTOL)


def test_float_inputs_with_window_and_three_times_power_of_two_length(torch):
    """Non-two-level float vectors (fp64 exact re-evaluation) through the drop-in classes, with lag
    windows that select transform lengths 3*2^k, candidates longer and shorter than the reference."""
    from ffsubsync_amd import _native
    from ffsubsync_amd.aligners import FFTAligner, MaxScoreAligner

    rng = np.random.RandomState(21)
    seen = set()
    for R, S, off, mo in ((9000, 14000, 310, 700), (30000, 21000, -1200, 2500), (80000, 90000, 3000, 5000)):
        ref = np.repeat(rng.rand(R // 20 + 1) < 0.4, 20)[:R] * rng.choice([0.25, 0.5, 1.0], size=R)
        idx = np.arange(S) - off  # sub[i] = ref[i + off]  ->  best offset = +off ... as the reference defines it
        sub = np.where((idx + 2 * off >= 0) & (idx + 2 * off < R), ref[np.clip(idx + 2 * off, 0, R - 1)], 0.0)
        sub = sub * 0.9 + 0.05 * (rng.rand(S) < 0.1)
        seen.add(_native.plan_length(R, S, mo))
        got_s, got_o = FFTAligner(max_offset_samples=mo).fit_transform(ref, sub, get_score=True)
        exp_s, exp_o = orc.fft_align(ref, sub, mo)
        assert got_o == exp_o and got_s == pytest.approx(exp_s, rel=SCORE_RTOL), (R, S, mo)
        sub_b = np.roll(sub, 7)
        (s, o), win = MaxScoreAligner(FFTAligner, None, 100, mo / 100.0).fit_transform(ref, [sub_b, sub])
        (es, eo), ei = orc.max_score_align(ref, [sub_b, sub], mo)
        assert (o, win is [sub_b, sub][ei]) == (eo, True) and s == pytest.approx(es, rel=SCORE_RTOL)
    assert any(n % 3 == 0 for n in seen), seen


def test_batch_api_against_oracle(torch):
    """Throughput path: several 15-minute problems in one ffs_align_batch call, generated on the
    GPU, checked pair by pair against the CPU oracle on the same vectors."""
    from ffsubsync_amd import batch
    from workloads import synth

    specs = [synth.make_pair_spec(100 + i, duration_s=900.0) for i in range(5)]
    db8 = synth.build_device_batch(specs, packed=False)
    db = db8.to_bits()
    host = db8.data.cpu().numpy()
    bits = np.unpackbits(db.data.cpu().numpy(), bitorder="little")
    for p, sp in enumerate(specs):  # GPU rasteriser == numpy rasteriser, bytes and bit-packed
        ref, cands = synth.pair_arrays(sp)
        assert np.array_equal(host[db8.offs[p, 0]: db8.offs[p, 0] + db8.lens[p, 0]], ref)
        assert np.array_equal(host[db8.offs[p, 3]: db8.offs[p, 3] + db8.lens[p, 3]], cands[2])
        assert np.array_equal(bits[8 * db.offs[p, 0]: 8 * db.offs[p, 0] + db.lens[p, 0]], ref)
        assert np.array_equal(bits[8 * db.offs[p, 3]: 8 * db.offs[p, 3] + db.lens[p, 3]], cands[2])
    assert db.required_fft_length(6000) <= db.required_fft_length(None) <= db.required_fft_length(None, reference_length=True)
    al = batch.BatchAligner(db.required_fft_length(6000), 7, max_offset_samples=6000, pairs_in_flight=2)
    cres, pres = al.solve(db)
    for p, sp in enumerate(specs):
        fref, fc = synth.pair_float_arrays(sp)
        for j in range(7):
            conv, S = orc.convolve_full(fref, fc[j])
            m = orc.mask_extreme_offsets(conv, S, 6000)
            k = int(np.argmax(m))
            s, o = m[k], len(m) - 1 - k - S
            top2 = np.partition(m, -2)[-2:]
            got_o = int(cres[p, j]["offset"])
            assert cres[p, j]["score"] == pytest.approx(s, rel=SCORE_RTOL)
            if top2[1] - top2[0] > 0.5:
                assert got_o == o  # unique maximum: bit-identical offset
            else:
                # plateau of (near-)exact ties, typical of wrong-ratio candidates: the reference's pick
                # is decided by fp64 rounding noise; ours must be one of the maximal lags
                assert m[len(m) - 1 - got_o - S] >= s - 1e-6 * abs(s)
        (s, o), idx = orc.max_score_align(fref, fc, 6000)
        assert (pres[p]["best_cand"], pres[p]["offset"]) == (idx, o)
        assert idx == sp.true_ratio_index
    # idempotence + sub-range solve give the same records
    cres2, pres2 = al.solve(db, 1, 4)
    assert np.array_equal(pres2, pres[1:4]) and np.array_equal(cres2, cres[1:4])


def test_vad_energy_and_bounds(torch):
    from ffsubsync_amd import _native
    from ffsubsync_amd.speech_transformers import (ComputeSpeechFrameBoundariesMixin, PCMSpeechTransformer,
                                                    _make_energy_detector)

    pcm, state = vo.synth_pcm(480 * 20000 + 123, seed=5)
    # boundary frames: exactly at threshold, one below, silent, full-scale
    pcm[:480] = 0
    pcm[480:960] = 0
    pcm[480:780] = 400
    pcm[960:1440] = 0
    pcm[960:1260] = 400
    pcm[960] = 399
    pcm[1440:1920] = -32768
    exp = vo.detect_fast(pcm, non_speech_label=0.0)
    det = _make_energy_detector(100, 48000, 0.0)
    got = det(pcm.tobytes())
    assert got.dtype == np.float64 and np.array_equal(got, exp)
    assert list(got[:4]) == [0.0, 1.0, 0.0, 1.0]
    got2 = _make_energy_detector(100, 48000, -1.0)(np.frombuffer(pcm.tobytes(), np.uint8))
    assert np.array_equal(got2, vo.detect_fast(pcm, non_speech_label=-1.0))
    # unaligned start / odd frame length take the scalar path
    lab = _native.vad_energy(torch.from_numpy(pcm).cuda()[3:], 441, 50.0, 0.0).cpu().numpy()
    assert np.array_equal(lab, vo.detect_fast(pcm[3:], 100, 44100))
    # chunk loop == oracle chunk loop
    t = PCMSpeechTransformer("energy", 100, 48000, 0.0).fit(pcm)
    assert np.array_equal(t.transform(), vo.chunked_detect(pcm))
    m = ComputeSpeechFrameBoundariesMixin().fit_boundaries(exp)
    assert (m.start_frame_, m.end_frame_) == orc.speech_boundaries(exp) and m.num_frames == m.end_frame_ - m.start_frame_
    z = ComputeSpeechFrameBoundariesMixin().fit_boundaries(np.zeros(1000))
    assert z.start_frame_ is None and z.num_frames is None
    with pytest.raises(ValueError, match="Unable to detect speech"):
        PCMSpeechTransformer().fit(b"")


def test_exact_tie_rule_is_first_maximum_in_k(torch):
    """Where the exact integer correlation has tied maxima the reference's answer depends on fp64
    rounding noise; the device path is deterministic: np.argmax's rule (first k = largest offset)
    applied to the exact values.  Checked on both the direct and the FFT path."""
    from workloads import synth
    from ffsubsync_amd.aligners import FFTAligner

    for n_ref, n_sub, seed in [(700, 300, 4), (700, 300, 8), (700, 300, 11), (3000, 2500, 4), (5000, 4000, 21)]:
        ref, sub = synth.simple_pair(n_ref, n_sub, 123, seed=seed, flip=0.02)
        for mo in (None, 150):
            conv, S = orc.convolve_full(ref, sub)
            exact = np.rint(orc.mask_extreme_offsets(conv, S, mo))  # integers for 0/1 inputs
            k = int(np.argmax(exact))
            score, offset = FFTAligner(mo).fit_transform(ref, sub, get_score=True)
            assert (offset, float(score)) == (len(exact) - 1 - k - S, float(exact[k]))


def test_pruned_last_pass_equals_full_last_pass(torch, monkeypatch):
    """With a lag window the last pass only evaluates the output bins the window can reach; the
    result records must be identical to those of the full column transform."""
    from ffsubsync_amd import batch
    from workloads import synth

    specs = [synth.make_pair_spec(200 + i, duration_s=1200.0) for i in range(3)]
    db = synth.build_device_batch(specs)
    n_fft = db.required_fft_length(6000)
    pruned = batch.BatchAligner(n_fft, 7, max_offset_samples=6000, pairs_in_flight=2).solve(db)
    monkeypatch.setenv("FFS_DISABLE_PRUNED_PASS_C", "1")
    full = batch.BatchAligner(n_fft, 7, max_offset_samples=6000, pairs_in_flight=2).solve(db)
    assert np.array_equal(pruned[0]["offset"], full[0]["offset"]) and np.array_equal(pruned[0]["score"], full[0]["score"])
    assert np.array_equal(pruned[1], full[1])
    assert np.abs(pruned[0]["score_f32"] - full[0]["score_f32"]).max() < 0.05
    for p, sp in enumerate(specs):
        assert pruned[1][p]["best_cand"] == sp.true_ratio_index


def test_block_segmented_mode_equals_single_transform(torch, monkeypatch):
    """With a narrow lag window a 3*2^k plan cuts every candidate into three blocks, correlates each
    with its stretch of the reference by a transform of a third of the length and adds the spectrum
    products before the way back; the records must equal those of the one-transform pipeline."""
    from ffsubsync_amd import batch
    from workloads import synth

    specs = [synth.make_pair_spec(900 + i, duration_s=d) for i, d in enumerate((7200.0, 7100.0, 6500.0, 3500.0, 1700.0))]
    for group, n_cand_used in ((specs[:3], 7), (specs[3:4], 7), (specs[4:], 7)):
        db = synth.build_device_batch(group)
        n_fft = db.required_fft_length(6000)
        assert n_fft % 3 == 0 and n_fft // 3 >= 65536, n_fft
        a = batch.BatchAligner(n_fft, n_cand_used, max_offset_samples=6000, pairs_in_flight=2).solve(db)
        monkeypatch.setenv("FFS_DISABLE_SEGMENTED", "1")
        b = batch.BatchAligner(n_fft, n_cand_used, max_offset_samples=6000, pairs_in_flight=2).solve(db)
        monkeypatch.delenv("FFS_DISABLE_SEGMENTED")
        assert np.array_equal(a[0]["offset"], b[0]["offset"]) and np.array_equal(a[0]["score"], b[0]["score"])
        assert np.array_equal(a[1], b[1])
        assert np.abs(a[0]["score_f32"].astype(np.float64) - a[0]["score"]).max() < 0.5
        for p, sp in enumerate(group):
            assert a[1][p]["best_cand"] == sp.true_ratio_index
    # single candidate, asymmetric window, float inputs (generic load path with a leading-zero reference block)
    from ffsubsync_amd.aligners import FFTAligner
    from ffsubsync_amd._native import plan_length as _native_plan_length

    rng = np.random.RandomState(33)
    ref = np.repeat(rng.rand(15000) < 0.4, 20)[:300000] * rng.choice([0.5, 1.0], size=300000)
    sub = np.concatenate([np.zeros(4321), ref[:290000]]) * 0.9
    assert _native_plan_length(300000, 294321, 4500) % 3 == 0
    for mo in (6000, 4500):
        got = FFTAligner(max_offset_samples=mo).fit_transform(ref, sub, get_score=True)
        exp = orc.fft_align(ref, sub, mo)
        assert got[1] == exp[1] == -4321 and got[0] == pytest.approx(exp[0], rel=SCORE_RTOL)


def test_window_shortened_transform_equals_full_length(torch):
    """With a lag window the plan may use a transform shorter than the reference's N (no aliasing
    reaches the windowed lags, ffs_plan_length): every result record must equal the full-length one."""
    from ffsubsync_amd import _native, batch
    from workloads import synth
    from ffsubsync_amd.aligners import _Vec, solve_pairs

    assert _native.plan_length(720000, 750751, 6000) == 3 << 18 and _native.fft_length(720000, 750751) == 1 << 21
    for name in ("config1_6000", "pipeline_10min", "mask100", "mask_negative_index", "sparse2"):
        c = SMALL[name]
        pair = [(_Vec(c["ref"]), [_Vec(s) for s in c["cands"]])]
        short = solve_pairs(pair, c["max_offset"], c["max_offset"])
        full = solve_pairs(pair, c["max_offset"], c["max_offset"], full_length=True)
        assert np.array_equal(short[0]["offset"], full[0]["offset"]) and np.array_equal(short[0]["score"], full[0]["score"])
        assert np.array_equal(short[1], full[1])
    specs = [synth.make_pair_spec(300 + i, duration_s=2400.0) for i in range(3)]
    db = synth.build_device_batch(specs)
    n_short, n_full = db.required_fft_length(6000), db.required_fft_length(6000, reference_length=True)
    assert n_short < n_full
    a = batch.BatchAligner(n_short, 7, max_offset_samples=6000, pairs_in_flight=2).solve(db)
    b = batch.BatchAligner(n_full, 7, max_offset_samples=6000, pairs_in_flight=2).solve(db)
    assert np.array_equal(a[0]["offset"], b[0]["offset"]) and np.array_equal(a[0]["score"], b[0]["score"])
    assert np.array_equal(a[1], b[1])


def test_flat_topped_peak_uses_exhaustive_fallback(torch):
    """A silent reference makes the correlation exactly flat over thousands of lags: far more exact
    ties than the per-block / per-candidate nominee lists hold.  The overflow pool must then
    re-evaluate every tied lag and apply np.argmax's rule (first k = largest offset) -- no
    FFS_FLAG_AMBIGUOUS left -- on both the full and the pruned last pass, for bytes and floats."""
    from ffsubsync_amd import _native
    from ffsubsync_amd.aligners import _Vec, solve_pairs

    rng = np.random.RandomState(5)
    ref = np.zeros(30000)
    # every sample < 0.5, so every product with the silent reference is positive: the score grows
    # with the overlap and is exactly flat where the overlap is complete, d in [0, R-S]
    sub = 0.4 * (rng.rand(20000) < 0.3)
    sub_f = rng.choice([0.0, 0.125, 0.25, 0.375], size=20000)  # four levels -> float path
    for s in (sub, sub_f):
        for mo in (None, 6000):
            conv, S = orc.convolve_full(ref, s)
            m = orc.mask_extreme_offsets(conv, S, mo)
            top = m.max()
            k = int(np.argmax(m >= top - 1e-6))  # first index of the (exactly tied) plateau
            assert (m >= top - 1e-6).sum() > 5000
            cres, pres = solve_pairs([(_Vec(ref), [_Vec(s)])], mo, mo)
            assert int(cres[0, 0]["offset"]) == len(m) - 1 - k - S
            assert cres[0, 0]["score"] == pytest.approx(top, rel=1e-9)
            assert not (int(cres[0, 0]["flags"]) & _native.FLAG_AMBIGUOUS)
            assert int(pres[0]["best_cand"]) == 0


def test_maximum_length_solve(torch):
    """Largest supported transform (N = 2^24, 46 h of 100 Hz frames): a windowed and a window-less
    solve through the batch entry point, checked against an exact evaluation of the winning lag."""
    from workloads import synth
    from ffsubsync_amd.aligners import FFTAligner

    ref, sub = synth.simple_pair(9_000_000, 7_500_000, -4321, seed=9)
    for mo in (6000, None):
        score, offset = FFTAligner(mo).fit_transform(ref, sub, get_score=True)
        assert offset == -4321
        i = np.arange(sub.size)
        ok = (i + offset >= 0) & (i + offset < ref.size)
        exact = float(np.sum((2.0 * sub[ok] - 1) * (2.0 * ref[i[ok] + offset] - 1)))
        assert float(score) == exact


def test_vad_token_smoothing_matches_restatement(torch):
    """ffs_vad_tokenize (auditok-style smoothing, parity unpinned) against the Python restatement:
    random validity patterns exercising min/max length, tolerated silence, truncation, several labels,
    chunk boundaries; then the whole auditok-like detector on PCM."""
    from ffsubsync_amd import _native
    from ffsubsync_amd.speech_transformers import PCMSpeechTransformer, _make_auditok_detector

    rng = np.random.RandomState(3)
    for trial in range(12):
        n = int(rng.choice([1, 7, 500, 10000, 23456]))
        p_on = rng.choice([0.02, 0.2, 0.6, 0.95])
        runs = rng.geometric(1.0 / rng.choice([3, 15, 80, 700]), size=n // 2 + 4)
        valid = np.repeat(rng.rand(runs.size) < p_on, runs)[:n]
        if valid.size < n:
            valid = np.concatenate([valid, np.zeros(n - valid.size, bool)])
        for label in (0.0, 0.25, -1.0):
            for chunk in (10000, 997):
                got = _native.vad_tokenize(torch.from_numpy(valid.astype(np.float32)).cuda(), chunk, 20.0, 500, 25.0, label)
                want = vo.tokenize(valid, label, chunk)
                assert np.array_equal(got.cpu().numpy().astype(float), want), (trial, n, label, chunk)
    pcm, _ = vo.synth_pcm(480 * 25000 + 77, seed=8)
    want = np.concatenate([vo.tokenize_chunk(vo.detect_fast(pcm[o:o + 4800000]) > 0.5, 0.0)
                           for o in range(0, pcm.size, 4800000)])
    assert np.array_equal(_make_auditok_detector(100, 48000, 0.0)(pcm[:4800000].tobytes()), want[:10000])
    assert np.array_equal(PCMSpeechTransformer("auditok", 100, 48000, 0.0).fit(pcm).transform(), want)


def test_aligner_is_usable_from_a_thread_pool(torch):
    """The reference drives its transformers from a 4-thread pool (speech_transformers.py:872-873);
    every thread gets its own plan, so concurrent solves must not disturb each other."""
    from concurrent.futures import ThreadPoolExecutor

    from workloads import synth
    from ffsubsync_amd.aligners import FFTAligner

    jobs = []
    for i in range(8):
        off = 100 * i - 350
        ref, sub = synth.simple_pair(40000, 36000, off, seed=30 + i)
        jobs.append((ref, sub, off))

    def solve(job):
        ref, sub, off = job
        out = [FFTAligner(6000).fit_transform(ref, sub) for _ in range(5)]
        return out, off

    with ThreadPoolExecutor(max_workers=4) as ex:
        for out, off in ex.map(solve, jobs):
            assert out == [off] * 5


def test_extreme_densities_stay_exact(torch):
    """Activity densities of 2 % / 98 % give the correlation a huge DC term and the fp32 pipeline its
    largest error (~0.3 at N = 2^21); the margin widens with the maximum so the exact winner is still
    nominated.  Offsets and scores must equal an exact evaluation."""
    from ffsubsync_amd.aligners import FFTAligner

    for dens, seed in [(0.02, 1), (0.98, 2), (0.03, 3)]:
        rng = np.random.RandomState(seed)
        ref = (rng.rand(720000) < dens).astype(np.uint8)
        sub = np.zeros(700000, np.uint8)
        off = 1234 if dens < 0.5 else -777
        idx = np.arange(sub.size) + off
        ok = (idx >= 0) & (idx < ref.size)
        sub[ok] = ref[idx[ok]]
        flip = rng.rand(sub.size) < 0.002
        sub = np.where(flip, 1 - sub, sub).astype(np.uint8)
        for mo in (6000, None):
            score, offset = FFTAligner(mo).fit_transform(ref, sub, get_score=True)
            s_o, o_o = orc.fft_align(ref, sub, mo)
            assert offset == o_o == off
            assert float(score) == float(np.rint(s_o))


def test_randomised_problems_against_oracle(torch):
    """Fuzz over lengths (direct kernel, every FFT split up to 2^17), densities, sample levels, lag
    windows (including ones wider than the data and Python-negative-slice ones) and candidate counts:
    offsets must equal the oracle's whenever its top-2 gap exceeds 0.5, scores always match."""
    from ffsubsync_amd.aligners import _Vec, solve_pairs

    trials = int(os.environ.get("FFS_FUZZ_TRIALS", "140"))  # raise for a soak run
    big = trials > 140
    rng = np.random.RandomState(2024 if not big else int(os.environ.get("FFS_FUZZ_SEED", "7")))
    checked = unique = 0
    for trial in range(trials):
        R = int(rng.choice([rng.randint(40, 400), rng.randint(400, 6000), rng.randint(6000, 70000)] +
                           ([rng.randint(70000, 500000)] if big else [])))
        S = int(max(20, R * rng.uniform(0.3, 1.6)))
        dens = rng.choice([0.05, 0.3, 0.5, 0.9])
        seg = np.maximum(1, rng.geometric(1.0 / rng.choice([2, 20, 200]), size=R))
        ref01 = np.repeat(rng.rand(seg.size) < dens, seg)[:R]
        n_cand = int(rng.choice([1, 1, 2, 3, 7] + ([5, 8] if big else [])))
        lo_hi_ref = [(0.0, 1.0), (-1.0, 1.0), (0.25, 1.0)][rng.randint(3)]
        ref = np.where(ref01, lo_hi_ref[1], lo_hi_ref[0])
        cands = []
        for j in range(n_cand):
            off = int(rng.randint(-S // 2, R // 2 + 1))
            idx = np.arange(S) + off
            ok = (idx >= 0) & (idx < R)
            c01 = np.zeros(S, bool)
            c01[ok] = ref01[idx[ok]]
            c01 ^= rng.rand(S) < rng.choice([0.0, 0.02, 0.3])
            amp = [1.0, 0.96, 0.999][rng.randint(3)]
            cands.append(c01 * amp)
        mo = [None, None, 6000, int(rng.randint(0, 3 * R)), int(rng.randint(1, 200))][rng.randint(5)]
        cres, pres = solve_pairs([(_Vec(ref), [_Vec(c) for c in cands])], mo, mo)
        best = None
        for j, c in enumerate(cands):
            conv, S_ = orc.convolve_full(ref, c)
            m = orc.mask_extreme_offsets(conv, S_, mo)
            k = int(np.argmax(m))
            s_o, o_o = m[k], len(m) - 1 - k - S_
            got_s, got_o = float(cres[0, j]["score"]), int(cres[0, j]["offset"])
            checked += 1
            if not np.isfinite(s_o):
                assert got_s == -np.inf and got_o == o_o, (trial, j)
                continue
            assert got_s == pytest.approx(s_o, rel=1e-9, abs=1e-6), (trial, j, R, S, mo)
            fin = np.sort(m[np.isfinite(m)])
            if fin.size < 2 or fin[-1] - fin[-2] > 0.5:
                unique += 1
                assert got_o == o_o, (trial, j, R, S, mo, got_o, o_o)
            else:
                assert m[len(m) - 1 - got_o - S_] >= s_o - 1e-6 * max(1.0, abs(s_o)), (trial, j)
        assert not (cres["flags"] & 2).any()
        # MaxScoreAligner.transform: drop |offset| > max, first maximum wins (aligners.py:154-167)
        kept = [(float(cres[0, j]["score"]), j) for j in range(n_cand)
                if mo is None or abs(int(cres[0, j]["offset"])) <= mo]
        if not kept:
            assert int(pres[0]["best_cand"]) == -1
        else:
            top = max(s for s, _ in kept)
            assert int(pres[0]["best_cand"]) == next(j for s, j in kept if s == top)
            assert float(pres[0]["score"]) == top
    assert unique > checked // 2
