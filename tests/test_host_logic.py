"""Host-side logic that needs no GPU: sklearn shim, golden-section search, generators, sharding and
the MaxScoreAligner control flow (driven with a CPU stand-in for the base aligner)."""
import json
import os

import numpy as np
import pytest

import golden_cases
from workloads import synth
from ffsubsync_amd.aligners import FailedToFindAlignmentException, MaxScoreAligner
from ffsubsync_amd.batch import shard_bounds
from ffsubsync_amd.golden_section_search import gss
from ffsubsync_amd.sklearn_shim import Pipeline, TransformerMixin, make_pipeline
from oracle import aligners_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "aligner_golden.json")))


class OracleBackedAligner(TransformerMixin):
    """Duck-typed base aligner (no _solve_many): exercises MaxScoreAligner's generic per-candidate
    path, the one a foreign aligner object would take."""

    def __init__(self, max_offset_samples=None):
        self.max_offset_samples = max_offset_samples

    def fit(self, ref, sub, get_score=False):
        self.res = orc.fft_align(ref, sub, self.max_offset_samples)
        self.get_score_ = get_score
        return self

    def transform(self, *_):
        return self.res if self.get_score_ else self.res[1]


class Add(TransformerMixin):
    def __init__(self, k):
        self.k = k

    def fit(self, X, *_):
        self.seen_ = X
        return self

    def transform(self, X):
        return X + self.k


def test_pipeline_contract():
    pipe = Pipeline([("a", Add(1)), ("b", Add(10))])
    assert pipe.fit_transform(5) == 16
    assert pipe.fit(5) is pipe and pipe.transform(5) == 16  # transform is a property returning a callable
    assert pipe.named_steps["b"].k == 10 and pipe[-1].k == 10 and pipe["a"].k == 1 and len(pipe[:1]) == 1
    mp = make_pipeline(Add(1), Add(2))
    assert [n for n, _ in mp.steps] == ["add-1", "add-2"]
    with pytest.raises(ValueError):
        Pipeline([("a", Add(1)), ("a", Add(2))])


def test_gss_matches_reference_trace():
    ref, sub = golden_cases.gss_case()
    seen = []

    def objective(ratio, last):
        seen.append((ratio, last))
        return -orc.fft_align(ref, golden_cases.scaled(sub, ratio), 6000)[0]

    lo, hi = gss(objective, 0.9, 1.1)
    assert [repr(x) for x, _ in seen] == GOLD["gss"]["ratios"]
    assert [last for _, last in seen].count(True) == 1 and seen[-1][1]
    assert hi - lo <= 1.1e-4
    a, b = gss(lambda x, last: (x - 2) ** 2, 1, 5, 1e-5)
    assert [repr(a), repr(b)] == GOLD["gss_doc_example"]
    with pytest.raises(TypeError):  # the reference calls f(d, flag) directly on one branch (:69)
        gss(lambda x: -x, 1, 5, 1e-5)


def test_max_score_aligner_control_flow():
    c = golden_cases.build_cases(include_large=False)["sparse2"]
    g = GOLD["cases"]["sparse2"]
    cands = list(c["cands"])
    msa = MaxScoreAligner(OracleBackedAligner, None, 100, 60)
    assert msa.max_offset_samples == 6000 and msa.base_aligner.max_offset_samples == 6000
    (score, offset), winner = msa.fit_transform(c["ref"], cands)
    assert winner is cands[g["best_filtered"]["index"]] and offset == g["best_filtered"]["offset"]
    assert len(msa._scores) == 7
    msa.fit(c["ref"], cands[:2])  # _scores is append-only across fits (aligners.py:109, 144)
    assert len(msa._scores) == 9
    # an instance keeps its own window and disables filtering (aligners.py:104-108)
    inst = MaxScoreAligner(OracleBackedAligner(max_offset_samples=10))
    assert inst.max_offset_samples is None
    # nothing within the limit -> the reference's error message
    c0 = golden_cases.build_cases(include_large=False)["mask_all"]
    with pytest.raises(FailedToFindAlignmentException, match="Synchronization failed"):
        MaxScoreAligner(OracleBackedAligner, None, 100, 0).fit_transform(c0["ref"], list(c0["cands"]))


def test_max_score_aligner_gss_records_only_last():
    ref, sub = golden_cases.gss_case()
    msa = MaxScoreAligner(OracleBackedAligner(max_offset_samples=6000))
    ratios = []

    def maker(r):
        ratios.append(r)
        return golden_cases.ScaledPipe(sub, r)

    msa.fit(ref, [maker])
    (score, offset), pipe = msa.transform()
    assert len(msa._scores) == 1 and len(ratios) == len(GOLD["gss"]["ratios"])
    assert repr(pipe.ratio) == GOLD["gss"]["final_ratio"] and offset == GOLD["gss"]["offset"]


def test_synth_generator_properties():
    spec = synth.make_pair_spec(3, duration_s=900.0)
    ref, cands = synth.pair_arrays(spec)
    assert ref.dtype == np.uint8 and ref.size == 90000 and 0.2 < ref.mean() < 0.6
    assert len(cands) == 7 and all(c.size == n for c, n in zip(cands, spec.cand_len))
    assert spec.cand_amp[0] == 1.0 and spec.cand_amp[1] == pytest.approx(23.976 / 24.0)
    fref, fc = synth.pair_float_arrays(spec)
    (score, offset), idx = orc.max_score_align(fref, fc, 6000)
    assert idx == spec.true_ratio_index and abs(offset - spec.true_offset_samples) <= 15
    # uniqueness of the winner: oracle top-2 gap > 0.5
    conv, S = orc.convolve_full(fref, fc[idx])
    m = orc.mask_extreme_offsets(conv, S, 6000)
    top2 = np.sort(m[np.isfinite(m)])[-2:]
    assert top2[1] - top2[0] > 0.5
    assert synth.make_pair_spec(3, duration_s=900.0).ref_starts.tolist() == spec.ref_starts.tolist()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 1024, 1030):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) <= (n + w - 1) // w


def test_product_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ffsubsync_amd.aligners import FFTAligner

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FFTAligner().fit("1001", "1001")


def test_npz_speech_round_trip(tmp_path):
    """--serialize-speech / DeserializeSpeechTransformer (ffsubsync.py:639-644, speech_transformers.py:987-1009)."""
    from ffsubsync_amd.speech_transformers import DeserializeSpeechTransformer, serialize_speech

    speech = np.array([0.0, 1.0, 0.6, 1.0, 0.0, 0.4])
    serialize_speech(str(tmp_path / "ref.npz"), speech)
    got = DeserializeSpeechTransformer(non_speech_label=-0.5).fit(str(tmp_path / "ref.npz")).transform()
    assert got.tolist() == [-0.5, 1.0, -0.5, 1.0, -0.5, -0.5]
    np.save(str(tmp_path / "ref.npy"), speech)
    assert DeserializeSpeechTransformer(0.0).fit(str(tmp_path / "ref.npy")).transform().tolist() == [0, 1, 0, 1, 0, 0]
    np.savez(str(tmp_path / "bad.npz"), other=speech)
    with pytest.raises(ValueError, match='could not find "speech" array'):
        DeserializeSpeechTransformer(0.0).fit(str(tmp_path / "bad.npz"))


def test_install_swaps_the_classes_the_caller_binds(monkeypatch):
    """ffsubsync/ffsubsync.py:14 binds FFTAligner / MaxScoreAligner by name from ffsubsync.aligners;
    install() must replace them in both modules (the seam tests/test_quality_gate.py:98-102 patches)
    and adopt the reference's exception class so the caller's `except` clauses keep working."""
    import sys
    import types

    import ffsubsync_amd
    from ffsubsync_amd import aligners as amd_aligners

    class RefFailed(Exception):
        pass

    pkg = types.ModuleType("ffsubsync")
    pkg.__path__ = []
    ref_al = types.ModuleType("ffsubsync.aligners")
    ref_al.FFTAligner, ref_al.MaxScoreAligner, ref_al.FailedToFindAlignmentException = object, object, RefFailed
    ref_main = types.ModuleType("ffsubsync.ffsubsync")
    ref_main.FFTAligner, ref_main.MaxScoreAligner = object, object
    pkg.aligners, pkg.ffsubsync = ref_al, ref_main
    for name, mod in (("ffsubsync", pkg), ("ffsubsync.aligners", ref_al), ("ffsubsync.ffsubsync", ref_main)):
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setattr(amd_aligners, "FailedToFindAlignmentException", amd_aligners.FailedToFindAlignmentException)
    monkeypatch.setattr(ffsubsync_amd, "FailedToFindAlignmentException", ffsubsync_amd.FailedToFindAlignmentException)
    ffsubsync_amd.install()
    for mod in (ref_al, ref_main):
        assert mod.FFTAligner is amd_aligners.FFTAligner and mod.MaxScoreAligner is amd_aligners.MaxScoreAligner
    assert amd_aligners.FailedToFindAlignmentException is RefFailed
    # the drop-in now raises the reference's exception type
    c0 = golden_cases.build_cases(include_large=False)["mask_all"]
    with pytest.raises(RefFailed, match="Synchronization failed"):
        MaxScoreAligner(OracleBackedAligner, None, 100, 0).fit_transform(c0["ref"], list(c0["cands"]))


def test_plan_length_is_alias_free_for_the_window():
    """ffs_plan_length's claim, checked in numpy: a circular correlation of that length over the
    prefixes that can reach the lag window reproduces the reference's masked `convolve` values at
    every lag of the window (aligners.py:31-43, 67-74)."""
    from ffsubsync_amd import _native
    from oracle import aligners_oracle as orc

    rng = np.random.RandomState(5)
    seen_r3 = seen_short = 0
    for trial in range(60):
        R = int(rng.randint(30, 9000))
        S = int(max(10, R * rng.uniform(0.3, 1.7)))
        mo = [0, 7, 100, 600, 6000, 3 * R, None][rng.randint(7)]
        ref = (rng.rand(R) < 0.4).astype(float)
        sub = (rng.rand(S) < 0.4).astype(float) * 0.97
        conv, _ = orc.convolve_full(ref, sub)
        masked = orc.mask_extreme_offsets(conv, S, mo)
        n_ref = len(conv)
        n = _native.plan_length(R, S, mo)
        ks = np.flatnonzero(np.isfinite(masked))
        # lags whose overlap is empty are exactly 0 and never reach the transforms (the zero rule,
        # CAND_HAS_ZERO in ffs_kernels.h): the plan only has to carry d in (-S, R)
        d_all = n_ref - 1 - S - ks
        assert np.abs(masked[ks[(d_all >= R) | (d_all <= -S)]]).max(initial=0.0) < 1e-6
        ks = ks[(d_all < R) & (d_all > -S)]
        if ks.size == 0:
            assert n == 2
            continue
        seen_r3 += n % 3 == 0
        seen_short += n < n_ref
        d = n_ref - 1 - S - ks  # lag of every surviving k
        d_lo, d_hi = int(d.min()), int(d.max())
        s_eff = max(1, min(S, R - d_lo))  # samples that meet the other vector at some lag of the window
        r_eff = max(1, min(R, S + d_hi))
        assert n >= max(s_eff + d_hi, r_eff - d_lo)
        a = np.zeros(n)
        a[:s_eff] = 2.0 * sub[:s_eff] - 1.0
        b = np.zeros(n)
        b[:r_eff] = 2.0 * ref[:r_eff] - 1.0
        circ = np.real(np.fft.ifft(np.conj(np.fft.fft(a)) * np.fft.fft(b)))  # circ[m] = sum_i a[i] b[(i+m) % n]
        got = circ[np.where(d >= 0, d, d + n)]
        assert np.abs(got - masked[ks]).max() < 1e-6 * max(1.0, np.abs(masked[ks]).max()), (trial, R, S, mo, n)
    assert seen_short > 10 and seen_r3 > 0


def test_select_candidates_view():
    from ffsubsync_amd.batch import DeviceBatch

    offs = np.arange(12).reshape(3, 4) * 64
    lens = offs + 5
    lo, hi = np.zeros((3, 4)), np.arange(12).reshape(3, 4) / 10.0
    db = DeviceBatch(None, offs, lens, lo, hi)
    one = db.select_candidates([2, 0, 1])
    assert one.n_pairs == 3 and one.n_cand == 1
    assert one.offs.tolist() == [[0, 192], [256, 320], [512, 640]]
    assert one.hi[:, 1].tolist() == [0.3, 0.5, 1.0] and one.lens[:, 0].tolist() == [5, 261, 517]


def test_bench_starts_itself_under_the_launcher(monkeypatch):
    """`python bench.py --gpus N` without a launcher must re-run itself under torch.distributed.run with one
    process per GPU on 127.0.0.1 (a driver can produce the scaling record with a plain command)."""
    import importlib
    import sys

    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    import torch

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--scaling", "strong"])
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--scaling", "strong"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ


def test_bench_says_so_when_the_gpus_asked_for_are_not_there(monkeypatch, capsys):
    """`--gpus N` with fewer than N visible devices: ONE JSON line with value null and an error text, exit code 2 --
    never a silent measurement of something else."""
    import importlib
    import json
    import sys

    import torch

    bench = importlib.import_module("bench")
    monkeypatch.setattr(bench.subprocess, "call", lambda *a, **k: pytest.fail("must not launch"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 2
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 8 and "8 asked for, 1 HIP device" in line["error"]


def test_strong_and_weak_scaling_shards():
    """bench.py --scaling strong splits the same pairs over the ranks (BASELINE configs[3]); the shards tile
    the batch exactly, in order, for every world size."""
    for n in (1024, 8192, 1000, 7):
        for world in (1, 2, 4, 8):
            bounds = [shard_bounds(n, r, world) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
            per = (n + world - 1) // world
            assert all(hi - lo <= per for lo, hi in bounds)


def test_two_level_pack_equals_the_numpy_detection():
    """ffs_two_level_pack (host-only): the levels and bits the drop-in classes derive from the float64 vectors
    FFTAligner.fit receives -- against the numpy formulation it replaced (min / max / == hi / all(== hi | == lo))."""
    from ffsubsync_amd import _native
    from ffsubsync_amd.aligners import _Vec

    rng = np.random.RandomState(8)
    for trial in range(200):
        n = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 1000, 4097]))
        lo, hi = sorted(rng.choice([-1.0, 0.0, 0.25, 0.96, 1.0, 3.5], size=2, replace=False))
        x = np.where(rng.rand(n) < rng.choice([0.0, 0.1, 0.5, 1.0]), hi, lo).astype(np.float64)
        kind = trial % 5
        if kind == 3 and n > 2:
            x[rng.randint(n)] = 0.5 * (lo + hi)          # a third level
        if kind == 4 and n > 1:
            x[rng.randint(n)] = [np.nan, np.inf, -np.inf][trial % 3]
        got = _native.two_level_pack(x)
        want_two = bool(np.all((x == x.max()) | (x == x.min()))) and np.isfinite(x.min()) and np.isfinite(x.max())
        assert (got is not None) == want_two, (trial, n, kind)
        v = _Vec(x)
        assert v.two_level == want_two and len(v) == n
        if want_two:
            assert (v.lo, v.hi) == (float(x.min()), float(x.max()))
            bits = np.packbits((x == x.max()) & (x.max() != x.min()), bitorder="little")
            assert np.array_equal(v.packed[: bits.size], bits) and not v.packed[bits.size:].any()
            assert v.packed.size == (n + 31) // 32 * 4
        else:
            assert v.packed is None
    # strings and integer lists still come through (aligners.py:51-57)
    v = _Vec("0110")
    assert v.two_level and (v.lo, v.hi) == (0.0, 1.0) and v.packed[0] == 0b0110


def test_algorithm_names_are_validated_and_explicit_choices_stick(monkeypatch):
    """FFS_ALGORITHM is validated (case and blanks ignored, unknown values a clear ValueError instead of a
    ctypes ArgumentError), and `algorithm_code` accepts names and FFS_ALGO_* codes only."""
    from ffsubsync_amd import _native

    assert _native.algorithm_code(" FFT ") == _native.FFS_ALGO_FFT and _native.algorithm_code("runs") == _native.FFS_ALGO_RUNS
    assert _native.algorithm_code(_native.FFS_ALGO_AUTO) == _native.FFS_ALGO_AUTO
    for bad in ("fast", 7, None, 1.5):
        with pytest.raises(ValueError):
            _native.algorithm_code(bad)
    monkeypatch.delenv("FFS_ALGORITHM", raising=False)
    assert _native.env_algorithm() == "auto"
    monkeypatch.setenv("FFS_ALGORITHM", " Runs")
    assert _native.env_algorithm() == "runs"
    monkeypatch.setenv("FFS_ALGORITHM", "quick")
    with pytest.raises(ValueError, match="FFS_ALGORITHM"):
        _native.env_algorithm()


def test_device_copy_cache_key_sees_every_in_place_edit():
    """the cache key of a fitted vector's device copy is a checksum of ALL its bytes: zeroing ten seconds of a
    two-hour vector (which 64 strided probes miss nine times out of ten) changes it; equal content under another object
    identity does not."""
    from ffsubsync_amd import subtitle_raster as sr

    rng = np.random.RandomState(0)
    x = (rng.rand(720_000) < 0.4).astype(np.float64)
    k0 = sr._vector_key(x)
    assert sr._vector_key(x.copy()) == k0
    for start in rng.randint(0, 719_000, 20):
        y = x.copy()
        y[start:start + 1000] = 1.0 - y[start:start + 1000]
        assert sr._vector_key(y) != k0
    assert sr._vector_key(x.astype(np.float32)) != k0 and sr._vector_key(x.reshape(2, -1)) != k0
    assert sr._vector_key(np.zeros(0)) is None and sr._vector_key([1.0, 0.0]) is None


def test_trackset_tables_and_arena_reuse():
    """TrackSet lays the per-track interval arrays out in one table per column (one concatenation each, the largest end of
    every track by one reduceat); with ``arena=`` the tables of an earlier TrackSet are filled again.  Same tables as the
    straightforward per-track construction: empty tracks, tracks without metadata flags, list inputs."""
    import numpy as np

    from ffsubsync_amd.batch import TrackSet

    rng = np.random.RandomState(4)
    tracks = []
    for k in range(40):
        n = int(rng.randint(0, 50)) if k % 7 else 0
        s = np.sort(rng.randint(0, 10 ** 9, n)).astype(np.int64)
        e = s + rng.randint(1, 10 ** 6, n)
        m = None if k % 3 == 0 else (rng.rand(n) < 0.1).astype(np.uint8)
        tracks.append((s.tolist(), e, m) if k == 5 else (s, e, m))

    def check(ts):
        assert np.array_equal(ts.counts, [len(t[0]) for t in tracks])
        assert np.array_equal(ts.firsts, np.concatenate([[0], np.cumsum(ts.counts)[:-1]]))
        assert np.array_equal(ts.start_us, np.concatenate([np.asarray(t[0], dtype=np.int64) for t in tracks]))
        assert np.array_equal(ts.end_us, np.concatenate([np.asarray(t[1], dtype=np.int64) for t in tracks]))
        assert np.array_equal(ts.meta, np.concatenate([np.zeros(len(t[0]), np.uint8) if t[2] is None else t[2] for t in tracks]))
        assert np.array_equal(ts.end_max, [int(np.max(t[1])) if len(t[1]) else 0 for t in tracks])
        assert ts.start_us.dtype == np.int64 and ts.end_us.dtype == np.int64 and ts.meta.dtype == np.uint8

    first = TrackSet(tracks)
    check(first)
    again = TrackSet(tracks, arena=first)
    check(again)
    assert again.start_us.base is first._base[0]  # the same memory
    bigger = TrackSet(tracks + tracks, arena=again)  # does not fit: fresh tables
    assert bigger.start_us.size == 2 * again.start_us.size and bigger._base is not again._base
    assert TrackSet([(np.zeros(0, np.int64), np.zeros(0, np.int64), None)]).meta is None
    assert TrackSet([]).start_us.size == 0


def test_golden_comparison_rule():
    """workloads/golden_check.py, the one rule behind `pairs_matching_reference_golden` and the golden GPU tests: winner
    bit-identical, scores within 1e-5, a candidate's offset bit-identical where the reference's top-2 gap exceeds 0.5 and
    inside the reference's own plateau of near-maximal lags otherwise."""
    import numpy as np

    from workloads import golden_check as gc

    g = {"seed": 3, "index": 1, "offset": 40, "score": "1000.0000000001",
         "per_candidate": [["900.0", -7], ["1000.0000000001", 40]],
         "per_candidate_top2_gap": [0.0, 12.0], "per_candidate_plateau": [[5, -9, -3], [1, 40, 40]]}
    pd = np.dtype([("best_cand", "i4"), ("offset", "i8"), ("score", "f8")])
    cd = np.dtype([("offset", "i8"), ("score", "f8")])

    def recs(winner, cands):
        return np.array(winner, dtype=pd), np.array(cands, dtype=cd)

    p, c = recs((1, 40, 1000.0), [(-3, 900.0), (40, 1000.0)])  # the tie resolved to another lag of the plateau: fine
    assert gc.pair_mismatches(g, p, c) == []
    p, c = recs((1, 40, 1000.0), [(-2, 900.0), (40, 1000.0)])  # outside the plateau
    assert [m[0] for m in gc.pair_mismatches(g, p, c)] == ["offset outside the reference's plateau"]
    p, c = recs((1, 41, 1000.0), [(-7, 900.0), (41, 1000.0)])  # a clear maximum at another lag
    assert sorted(m[0] for m in gc.pair_mismatches(g, p, c)) == ["offset", "winner"]
    p, c = recs((1, 40, 1000.02), [(-7, 900.0), (40, 1000.02)])  # 2e-5 relative
    assert sorted(m[0] for m in gc.pair_mismatches(g, p, c)) == ["score", "winner"]
    assert gc.count_ties(g) == 1
    ok, total, first = gc.matching({3: g}, [2, 3], [None, p], [None, c])
    assert (ok, total) == (0, 1) and len(first) == 1
    wl = gc.load("windowless_golden")
    assert len(wl) >= 64 and all("per_candidate_plateau" in v for v in wl.values())
