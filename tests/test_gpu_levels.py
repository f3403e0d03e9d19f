"""Multi-level float references on the run-boundary path (round 5): the `weighted` fused VAD's four-level vector
(ffsubsync/speech_transformers.py:290-293) as threshold lists with integer multiplicities (csrc/ffs_runs.h, LevelInfo) --
against the unmodified reference's goldens (tests/golden/float_golden.json), the transform path's records, and the
fallbacks: more than four levels, steps that share no small quantum, a level the sampling missed.  Need a real MI355X.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "float_golden.json")))["pairs"]
REL, TIE = 1e-5, 1e-6


def _solve(db, n_fft, max_off, algorithm, n_cand=7, pairs_in_flight=4):
    from ffsubsync_amd import batch

    al = batch.BatchAligner(n_fft, n_cand, max_off, pairs_in_flight=pairs_in_flight, algorithm=algorithm)
    out = al.solve(db)
    stats = al.plan.runs_stats()
    al.close()
    return out, stats


def _check(g, index, offset, score, per):
    assert (index, offset) == (g["index"], g["offset"]), (g["seed"], index, offset)
    assert abs(score - float(g["score"])) <= REL * abs(float(g["score"]))
    for j, (sc, off) in enumerate(g["per_candidate"]):
        assert abs(per[j][0] - float(sc)) <= REL * abs(float(sc)), (g["seed"], j, per[j], sc)
        if g["per_candidate_top2_gap"][j] > TIE:
            assert per[j][1] == off, (g["seed"], j, per[j], off)


def _agree(a, b, rel=1e-9):
    """Offsets identical; scores equal up to the two evaluations' fp64 rounding (integer counts x levels here, a dot
    product over the caller's samples on the transform path)."""
    assert np.array_equal(a[0]["offset"], b[0]["offset"])
    assert np.allclose(a[0]["score"], b[0]["score"], rtol=rel, atol=1e-6)
    assert np.array_equal(a[1]["best_cand"], b[1]["best_cand"]) and np.array_equal(a[1]["offset"], b[1]["offset"])


@pytest.mark.parametrize("max_off", [6000, None])
def test_four_level_references_equal_the_reference_goldens_on_the_run_path(max_off):
    from ffsubsync_amd import _native
    from workloads import synth

    specs = [synth.make_pair_spec(g["seed"]) for g in GOLD]
    db = synth.build_fused_batch(specs)
    assert db.call_dtype == (_native.FFS_DTYPE_F64, _native.FFS_DTYPE_U1)
    if max_off is None:
        db = db.select_candidates([sp.true_ratio_index for sp in specs])
    n_fft = db.required_fft_length(max_off)
    n_cand = db.n_cand
    got, st = _solve(db, n_fft, max_off, "runs" if max_off is None else "auto", n_cand=n_cand)
    assert st[0] == 1 and st[2] == 0, st  # the run-boundary path, no sub-batch through the transforms
    want, st_f = _solve(db, n_fft, max_off, "fft", n_cand=n_cand)
    assert st_f == (0, 0, 0)
    _agree(got, want)
    cres, pres = got
    for i, g in enumerate(GOLD):
        if max_off is None:
            sc, off = float(g["single_none"][0]), g["single_none"][1]
            assert int(cres[i, 0]["offset"]) == off and abs(float(cres[i, 0]["score"]) - sc) <= REL * abs(sc)
        else:
            _check(g, int(pres[i]["best_cand"]), int(pres[i]["offset"]), float(pres[i]["score"]),
                   [(float(c["score"]), int(c["offset"])) for c in cres[i]])


def _batch(refs, cands01, ref_dtype=np.float64, amps=None):
    """Mixed-type DeviceBatch: float references (one per pair), bit-packed candidates."""
    import torch

    from ffsubsync_amd import _native, batch

    n_pairs, n_cand = len(refs), len(cands01[0])
    lens = np.array([[len(r)] + [len(c) for c in cs] for r, cs in zip(refs, cands01)], dtype=np.int64)
    esz = np.dtype(ref_dtype).itemsize
    nbytes = lens.copy()
    nbytes[:, 0] = lens[:, 0] * esz
    nbytes[:, 1:] = (lens[:, 1:] + 31) // 32 * 4
    offs, total = batch._layout(lens, nbytes)
    host = np.zeros(total, dtype=np.uint8)
    for p in range(n_pairs):
        r = np.ascontiguousarray(refs[p], dtype=ref_dtype)
        host[offs[p, 0]: offs[p, 0] + r.nbytes] = r.view(np.uint8)
        for j, c in enumerate(cands01[p]):
            pk = np.packbits(np.concatenate([c, np.zeros(-c.size % 32, np.uint8)]), bitorder="little")
            host[offs[p, 1 + j]: offs[p, 1 + j] + pk.size] = pk
    lo = np.zeros(lens.shape)
    hi = np.ones(lens.shape)
    if amps is not None:
        hi[:, 1:] = amps
    rd = _native.FFS_DTYPE_F64 if ref_dtype == np.float64 else _native.FFS_DTYPE_F32
    return batch.DeviceBatch(torch.from_numpy(host).cuda(), offs, lens, lo, hi, _native.FFS_DTYPE_U1, ref_dtype=rd)


def _runs01(rng, n, mean_run):
    x = np.zeros(n, np.uint8)
    i, v = 0, int(rng.rand() < 0.5)
    while i < n:
        ln = int(rng.randint(max(1, mean_run // 3), mean_run * 2))
        x[i:i + ln] = v
        i += ln
        v ^= 1
    return x


@pytest.mark.parametrize("case", ["weighted_label_0.1", "three_levels_f32", "two_levels_f64", "steps_3_to_2"])
def test_level_sets_the_run_path_accepts(case):
    """Other level sets through the threshold lists: the weighted fusion with a non-zero non_speech_label, three levels in
    float32, a two-level float vector, steps 3 : 2 (quantum = half the smaller step) -- offsets and scores as the transform
    path returns them (whose exact re-evaluation is an fp64 dot product over the same samples) and as the CPU oracle does."""
    from oracle import aligners_oracle as orc

    rng = np.random.RandomState(42)
    n_pairs, n_cand, n = 6, 3, 90_000
    refs, cands = [], []
    for p in range(n_pairs):
        a, b = _runs01(rng, n, 300), _runs01(rng, n, 260)
        if case == "weighted_label_0.1":
            l = 0.1
            r = 0.6 * np.where(a, 1.0, l) + 0.4 * np.where(b, 1.0, l)
        elif case == "three_levels_f32":
            r = (0.5 * a + 0.25 * (a & b)).astype(np.float32)  # {0, .5, .75}: steps 2 : 1
        elif case == "two_levels_f64":
            r = np.where(a, 0.8, 0.05)
        else:
            r = 0.6 * a + 0.4 * (a & b) * 0.0 + 0.4 * b * a  # {0, .6, 1}: steps 3 : 2
        refs.append(r)
        shift = int(rng.randint(-900, 900))
        cs = []
        for j in range(n_cand):
            c = np.roll(a, shift * (j == 0)) ^ (rng.rand(n) < 0.02).astype(np.uint8) if j == 0 else _runs01(rng, n - 500 * j, 280)
            cs.append(np.ascontiguousarray(c, dtype=np.uint8))
        cands.append(cs)
    rd = np.float32 if case == "three_levels_f32" else np.float64
    amps = np.tile([1.0, 0.96, 1.0], (n_pairs, 1))
    db = _batch(refs, cands, rd, amps)
    n_fft = db.required_fft_length(3000)
    got, st = _solve(db, n_fft, 3000, "auto", n_cand=n_cand, pairs_in_flight=3)
    assert st[0] == 1 and st[2] == 0, (case, st)
    want, _ = _solve(db, n_fft, 3000, "fft", n_cand=n_cand, pairs_in_flight=3)
    _agree(got, want, rel=1e-6 if rd == np.float32 else 1e-9)
    for p in range(0, n_pairs, 2):
        for j in range(n_cand):
            sc, off = orc.fft_align(np.asarray(refs[p], dtype=np.float64), cands[p][j].astype(np.float64) * amps[p, j], 3000)
            assert int(got[0][p, j]["offset"]) == off, (case, p, j)
            assert float(got[0][p, j]["score"]) == pytest.approx(sc, rel=1e-6, abs=1e-5)


@pytest.mark.parametrize("case", ["five_levels", "no_common_quantum", "rare_level", "continuous"])
def test_level_sets_that_fall_back_to_the_transforms(case):
    """What the threshold decomposition cannot take -- more than four levels, steps without a small common quantum, a
    level the 2048-sample probe misses (found by the full pass: `ok` is cleared), noise -- goes through the transforms,
    per sub-batch; records identical to FFS_ALGO_FFT's."""
    rng = np.random.RandomState(7)
    n_pairs, n_cand, n = 4, 2, 60_000
    refs, cands = [], []
    for p in range(n_pairs):
        a, b = _runs01(rng, n, 250), _runs01(rng, n, 330)
        r = 0.6 * a + 0.4 * b
        if p == 1:  # (pair 1 of every case is the odd one; pairs_in_flight = 2 puts it into the first sub-batch only)
            if case == "five_levels":
                r = r + 0.05 * _runs01(rng, n, 500)
            elif case == "no_common_quantum":
                r = 0.6180339887 * a + 0.3819660113 * b * 0.77
            elif case == "rare_level":
                r = r.copy()
                r[12_345] = 0.5
            else:
                r = r + rng.uniform(-0.01, 0.01, n)
        refs.append(r)
        cands.append([np.ascontiguousarray(np.roll(a, 100 * (j + 1)) ^ (rng.rand(n) < 0.03).astype(np.uint8)) for j in range(n_cand)])
    db = _batch(refs, cands)
    n_fft = db.required_fft_length(2000)
    got, st = _solve(db, n_fft, 2000, "auto", n_cand=n_cand, pairs_in_flight=2)
    assert st == (1, 2, 1), (case, st)  # two sub-batches, the one with the odd pair through the transforms
    want, _ = _solve(db, n_fft, 2000, "fft", n_cand=n_cand, pairs_in_flight=2)
    assert np.array_equal(got[0]["offset"], want[0]["offset"])
    assert np.array_equal(got[0]["score"][:2], want[0]["score"][:2])  # the sub-batch the transforms solved: their very records
    assert np.allclose(got[0]["score"], want[0]["score"], rtol=1e-9, atol=1e-6)


def test_multi_level_reference_against_list_candidates():
    """Roles of three kinds in one call: float64 four-level references, candidates as boundary lists (FFS_DTYPE_RUNS) --
    the candidate's edge windows come from its list; records as with bit-packed candidates."""
    import torch

    from ffsubsync_amd import _native, batch
    from workloads import synth

    specs = [synth.make_pair_spec(g["seed"]) for g in GOLD[:6]]
    db = synth.build_fused_batch(specs)
    n_fft = db.required_fft_length(6000)
    want, _ = _solve(db, n_fft, 6000, "auto")
    # the candidates' lists: convert the bit-packed candidate vectors, keep the float references where they are
    cand_bits = batch.DeviceBatch(db.data, db.offs[:, 1:], db.lens[:, 1:], db.lo[:, 1:], db.hi[:, 1:], _native.FFS_DTYPE_U1)
    two = batch.DeviceBatch(db.data, np.concatenate([db.offs[:, 1:2], db.offs[:, 1:]], axis=1),
                            np.concatenate([db.lens[:, 1:2], db.lens[:, 1:]], axis=1), np.concatenate([db.lo[:, 1:2], db.lo[:, 1:]], axis=1),
                            np.concatenate([db.hi[:, 1:2], db.hi[:, 1:]], axis=1), _native.FFS_DTYPE_U1)
    lists = two.to_runs(cap=8192)  # (column 0 is a throw-away copy of candidate 0: to_runs converts whole batches)
    base = (db.data.numel() + 63) // 64 * 64
    pad = base - db.data.numel()
    data = torch.cat([db.data, torch.zeros(pad, dtype=torch.uint8, device=db.data.device), lists.data])
    offs = np.concatenate([db.offs[:, :1], lists.offs[:, 1:] + base], axis=1)
    mixed = batch.DeviceBatch(data, offs, db.lens, db.lo, db.hi, _native.FFS_DTYPE_RUNS, ref_dtype=_native.FFS_DTYPE_F64)
    got, st = _solve(mixed, n_fft, 6000, "auto")
    assert st[0] == 1 and st[2] == 0, st
    for f in ("score", "offset", "flags"):
        assert np.array_equal(got[0][f], want[0][f]), f
    assert np.array_equal(got[1], want[1])
