"""Boundary lists as an input format (FFS_DTYPE_RUNS, round 5): the subtitle rasteriser that writes lists instead of
bitmaps, the bits -> list conversion, and solves fed with lists -- against the bit rasteriser (itself pinned to the
unmodified reference by raster_golden.npz), numpy, and the bit-input solves of the same vectors ("identical records").
Through the C ABI; need a real MI355X.
"""
import json
import os

import numpy as np
import pytest

from oracle import raster_oracle as ro

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "raster_golden.npz"))
RATIOS = [float(r) for r in GOLD["ratios"]]
HEAD = json.load(open(os.path.join(HERE, "golden", "headline_golden.json")))["pairs"]


@pytest.fixture(scope="module")
def torch():
    import torch as t

    assert t.cuda.is_available()
    return t


def _bits_of(words, n):
    return np.unpackbits(words.cpu().numpy().view(np.uint8), bitorder="little")[:n]


def _list_of(bits01):
    """numpy model of a boundary list: positions where the value changes (a run that reaches the end closes at len) and
    the ones in front of each."""
    x = np.concatenate([[0], bits01.astype(np.int64), [0]])
    pos = np.flatnonzero(np.diff(x))
    ones_before = np.concatenate([[0], np.cumsum(bits01.astype(np.int64))])[pos]
    return pos, ones_before, int(bits01.sum())


def _lists_from_tracks(tracks, track_of, ratio):
    from ffsubsync_amd import batch

    ts = batch.TrackSet(tracks)
    data, offs, lens, bounds = ts.rasterize_runs(track_of, ratio)
    return ts, data, offs, lens, bounds


@pytest.mark.parametrize("name", ["a", "b"])
def test_list_rasteriser_equals_the_reference_rasters(torch, name):
    """raster_golden.npz = SubtitleScaler + SubtitleSpeechTransformer of the unmodified reference: the list, expanded to
    bits, must be that raster; its ones-in-front column must be the running count."""
    from ffsubsync_amd import _native

    s, e, m = GOLD[name + "_start_us"], GOLD[name + "_end_us"], GOLD[name + "_meta"]
    assert int(GOLD[name + "_start_seconds"]) == 0
    ts, data, offs, lens, bounds = _lists_from_tracks([(s, e, m)], np.zeros(len(RATIOS), np.int64), np.array(RATIOS))
    for j in range(len(RATIOS)):
        want = (GOLD["%s_r%d" % (name, j)] != 0).astype(np.uint8)
        assert lens[j] == want.size
        block = data[int(offs[j]):]
        got = _bits_of(_native.runs_to_bits(block, int(lens[j])), int(lens[j]))
        assert np.array_equal(got, want), (name, j)
        pos, ones_before, ones = _native.runs_list_host(block)
        wpos, wones, wtot = _list_of(want)
        assert np.array_equal(pos, wpos) and np.array_equal(ones_before, wones) and ones == wtot
        assert len(pos) <= bounds[j]


def test_list_rasteriser_overlaps_touching_unsorted_metadata_empty(torch):
    """Overlapping and touching subtitles merge, unsorted tracks are sorted in staging, metadata lines and empty /
    negative-duration intervals are skipped, tracks longer than one 1024-subtitle chunk carry their state across chunks;
    the same vectors through the bit rasteriser (ffs_rasterize_batch_bits) are the yardstick, bit for bit."""
    from ffsubsync_amd import _native, batch

    rng = np.random.RandomState(5)
    tracks = []
    # 0: hand-made: overlap, touch, containment, duplicate, zero and negative duration, metadata
    s = np.array([0, 1_000_000, 1_500_000, 3_000_000, 3_000_000, 5_000_000, 6_000_000, 6_000_000, 9_000_000, 9_500_000, 12_000_000])
    e = np.array([500_000, 2_000_000, 1_800_000, 4_000_000, 3_500_000, 6_000_000, 7_000_000, 6_000_000, 8_000_000, 11_000_000, 12_010_000])
    m = np.array([0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0], dtype=np.uint8)
    tracks.append((s, e, m))
    # 1: 3000 random, unsorted, heavily overlapping subtitles (three chunks)
    s1 = rng.randint(0, 600_000_000, 3000).astype(np.int64)
    e1 = s1 + rng.randint(-200_000, 4_000_000, 3000)
    tracks.append((s1, e1, (rng.rand(3000) < 0.05).astype(np.uint8)))
    # 2: sorted, touching chains (end == next start), no metadata array
    s2 = np.arange(0, 2000, dtype=np.int64) * 1_000_000
    e2 = s2 + np.where(np.arange(2000) % 3 == 0, 1_000_000, 700_000)
    tracks.append((s2, e2, None))
    # 3: a single subtitle; 4: nothing but metadata
    tracks.append((np.array([1_230_000]), np.array([4_560_000]), None))
    tracks.append((np.array([0, 2_000_000]), np.array([1_000_000, 3_000_000]), np.array([1, 1], dtype=np.uint8)))
    ratios = [1.0, 1.0417, 0.96, 1.001]
    track_of = np.repeat(np.arange(len(tracks)), len(ratios))
    ratio = np.tile(ratios, len(tracks))
    ts = batch.TrackSet(tracks)
    d_bits, o_bits, l_bits = ts.rasterize(track_of, ratio)
    d_runs, o_runs, l_runs, bounds = ts.rasterize_runs(track_of, ratio)
    assert np.array_equal(l_bits, l_runs)
    for v in range(track_of.size):
        n = int(l_bits[v])
        want = _bits_of(d_bits[int(o_bits[v]): int(o_bits[v]) + (n + 31) // 32 * 4], n)
        block = d_runs[int(o_runs[v]):]
        got = _bits_of(_native.runs_to_bits(block, n), n)
        assert np.array_equal(got, want), v
        pos, ones_before, ones = _native.runs_list_host(block)
        wpos, wones, wtot = _list_of(want)
        assert np.array_equal(pos, wpos) and np.array_equal(ones_before, wones) and ones == wtot, v
        raw = block[:16].view(torch.int32).cpu().numpy()
        assert raw[2] == n and raw[0] <= bounds[v]
        sent = block[16 + 8 * int(raw[0]): 24 + 8 * int(raw[0])].view(torch.int32).cpu().numpy()
        assert sent[0] == np.iinfo(np.int32).max and sent[1] == wtot


def test_list_rasteriser_refuses_positive_start_seconds(torch):
    from ffsubsync_amd import _native

    out = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    with pytest.raises(_native.NativeError):
        _native.rasterize_batch_runs(np.array([0]), np.array([10 ** 6]), None, [0], [1], [1.0], [0], [3], [102], out, 100.0, 5.0)


@pytest.mark.parametrize("n,density,run", [(720_000, 0.35, 300), (70_001, 0.5, 3), (33, 0.5, 1), (4096, 0.0, 1), (4097, 1.0, 1)])
def test_runs_from_bits_round_trip(torch, n, density, run):
    from ffsubsync_amd import _native

    rng = np.random.RandomState(n)
    if density in (0.0, 1.0):
        x = np.full(n, int(density), dtype=np.uint8)
    else:
        x = np.repeat((rng.rand(n // run + 1) < density).astype(np.uint8), run)[:n]
    words = torch.from_numpy(np.packbits(np.concatenate([x, np.zeros(-n % 32, np.uint8)]), bitorder="little").view(np.int32).copy()).cuda()
    wpos, wones, wtot = _list_of(x)
    cap = max(len(wpos) + 1, 4)
    block = _native.runs_from_bits(words, n, cap)
    pos, ones_before, ones = _native.runs_list_host(block)
    assert np.array_equal(pos, wpos) and np.array_equal(ones_before, wones) and ones == wtot
    assert np.array_equal(_bits_of(_native.runs_to_bits(block, n), n), x)
    if len(wpos) > 8:  # too small a block: the header says so (n >= cap), nothing is written beyond the block
        small = torch.full((4 + 2 * 8 + 16,), -7, dtype=torch.int32, device="cuda")
        _native.runs_from_bits(words, n, 8, out=small)
        raw = small.cpu().numpy()
        assert raw[0] >= 8 and (raw[4 + 16:] == -7).all()


@pytest.mark.parametrize("n_vec", [24, 300, 900])
def test_batched_extraction_in_every_workgroup_size(torch, n_vec):
    """``ffs_runs_from_bits_batch`` picks the extraction kernel by the number of vectors of the call (round 6: 1024-thread
    workgroups up to 256 vectors, 512 threads up to 768, 256 beyond): random vectors of 1 .. 40 000 samples -- lengths that
    are multiples of 32, runs that reach the end, empty and full vectors, run lengths from 1 (every word holds boundaries:
    the per-wave rings drain while they fill) to 3 000 -- against the numpy model, entry by entry; blocks that are too
    small are truncated at their capacity, nothing is written behind them."""
    from ffsubsync_amd import _native

    rng = np.random.RandomState(n_vec)
    vecs, words, offs, total = [], [], [], 0
    for v in range(n_vec):
        n = int(rng.choice([rng.randint(1, 200), rng.randint(200, 40000), 32 * rng.randint(1, 600)]))
        kind = v % 7
        if kind == 0:
            x = np.zeros(n, np.uint8)
        elif kind == 1:
            x = np.ones(n, np.uint8)
        else:
            run = int(rng.choice([1, 2, 7, 60, 400, 3000]))
            x = np.repeat((rng.rand(n // run + 2) < rng.choice([0.1, 0.5, 0.9])).astype(np.uint8), run)[:n]
            if kind == 2:
                x[-1] = 1  # a run that reaches the end
        vecs.append(x)
        w = np.packbits(np.concatenate([x, np.zeros(-n % 32, np.uint8)]), bitorder="little").view(np.int32)
        offs.append(total)
        words.append(w)
        total += (w.size + 15) // 16 * 16
    host = np.zeros(total, np.int32)
    for w, o in zip(words, offs):
        host[o:o + w.size] = w
    dev = torch.from_numpy(host).cuda()
    models = [_list_of(x) for x in vecs]
    caps = np.array([(len(m[0]) + 1) if v % 5 else max(1, len(m[0]) // 2) for v, m in enumerate(models)], dtype=np.int64)
    block_words = 4 + 2 * caps + 8  # header, entries, eight guard words
    boffs = np.concatenate([[0], np.cumsum(block_words)[:-1]])
    blocks = torch.full((int(block_words.sum()),), -7, dtype=torch.int32, device="cuda")
    _native.runs_from_bits_batch(dev.data_ptr() + 4 * np.array(offs, dtype=np.uint64), [x.size for x in vecs],
                                 blocks.data_ptr() + 4 * boffs.astype(np.uint64), caps)
    raw = blocks.cpu().numpy()
    for v, (x, (pos, ones_before, ones)) in enumerate(zip(vecs, models)):
        b = raw[boffs[v]: boffs[v] + block_words[v]]
        cap = int(caps[v])
        assert b[2] == x.size and b[3] == cap, v
        assert (b[4 + 2 * cap:] == -7).all(), v  # nothing behind the block
        if len(pos) < cap:
            assert b[0] == len(pos) and b[1] == ones, (v, b[:4], len(pos))
            e = b[4: 4 + 2 * len(pos)].reshape(-1, 2)
            assert np.array_equal(e[:, 0], pos) and np.array_equal(e[:, 1], ones_before), v
            assert b[4 + 2 * len(pos)] == np.iinfo(np.int32).max and b[5 + 2 * len(pos)] == ones, v
        else:  # truncated: the header says so, the entries that fit are the list's first ones
            assert b[0] >= cap, (v, b[:4], cap)
            e = b[4: 4 + 2 * cap].reshape(-1, 2)
            assert np.array_equal(e[:, 0], pos[:cap]) and np.array_equal(e[:, 1], ones_before[:cap]), v


def _same_records(a, b):
    ca, pa = a
    cb, pb = b
    for f in ("score", "offset", "flags", "score_f32"):
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

    specs = [synth.make_pair_spec(g["seed"]) for g in HEAD[:96]]
    db = synth.build_device_batch(specs)
    yield specs, db, HEAD[:96]
    del db
    torch.cuda.empty_cache()


def test_lists_give_the_records_of_the_bits(headline):
    """configs[2] pairs: the vectors as bits, as lists converted from those bits, with only the candidates as lists
    (reference still bits) and with only the reference as a list -- identical records, equal to the unmodified
    reference's goldens."""
    import test_gpu_headline as th
    from ffsubsync_amd import _native, batch

    specs, db, gold = headline
    n_fft = db.required_fft_length(6000)
    a, st_a = _solve(db, n_fft, 6000, "auto")
    dl = db.to_runs(cap=8192)
    b, st_b = _solve(dl, n_fft, 6000, "auto")
    assert st_a[2] == 0 and st_b[2] == 0
    _same_records(a, b)
    th._check_seven(b[1], b[0], gold)
    # roles of different kinds: the reference's bits + the candidates' lists, in one buffer
    base = (db.data.numel() + 63) // 64 * 64  # (_torch_cat pads the first buffer to a multiple of 64 bytes)
    both = batch.DeviceBatch(_torch_cat(db.data, dl.data), np.concatenate([db.offs[:, :1], dl.offs[:, 1:] + base], axis=1),
                             db.lens, db.lo, db.hi, _native.FFS_DTYPE_RUNS, _native.FFS_DTYPE_U1)
    c, st_c = _solve(both, n_fft, 6000, "auto")
    assert st_c[2] == 0
    _same_records(a, c)
    # ... and the other way round: the reference's list (caller-owned: position + ones in front) + the candidates' bits,
    # whose plan-owned lists hold positions only
    other = batch.DeviceBatch(_torch_cat(db.data, dl.data), np.concatenate([dl.offs[:, :1] + base, db.offs[:, 1:]], axis=1),
                              db.lens, db.lo, db.hi, _native.FFS_DTYPE_U1, _native.FFS_DTYPE_RUNS)
    e, st_e = _solve(other, n_fft, 6000, "auto")
    assert st_e[2] == 0
    _same_records(a, e)
    # host-known bounds: the call needs nothing back from the device
    n_b = np.zeros(dl.offs.shape, dtype=np.int32)
    for i, o in enumerate(dl.offs.ravel()):
        n_b.ravel()[i] = int(dl.data[int(o): int(o) + 4].view(_native.require_gpu().int32).item())
    bounded = batch.DeviceBatch(dl.data, dl.offs, dl.lens, dl.lo, dl.hi, _native.FFS_DTYPE_RUNS, None, n_b + 2)
    d, st_d = _solve(bounded, n_fft, 6000, "auto")
    _same_records(a, d)


def _torch_cat(a, b):
    import torch

    pad = (-a.numel()) % 64
    return torch.cat([a, torch.zeros(pad, dtype=a.dtype, device=a.device), b]) if pad else torch.cat([a, b])


@pytest.mark.parametrize("max_off", [None, 150000])
def test_lists_in_wide_and_absent_windows(headline, max_off):
    """Windows of many tiles (k_runs_pick) and the edge masks taken from the lists (no bitmap to fetch them from):
    single-ratio solves, lists against bits."""
    specs, db, gold = headline
    from workloads import synth

    db16 = synth.build_device_batch(specs[:16])
    one = db16.select_candidates([sp.true_ratio_index for sp in specs[:16]])
    n_fft = one.required_fft_length(max_off)
    a, _ = _solve(one, n_fft, max_off, "runs", n_cand=1, pairs_in_flight=16)
    b, st = _solve(one.to_runs(cap=8192), n_fft, max_off, "runs", n_cand=1, pairs_in_flight=16)
    assert st[2] == 0
    _same_records(a, b)


def test_short_vectors_and_edge_masks_from_lists(torch):
    """Short, dense-ish vectors of unequal lengths with runs touching both ends: every lag's one-sided counts come from
    list-derived edge masks; bits against lists, and both against the CPU oracle."""
    from ffsubsync_amd import _native, batch
    from oracle import aligners_oracle as orc

    rng = np.random.RandomState(77)
    n_pairs, n_cand = 24, 3
    vecs, lens = [], np.zeros((n_pairs, 1 + n_cand), dtype=np.int64)
    for p in range(n_pairs):
        for j in range(1 + n_cand):
            n = int(rng.randint(5000, 9000))
            x = np.repeat((rng.rand(n // 40 + 2) < 0.5).astype(np.uint8), 40)[:n]
            if p % 3 == 0:
                x[:50] = 1
            if p % 4 == 1:
                x[-70:] = 1
            vecs.append(x)
            lens[p, j] = n
    nbytes = (lens + 31) // 32 * 4
    offs, total = batch._layout(lens, nbytes)
    host = np.zeros(total, dtype=np.uint8)
    for x, o in zip(vecs, offs.ravel()):
        pk = np.packbits(np.concatenate([x, np.zeros(-x.size % 32, np.uint8)]), bitorder="little")
        host[int(o): int(o) + pk.size] = pk
    db = batch.DeviceBatch(torch.from_numpy(host).cuda(), offs, lens, np.zeros(lens.shape), np.ones(lens.shape), _native.FFS_DTYPE_U1)
    for max_off in (None, 700):
        n_fft = max(db.required_fft_length(max_off), 16384)
        a, _ = _solve(db, n_fft, max_off, "runs", n_cand=n_cand, pairs_in_flight=8)
        b, st = _solve(db.to_runs(cap=1024), n_fft, max_off, "runs", n_cand=n_cand, pairs_in_flight=8)
        assert st[2] == 0
        _same_records(a, b)
        for p in range(0, n_pairs, 5):
            for j in range(n_cand):
                conv, n_sub = orc.convolve_full(vecs[p * (1 + n_cand)].astype(float), vecs[p * (1 + n_cand) + 1 + j].astype(float))
                masked = orc.mask_extreme_offsets(conv, n_sub, max_off)
                k = len(masked) - 1 - int(b[0][p, j]["offset"]) - n_sub  # the device's lag in the reference's array
                assert abs(masked[k] - masked.max()) < 1e-6 and float(b[0][p, j]["score"]) == pytest.approx(masked.max(), abs=1e-6)
                assert k == int(np.flatnonzero(masked >= masked.max() - 1e-6)[0])  # first maximum = largest lag among ties


def test_dense_lists_fall_back_to_the_transforms(torch):
    """Lists over the coincidence budget: the sub-batch is expanded to bits and solved by the transforms; FFS_ALGO_FFT on
    list inputs does the same for everything.  Records as from the bits."""
    from workloads import synth

    specs = [synth.make_pair_spec(7000 + i, run_scale=0.0625) for i in range(8)]
    db = synth.build_device_batch(specs)
    n_fft = db.required_fft_length(6000)
    a, _ = _solve(db, n_fft, 6000, "fft", pairs_in_flight=8)
    dl = db.to_runs()
    b, st = _solve(dl, n_fft, 6000, "auto", pairs_in_flight=8)
    assert st[:3] == (1, 1, 1)
    _same_records_but_f32(a, b)
    c, st_c = _solve(dl, n_fft, 6000, "fft", pairs_in_flight=8)
    assert st_c == (0, 0, 0)
    _same_records_but_f32(a, c)


def _same_records_but_f32(a, b):
    for f in ("score", "offset", "flags"):
        assert np.array_equal(a[0][f], b[0][f]), f
    assert np.array_equal(a[1], b[1])


def test_truncated_list_is_an_error(torch):
    from ffsubsync_amd import _native
    from workloads import synth

    specs = [synth.make_pair_spec(3)]
    db = synth.build_device_batch(specs)
    dl = db.to_runs(cap=64)  # far too small: every list is truncated
    with pytest.raises(_native.NativeError) as err:
        _solve(dl, db.required_fft_length(6000), 6000, "auto", pairs_in_flight=1)
    assert "truncated" in str(err.value)
    # the same error (not an out-of-bounds read) when the lists only get expanded for the transforms
    with pytest.raises(_native.NativeError) as err:
        _solve(dl, db.required_fft_length(6000), 6000, "fft", pairs_in_flight=1)
    assert "truncated" in str(err.value)
    # ffs_runs_to_bits on a truncated block: asynchronous, so no error -- but nothing is read or written past the block
    blk = dl.data[int(dl.offs[0, 0]):]
    guard = torch.full((int(dl.lens[0, 0] + 31) // 32 + 8,), -7, dtype=torch.int32, device="cuda")
    _native.runs_to_bits(blk, int(dl.lens[0, 0]), out=guard)
    assert (guard[int(dl.lens[0, 0] + 31) // 32:] == -7).all()


def test_host_bounds_of_a_full_histogram_cell_prove_nothing(torch):
    """a candidate list of 32 768 entries or more (16-bit histogram cells) must not take k_runs_corr even when
    the caller's bounds put its coincidence count within the budget (a sparse reference): the host-side shortcut leaves
    the decision to the device, the sub-batch goes through the transforms, records as from the bits."""
    from ffsubsync_amd import batch
    from ffsubsync_amd.constants import candidate_ratios
    from workloads import synth

    ratios = candidate_ratios()
    n_sub = 17_000  # non-overlapping 0.1 s subtitles every 0.4 s: 34 000 boundaries > RUNS_CAP
    s = (np.arange(n_sub, dtype=np.int64) * 400_000) + 1_000_000
    cand = (s, s + 100_000, np.zeros(n_sub, dtype=np.uint8))
    rs = np.arange(40, dtype=np.int64) * 150_000_000 + 7_000_000  # forty 20 s runs: a sparse reference
    ref = (rs, rs + 20_000_000, np.zeros(40, dtype=np.uint8))
    sparse = synth.make_subtitle_records(31, duration_s=6800.0)
    recs = [(ref, cand), (ref, sparse)]
    d_bits = batch.pairs_from_intervals(recs, ratios)
    d_runs = batch.pairs_from_intervals(recs, ratios, lists=True)
    assert int(d_runs.bounds[0, 1]) >= 32768
    n_fft = d_bits.required_fft_length(6000)
    want, _ = _solve(d_bits, n_fft, 6000, "fft", pairs_in_flight=1)
    got, st = _solve(d_runs, n_fft, 6000, "auto", pairs_in_flight=1)
    assert st[:3] == (1, 2, 1)  # pair 0 through the transforms, pair 1 on the run-boundary path
    _same_records_but_f32(want, got)


def test_interval_lists_to_records_without_a_bitmap(torch):
    """The device-resident pipeline of round 5: interval lists -> ffs_rasterize_batch_runs -> ffs_align_batch_runs with
    host-known bounds; same records as interval lists -> bit rasters -> extraction -> solve."""
    from ffsubsync_amd import batch
    from ffsubsync_amd.constants import candidate_ratios
    from workloads import synth

    ratios = candidate_ratios()
    recs = []
    for i in range(12):
        ref = synth.make_subtitle_records(100 + i, duration_s=1800.0)
        s, e, m = synth.make_subtitle_records(200 + i, duration_s=1800.0)
        recs.append((ref, (s + 1_370_000, e + 1_370_000, m)))
    d_bits = batch.pairs_from_intervals(recs, ratios)
    d_runs = batch.pairs_from_intervals(recs, ratios, lists=True)
    assert np.array_equal(d_bits.lens, d_runs.lens) and d_runs.bounds is not None
    n_fft = d_bits.required_fft_length(6000)
    a, _ = _solve(d_bits, n_fft, 6000, "auto", pairs_in_flight=12)
    b, st = _solve(d_runs, n_fft, 6000, "auto", pairs_in_flight=12)
    assert st[2] == 0
    _same_records(a, b)


def test_negative_start_seconds_reaches_the_list_rasteriser(torch):
    """``TrackSet.rasterize_runs`` takes ``start_seconds`` (speech_transformers.py:968-972) like the bit
    rasteriser: lists and bits of the same tracks with ``start_seconds = -3`` are the same vectors (lists expanded ==
    bits), ``pairs_from_intervals(lists=True)`` gives the records of the bit path, and both equal the restated rasteriser
    + aligner on the host; a positive ``start_seconds`` is refused for lists."""
    from ffsubsync_amd import _native, batch
    from ffsubsync_amd.constants import candidate_ratios
    from oracle import aligners_oracle as orc
    from oracle import raster_oracle as ro
    from workloads import synth

    ratios = candidate_ratios()
    recs = []
    for i in range(6):
        ref = synth.make_subtitle_records(300 + i, duration_s=900.0)
        s, e, m = synth.make_subtitle_records(400 + i, duration_s=900.0)
        recs.append((ref, (s + 2_110_000, e + 2_110_000, m)))
    for ss in (-3, -0.5):
        d_bits = batch.pairs_from_intervals(recs, ratios, start_seconds=ss)
        d_runs = batch.pairs_from_intervals(recs, ratios, start_seconds=ss, lists=True)
        assert np.array_equal(d_bits.lens, d_runs.lens)
        for p in range(len(recs)):
            for v in range(1 + len(ratios)):
                n = int(d_bits.lens[p, v])
                o_b, o_r = int(d_bits.offs[p, v]), int(d_runs.offs[p, v])
                want = d_bits.data[o_b: o_b + (n + 31) // 32 * 4].view(torch.int32)
                got = _native.runs_to_bits(d_runs.data[o_r:], n)[: (n + 31) // 32]
                assert torch.equal(got, want), (ss, p, v)
        n_fft = d_bits.required_fft_length(6000)
        a, _ = _solve(d_bits, n_fft, 6000, "auto", pairs_in_flight=6)
        b, st = _solve(d_runs, n_fft, 6000, "auto", pairs_in_flight=6)
        assert st[2] == 0
        _same_records(a, b)
        # pair 0, every ratio: the restated rasteriser (the reference's own arithmetic, raster_golden-pinned) + aligner
        (rs, re_, rm), (cs, ce, cm) = recs[0]
        host_ref = ro.rasterize(rs, re_, rm, 1.0, 100, ss)
        for j, ratio in enumerate(ratios):
            o_s, o_o = orc.fft_align(host_ref, ro.rasterize(cs, ce, cm, ratio, 100, ss), 6000)
            assert int(b[0][0, j]["offset"]) == o_o and float(b[0][0, j]["score"]) == pytest.approx(o_s, rel=1e-9), (ss, j)
    with pytest.raises(ValueError):
        batch.pairs_from_intervals(recs, ratios, start_seconds=2, lists=True)
    with pytest.raises(ValueError):
        batch.TrackSet([recs[0][1]]).rasterize_runs([0], [1.0], 100, 1.5)


def test_batched_gss_with_a_negative_start_seconds(torch):
    """``fit_gss_batch(start_seconds=-3)`` stays on the lists (use_lists = start_seconds <= 0) and must rasterise them with
    that start: same recorded evaluations as the per-file search through the bit rasteriser."""
    from ffsubsync_amd import batch_gss
    from oracle import raster_oracle as ro
    from workloads import synth

    refs, subs = [], []
    for i in range(5):
        s, e, m = synth.make_subtitle_records(520 + i, duration_s=900.0)
        truth = ro.rasterize(s, e, m, 1.0 + 0.01 * (i - 2), 100, -3)
        refs.append(np.concatenate([np.zeros(150 + 20 * i), (truth > 0).astype(float), np.zeros(300)]))
        subs.append((s, e, m))
    stats = {}
    got = batch_gss.fit_gss_batch(refs, subs, 6000, 100, -3, stats=stats)
    assert stats["lists"] is True
    want = batch_gss._fit_gss_batch_per_file(refs, subs, 6000, 100, -3)
    assert [(float(s_), o, r) for (s_, o), r in got] == [(float(s_), o, r) for (s_, o), r in want]
    zero = batch_gss.fit_gss_batch(refs, subs, 6000, 100, 0)
    assert [o for (_, o), _ in zero] != [o for (_, o), _ in got]  # (the start does move the answer: 300 samples)


def test_dense_stream_is_probed_and_sparse_data_returns_to_the_lists(torch):
    """FFS_ALGO_AUTO on a stream of dense calls: after the first call fell back to the transforms, the next ones sample
    the vectors (k_runs_probe) and go straight to the transforms without extracting lists; a sparse call after that takes
    the run-boundary path again.  Records as from FFS_ALGO_FFT throughout."""
    from ffsubsync_amd import batch
    from workloads import synth

    dense = synth.build_device_batch([synth.make_pair_spec(7000 + i, run_scale=0.0625) for i in range(8)])
    sparse = synth.build_device_batch([synth.make_pair_spec(i) for i in range(8)])
    n_fft = max(dense.required_fft_length(6000), sparse.required_fft_length(6000))
    want_dense, _ = _solve(dense, n_fft, 6000, "fft", pairs_in_flight=8)
    want_sparse, _ = _solve(sparse, n_fft, 6000, "fft", pairs_in_flight=8)
    al = batch.BatchAligner(n_fft, 7, 6000, pairs_in_flight=8, algorithm="auto")
    for k in range(3):
        got = al.solve(dense)
        _same_records_but_f32(want_dense, got)
        assert al.plan.runs_stats() == (k + 1, k + 1, k + 1)
    got = al.solve(sparse)  # probed (the previous call was dense), found sparse, solved from its lists
    _same_records_but_f32(want_sparse, got)
    assert al.plan.runs_stats() == (4, 4, 3)
    got = al.solve(dense)
    _same_records_but_f32(want_dense, got)
    assert al.plan.runs_stats() == (5, 5, 4)
    al.close()


def test_pinned_and_device_resident_tables_give_the_same_lists(torch):
    """TrackSet.pin() (tables uploaded straight from pinned memory) and TrackSet.to_device() (tables uploaded once) against
    pageable host tables: byte-identical list buffers, unsorted tracks included."""
    from ffsubsync_amd import batch

    rng = np.random.RandomState(9)
    tracks = []
    for k in range(6):
        s = np.sort(rng.randint(0, 400_000_000, 700)).astype(np.int64)
        if k % 2:
            rng.shuffle(s)
        tracks.append((s, s + rng.randint(100_000, 5_000_000, 700), (rng.rand(700) < 0.03).astype(np.uint8)))
    track_of = np.repeat(np.arange(6), 3)
    ratio = np.tile([1.0, 1.0417, 0.96], 6)
    want, offs, lens, _ = batch.TrackSet(tracks).rasterize_runs(track_of, ratio)
    for mode in ("pin", "to_device"):
        ts = batch.TrackSet(tracks)
        getattr(ts, mode)()
        got, o2, l2, _ = ts.rasterize_runs(track_of, ratio)
        assert np.array_equal(offs, o2) and np.array_equal(lens, l2)
        for v in range(track_of.size):  # (compare what each block holds: header + entries incl. the sentinel)
            n = int(want[int(offs[v]): int(offs[v]) + 4].view(torch.int32).item())
            nb = 16 + 8 * (n + 1)
            assert torch.equal(want[int(offs[v]): int(offs[v]) + nb], got[int(offs[v]): int(offs[v]) + nb]), (mode, v)


def test_large_host_tables_give_the_same_lists(torch):
    """8 MB of pageable subtitle tables (48 tracks of 10 000 heavily overlapping subtitles), one track unsorted: the same
    list blocks as device-resident tables, and their expansion equals the bit rasteriser's output."""
    from ffsubsync_amd import _native, batch

    rng = np.random.RandomState(21)
    tracks = []
    for k in range(48):
        s = np.sort(rng.randint(0, 2_000_000_000, 10_000)).astype(np.int64)
        if k == 17:
            rng.shuffle(s)
        tracks.append((s, s + rng.randint(50_000, 400_000, s.size), (rng.rand(s.size) < 0.01).astype(np.uint8)))
    track_of = np.arange(48)
    ratio = np.where(track_of % 2, 1.0417, 1.0)
    got, offs, lens, bounds = batch.TrackSet(tracks).rasterize_runs(track_of, ratio)
    dev = batch.TrackSet([tracks[k] if k != 17 else tuple(a[np.argsort(tracks[17][0], kind="stable")] for a in tracks[17])
                          for k in range(48)])
    dev.to_device()
    want, o2, l2, _ = dev.rasterize_runs(track_of, ratio)
    assert np.array_equal(offs, o2) and np.array_equal(lens, l2)
    bits, boffs, blens = batch.TrackSet(tracks).rasterize(track_of, ratio)
    for v in (0, 1, 17, 47):
        n = int(got[int(offs[v]): int(offs[v]) + 4].view(torch.int32).item())
        assert 0 < n <= int(bounds[v])
        nb = 16 + 8 * (n + 1)
        assert torch.equal(want[int(offs[v]): int(offs[v]) + nb], got[int(offs[v]): int(offs[v]) + nb]), v
        blk = got[int(offs[v]): int(offs[v]) + 16 + 8 * (int(bounds[v]) + 2)].view(torch.int32)
        words = (int(lens[v]) + 31) // 32
        assert int(blens[v]) == int(lens[v])
        assert torch.equal(_native.runs_to_bits(blk, int(lens[v]))[:words],
                           bits[int(boffs[v]): int(boffs[v]) + 4 * words].view(torch.int32)), v


# ---- the drop-in classes on boundary lists -------------------------------------------------------------------------
def test_rasters_from_intervals_carry_their_boundary_lists():
    """rasterize_candidates / DeviceSubtitleSpeechTransformer: every raster's list (straight from the intervals) expands
    to exactly the raster's bits, including a negative start_seconds; a positive one keeps the bits only."""
    from datetime import timedelta

    from ffsubsync_amd import _native
    from ffsubsync_amd.subtitle_raster import DeviceSubtitleSpeechTransformer, rasterize_candidates
    from oracle import raster_oracle as ro

    s, e, m = ro.synth_subtitles(77, n=140, minutes=9.0)
    ratios = [0.9, 1.0, 25 / 24, 1.1]
    for ss in (0, -3):
        rasters = rasterize_candidates(s, e, m, ratios, 100, ss)
        for r in rasters:
            assert r.runs is not None and r.runs_bound >= int(r.runs[0].item())
            assert np.array_equal(_native.runs_to_bits(r.runs, r.n).cpu().numpy(), r.packed_words().cpu().numpy()[: (r.n + 31) // 32])
    assert all(r.runs is None for r in rasterize_candidates(s, e, m, ratios, 100, 2))

    class Sub:
        def __init__(self, a, b):
            self.start, self.end, self.content = timedelta(microseconds=int(a)), timedelta(microseconds=int(b)), "x"

    t = DeviceSubtitleSpeechTransformer(100, 0, 1.0).fit([Sub(a, b) for a, b in zip(s, e)])
    r = t.transform()
    assert r.runs is not None
    assert np.array_equal(_native.runs_to_bits(r.runs, r.n).cpu().numpy(), r.packed_words().cpu().numpy()[: (r.n + 31) // 32])


def test_drop_in_solves_from_lists_like_from_bits_and_host_arrays():
    """MaxScoreAligner(FFTAligner) on DeviceRasters that carry lists (no extraction, no wait for the device), on the same
    rasters without lists, and on the reference-shaped host arrays: one answer; the plan took the run-boundary path."""
    from ffsubsync_amd import _native
    from ffsubsync_amd.aligners import FFTAligner, MaxScoreAligner
    from ffsubsync_amd.constants import candidate_ratios
    from ffsubsync_amd.subtitle_raster import DeviceRaster, rasterize_candidates
    from oracle import aligners_oracle as orc
    from oracle import raster_oracle as ro

    s, e, m = ro.synth_subtitles(5, n=210, minutes=12.0)
    ratios = candidate_ratios()
    truth = ro.rasterize(s, e, m, ratios[4], 100, 0)
    ref = np.concatenate([np.zeros(233), (truth > 0).astype(float), np.zeros(500)])
    host = [ro.rasterize(s, e, m, r, 100, 0) for r in ratios]
    (o_score, o_offset), idx = orc.max_score_align(ref, host, 6000)
    with_lists = rasterize_candidates(s, e, m, ratios)
    dref = DeviceRaster.from_host(ref)
    assert dref.runs is not None and all(c.runs is not None for c in with_lists)
    without = [DeviceRaster(c.bits, c.lo, c.hi, c.n) for c in with_lists]
    dref0 = DeviceRaster.from_host(ref, lists=False)
    assert dref0.runs is None
    answers = []
    for r, cands in ((dref, with_lists), (dref0, without), (dref, without), (ref, host)):
        (score, offset), winner = MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform(r, list(cands))
        answers.append((float(score), int(offset), [i for i, c in enumerate(cands) if c is winner][0]))
    assert answers[0] == answers[1] == answers[2] == answers[3]
    assert answers[0][1] == o_offset == 233 and answers[0][2] == idx == 4
    assert answers[0][0] == pytest.approx(o_score, rel=1e-9)
    # a dense vector (more boundaries than a list may hold) keeps its bits only and still solves
    rng = np.random.default_rng(3)
    dense = DeviceRaster.from_host((rng.random(200000) < 0.5).astype(float))
    assert dense.runs is None
    assert FFTAligner(6000).fit_transform(dense, with_lists[4]) == orc.fft_align(np.asarray(dense), host[4], 6000)[1]
