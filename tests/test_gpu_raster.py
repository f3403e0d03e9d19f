"""On-device subtitle rasteriser against the committed outputs of the unmodified reference classes
(SubtitleScaler + SubtitleSpeechTransformer), and the HBM-resident pipeline into the aligner."""
import os
from datetime import timedelta

import numpy as np
import pytest

from oracle import aligners_oracle as orc
from oracle import raster_oracle as ro
from oracle import vad_oracle as vo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "raster_golden.npz"))
RATIOS = [float(r) for r in GOLD["ratios"]]


@pytest.mark.parametrize("name", ["a", "b", "late_start"])
def test_device_rasters_match_reference(name):
    from ffsubsync_amd import _native

    s, e, m = GOLD[name + "_start_us"], GOLD[name + "_end_us"], GOLD[name + "_meta"]
    ss = int(GOLD[name + "_start_seconds"])
    for j, r in enumerate(RATIOS):
        got = _native.rasterize_subtitles(s, e, m, r, 100, ss).cpu().numpy()
        assert np.array_equal(got, GOLD["%s_r%d" % (name, j)])


class _Sub:
    def __init__(self, s, e, content):
        self.start, self.end, self.content = timedelta(microseconds=int(s)), timedelta(microseconds=int(e)), content


def test_speech_extract_dropin_and_boundaries():
    from ffsubsync_amd.subtitle_raster import DeviceSubtitleSpeechTransformer

    s, e, m = GOLD["a_start_us"], GOLD["a_end_us"], GOLD["a_meta"]
    subs = [_Sub(a, b, "[music]" if k else "hello") for a, b, k in zip(s, e, m)]
    t = DeviceSubtitleSpeechTransformer(100, 0, 1.0417, is_metadata=lambda c, edge: c == "[music]").fit(subs)
    want = ro.rasterize(s, e, m, 1.0, 100, 0) * min(1 / 1.0417, 1.0)  # times already scaled upstream
    assert np.array_equal(np.asarray(t.transform()), want)
    assert (t.start_frame_, t.end_frame_) == orc.speech_boundaries(want)
    assert t.num_frames == t.end_frame_ - t.start_frame_
    assert t.max_time_ == max(e) / 1e6


def test_hbm_resident_candidates_through_the_aligner():
    """Intervals -> seven device rasters -> MaxScoreAligner, no host activity vectors: same answer as
    the reference-shaped host path (oracle on the reference's float arrays)."""
    from ffsubsync_amd.aligners import FFTAligner, MaxScoreAligner
    from ffsubsync_amd.constants import candidate_ratios
    from ffsubsync_amd.subtitle_raster import rasterize_candidates

    s, e, m = ro.synth_subtitles(31, n=170, minutes=10.0)
    ratios = candidate_ratios()
    # reference vector: the same track at ratio 25/24, shifted by +412 frames, as a 0/1 host array
    truth = ro.rasterize(s, e, m, ratios[3], 100, 0)
    ref = np.concatenate([np.zeros(412), (truth > 0).astype(float), np.zeros(900)])
    cands = rasterize_candidates(s, e, m, ratios)
    (score, offset), winner = MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform(ref, cands)
    host = [ro.rasterize(s, e, m, r, 100, 0) for r in ratios]
    (o_score, o_offset), idx = orc.max_score_align(ref, host, 6000)
    assert winner is cands[idx] and idx == 3 and offset == o_offset == 412
    assert score == pytest.approx(o_score, rel=1e-9)
    # a DeviceRaster reference works too
    from ffsubsync_amd.subtitle_raster import DeviceRaster
    import torch

    dref = DeviceRaster(torch.from_numpy((ref > 0).astype(np.uint8)).cuda())
    assert FFTAligner(6000).fit_transform(dref, cands[3]) == 412


def test_batched_gss_equals_per_file_gss_and_oracle():
    from ffsubsync_amd.aligners import FFTAligner, MaxScoreAligner
    from ffsubsync_amd.batch_gss import fit_gss_batch
    from ffsubsync_amd.subtitle_raster import rasterize_candidates

    files = []
    for seed, true_ratio, shift in [(41, 1.0427, 250), (42, 0.9590, -130), (43, 1.0, 77)]:
        s, e, m = ro.synth_subtitles(seed, n=120, minutes=6.0)
        truth = (ro.rasterize(s, e, m, true_ratio, 100, 0) > 0).astype(float)
        ref = np.concatenate([np.zeros(max(shift, 0)), truth[max(-shift, 0):], np.zeros(500)])
        files.append((ref, (s, e, m), true_ratio))
    got = fit_gss_batch([f[0] for f in files], [f[1] for f in files], max_offset_samples=6000)
    for (ref, (s, e, m), true_ratio), ((score, offset), ratio) in zip(files, got):
        # the same search, one file at a time, through the drop-in class
        class Pipe:
            def __init__(self, r):
                self.r = r

            def fit_transform(self, *_):
                return rasterize_candidates(s, e, m, [self.r])[0]

        msa = MaxScoreAligner(FFTAligner(max_offset_samples=6000))
        msa.fit(ref, [lambda r: Pipe(r)])
        (s1, o1), pipe = msa.transform()
        assert (repr(pipe.r), o1, float(s1)) == (repr(ratio), offset, float(score))
        # (the score is not unimodal in the ratio, so -- as in the reference -- the search may settle
        # on a local optimum; parity is about reproducing its evaluations, not about finding true_ratio)
    # and against the CPU oracle's search on the first file (reference-shaped float rasters)
    ref, (s, e, m), _ = files[0]
    rec = {}

    def objective(r, last):
        sc, off = orc.fft_align(ref, ro.rasterize(s, e, m, r, 100, 0), 6000)
        if last:
            rec["v"] = (r, sc, off)
        return -sc

    orc.gss_trace(objective, 0.9, 1.1)
    assert repr(rec["v"][0]) == repr(got[0][1]) and rec["v"][2] == got[0][0][1]
    assert rec["v"][1] == pytest.approx(got[0][0][0], rel=1e-9)


def test_end_to_end_pcm_to_offset():
    """BASELINE config 5 at small scale: 48 kHz PCM from a pipe -> chunked on-GPU frame-energy VAD
    (pinned double-buffered ingest) -> seven HBM-resident subtitle rasters -> MaxScoreAligner."""
    import io

    from ffsubsync_amd.aligners import FFTAligner, MaxScoreAligner
    from ffsubsync_amd.constants import candidate_ratios
    from ffsubsync_amd.speech_transformers import PCMSpeechTransformer
    from ffsubsync_amd.subtitle_raster import rasterize_candidates

    ratios = candidate_ratios()
    s, e, m = ro.synth_subtitles(51, n=140, minutes=8.0)
    true_idx, shift = 5, 733
    speech = np.concatenate([np.zeros(shift, bool), ro.rasterize(s, e, m, ratios[true_idx], 100, 0) > 0, np.zeros(300, bool)])
    rng = np.random.RandomState(0)
    pcm = np.rint(rng.randn(speech.size * 480) * np.repeat(np.where(speech, 3000.0, 30.0), 480))
    pcm = np.clip(pcm, -32768, 32767).astype("<i2")
    seen = []
    t = PCMSpeechTransformer("energy", 100, 48000, 0.0, progress_handler=seen.append).fit(io.BytesIO(pcm.tobytes()))
    labels = t.transform()
    assert labels.dtype == np.float64 and np.array_equal(labels > 0.5, speech)
    assert len(seen) == -(-pcm.size * 2 // (960 * 10000)) and seen == sorted(seen)
    cands = rasterize_candidates(s, e, m, ratios)
    (score, offset), winner = MaxScoreAligner(FFTAligner, None, 100, 60).fit_transform(labels, cands)
    assert winner is cands[true_idx] and offset == shift
