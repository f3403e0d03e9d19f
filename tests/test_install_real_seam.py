"""ffsubsync_amd.install() against the REAL reference modules (build container only: needs
/root/reference).  The reference's third-party imports that are not installed here (ffmpeg, pysubs2, srt,
tqdm, ...) are replaced by empty stand-in modules, exactly as tests/golden/make_raster_golden.py does;
everything under test -- ffsubsync.ffsubsync.try_sync, Pipeline, SubtitleScaler,
SubtitleSpeechTransformer, SubtitleShifter -- is the unmodified reference code.

The scenario runs in a subprocess (the stub `ffsubsync` parent package must not leak into this test
session): after install(), the reference's own try_sync drives OUR MaxScoreAligner through its seven
framerate-ratio pipelines and applies the offset it returns.  There is no GPU here, so the base aligner
bound by name in ffsubsync.ffsubsync is a CPU stand-in with the FFTAligner interface (the numpy oracle);
the GPU FFTAligner behind the same seam is covered by the `-m gpu` tests."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCENARIO = r'''
import logging, sys, types
from datetime import timedelta
import numpy as np

for name in ("ffmpeg", "pysubs2", "srt", "tqdm", "webrtcvad", "auditok", "chardet", "charset_normalizer", "cchardet", "rich",
             "rich.console", "rich.logging"):
    sys.modules.setdefault(name, types.ModuleType(name))
for cls in ("SSAFile", "SSAEvent", "SSAStyle"):
    setattr(sys.modules["pysubs2"], cls, type(cls, (), {}))
class _SrtSubtitle:
    def __init__(self, content):
        self.content = content
sys.modules["srt"].Subtitle = _SrtSubtitle
sys.modules["tqdm"].tqdm = lambda *a, **k: None
pkg = types.ModuleType("ffsubsync"); pkg.__path__ = ["/root/reference/ffsubsync"]; sys.modules["ffsubsync"] = pkg
logging.disable(logging.CRITICAL)

import ffsubsync.ffsubsync as ref_main                      # the unmodified reference
import ffsubsync.aligners as ref_aligners
import ffsubsync.speech_transformers as ref_st
import ffsubsync.sklearn_shim as ref_shim
from ffsubsync.generic_subtitles import GenericSubtitle, GenericSubtitlesFile
from ffsubsync.subtitle_transformers import SubtitleScaler

import ffsubsync_amd
from ffsubsync_amd import aligners as amd_aligners
from ffsubsync_amd import speech_transformers as amd_st
from oracle import aligners_oracle as orc
from workloads import synth

ref_exc = ref_aligners.FailedToFindAlignmentException
ffsubsync_amd.install()
# 1. the names the caller binds (ffsubsync/ffsubsync.py:14) now resolve to the drop-in classes ...
assert ref_main.MaxScoreAligner is amd_aligners.MaxScoreAligner and ref_main.FFTAligner is amd_aligners.FFTAligner
assert ref_aligners.MaxScoreAligner is amd_aligners.MaxScoreAligner
# ... which ARE reference TransformerMixins (isinstance checks in foreign code keep working) ...
assert issubclass(amd_aligners.FFTAligner, ref_shim.TransformerMixin)
assert issubclass(amd_aligners.MaxScoreAligner, ref_shim.TransformerMixin)
assert amd_aligners.Pipeline is ref_shim.Pipeline
# ... raise the reference's exception type, and the VAD factory seam points at the GPU detector
assert amd_aligners.FailedToFindAlignmentException is ref_exc
assert ref_st._make_auditok_detector is amd_st._make_auditok_detector

# 2. the reference's try_sync end to end on top of the installed MaxScoreAligner
class OracleFFTAligner(ref_shim.TransformerMixin):          # CPU stand-in with the FFTAligner interface (no GPU here)
    def __init__(self, max_offset_samples=None):
        self.max_offset_samples = max_offset_samples
    def fit(self, ref, sub, get_score=False):
        self.res = orc.fft_align(np.asarray(ref, dtype=float), np.asarray(sub, dtype=float), self.max_offset_samples)
        self.get_score_ = get_score
        return self
    def transform(self, *_):
        return self.res if self.get_score_ else self.res[1]
ref_main.FFTAligner = OracleFFTAligner

true_ratio, true_shift_s = 25.0 / 24.0, 7.31
s_us, e_us, _ = synth.make_subtitle_records(77, duration_s=540.0)
subs = [GenericSubtitle(timedelta(microseconds=int(a)), timedelta(microseconds=int(b)), _SrtSubtitle("line %d" % i))
        for i, (a, b) in enumerate(zip(s_us, e_us))]

class FakeParser(ref_shim.TransformerMixin):                # stands in for the srt parser (third-party `srt` is absent)
    encoding, max_subtitle_seconds, start_seconds = "infer", 10, 0
    def fit(self, fname, *_):
        self.subs_ = GenericSubtitlesFile(subs, sub_format="srt", encoding="utf-8")
        return self
    def transform(self, *_):
        return self.subs_
parser = FakeParser()
ref_main.get_srt_pipe_maker = lambda args, srtin: (lambda scale: ref_st.make_subtitle_speech_pipeline(
    scale_factor=scale, parser=parser, encoding="infer", max_subtitle_seconds=10, start_seconds=0))

# reference activity: the same track stretched by true_ratio and shifted by +7.31 s, as the VAD would see it
scaled = SubtitleScaler(true_ratio).fit(GenericSubtitlesFile(subs, sub_format="srt", encoding="utf-8")).transform()
truth = ref_st.SubtitleSpeechTransformer(sample_rate=100, start_seconds=0).fit(scaled).transform()
ref_signal = np.concatenate([np.zeros(int(round(true_shift_s * 100))), (truth > 0).astype(float), np.zeros(500)])
reference_pipe = types.SimpleNamespace(transform=lambda _: ref_signal)

written = {}
GenericSubtitlesFile.write_file = lambda self, fname: written.update(fname=fname, subs=list(self))
ref_main.get_version = lambda: "0.5.0"  # the checkout has no git metadata / packaged __version__ resource
args = ref_main.make_parser().parse_args(["ref.mkv", "-i", "in.srt", "-o", "out.srt"])
args.skip_infer_framerate_ratio = True
result = {"retval": 0}
ok = ref_main.try_sync(args, reference_pipe, result)
assert ok is True and result["sync_was_successful"] is True, result
assert abs(result["framerate_scale_factor"] - true_ratio) < 1e-12, result
assert abs(result["offset_seconds"] - true_shift_s) <= 0.02, result
assert written["fname"] == "out.srt" and len(written["subs"]) == len(subs)
first = written["subs"][0].start.total_seconds()
want = subs[0].start.total_seconds() * true_ratio + result["offset_seconds"]
assert abs(first - want) < 1e-3, (first, want)
print("REAL_SEAM_OK offset=%.2f scale=%.6f" % (result["offset_seconds"], result["framerate_scale_factor"]))
'''


@pytest.mark.skipif(not os.path.isdir("/root/reference/ffsubsync"), reason="needs the reference checkout")
def test_install_against_the_real_ffsubsync_modules():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               PYTHONDONTWRITEBYTECODE="1")  # never write into /root/reference
    out = subprocess.run([sys.executable, "-c", SCENARIO], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "REAL_SEAM_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
