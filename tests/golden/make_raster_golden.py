"""Generate tests/golden/raster_golden.npz by running the UNMODIFIED reference classes
SubtitleScaler (ffsubsync/subtitle_transformers.py) and SubtitleSpeechTransformer
(ffsubsync/speech_transformers.py) on seeded subtitle records.  The reference modules import
ffmpeg / pysubs2 / srt at module level, which are not installed here; empty stand-in modules with the
few class names they touch are registered first (the rasterisation itself never calls into them).

    python tests/golden/make_raster_golden.py
"""
import logging
import os
import sys

sys.dont_write_bytecode = True  # importing the reference must not write __pycache__ into /root/reference
import types
from datetime import timedelta

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

for name in ("ffmpeg", "pysubs2", "srt", "tqdm", "webrtcvad", "auditok"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["pysubs2"].SSAFile = type("SSAFile", (), {})
sys.modules["pysubs2"].SSAEvent = type("SSAEvent", (), {})
sys.modules["pysubs2"].SSAStyle = type("SSAStyle", (), {})


class _SrtSubtitle:
    def __init__(self, content):
        self.content = content


sys.modules["srt"].Subtitle = _SrtSubtitle
sys.modules["tqdm"].tqdm = lambda *a, **k: None
_pkg = types.ModuleType("ffsubsync")
_pkg.__path__ = ["/root/reference/ffsubsync"]
sys.modules["ffsubsync"] = _pkg
logging.disable(logging.INFO)
from ffsubsync.generic_subtitles import GenericSubtitle, GenericSubtitlesFile  # noqa: E402
from ffsubsync.speech_transformers import SubtitleSpeechTransformer  # noqa: E402
from ffsubsync.subtitle_transformers import SubtitleScaler  # noqa: E402

from ffsubsync_amd.constants import candidate_ratios  # noqa: E402
from oracle import raster_oracle as ro  # noqa: E402


def reference_raster(start_us, end_us, meta, ratio, start_seconds):
    subs = [GenericSubtitle(timedelta(microseconds=int(s)), timedelta(microseconds=int(e)),
                            _SrtSubtitle("[music]" if m else "line %d" % i))
            for i, (s, e, m) in enumerate(zip(start_us, end_us, meta))]
    f = GenericSubtitlesFile(subs, sub_format="srt", encoding="utf-8")
    scaled = SubtitleScaler(ratio).fit(f).transform()
    t = SubtitleSpeechTransformer(sample_rate=100, start_seconds=start_seconds, framerate_ratio=ratio).fit(scaled)
    return t.transform(), t.start_frame_, t.end_frame_


def main():
    out = {}
    cases = [("a", 21, 0), ("b", 22, 0), ("late_start", 23, 30)]
    ratios = candidate_ratios() + [1.0837, 0.9123]  # two gss-like ratios
    for name, seed, start_seconds in cases:
        s_us, e_us, meta = ro.synth_subtitles(seed)
        # reference metadata rule for these contents: "[music]" lines are metadata (paired nester)
        out[name + "_start_us"], out[name + "_end_us"], out[name + "_meta"] = s_us, e_us, meta
        out[name + "_start_seconds"] = np.array(start_seconds)
        for j, r in enumerate(ratios):
            arr, sf, ef = reference_raster(s_us, e_us, meta, r, start_seconds)
            amp = min(1.0 / r, 1.0)
            assert set(np.unique(arr)) <= {0.0, amp}
            out["%s_r%d" % (name, j)] = (arr > 0).astype(np.uint8)
            out["%s_bounds%d" % (name, j)] = np.array([sf if sf is not None else -1, ef if ef is not None else -1])
            assert np.array_equal(arr, ro.rasterize(s_us, e_us, meta, r, 100, start_seconds)), (name, r)
    out["ratios"] = np.array(ratios)
    np.savez_compressed(os.path.join(HERE, "raster_golden.npz"), **out)
    print("wrote raster_golden.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
