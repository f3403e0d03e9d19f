"""ffsubsync_amd -- MI355X (gfx950) implementation of ffsubsync's alignment hot path.

Public surface (mirrors the reference modules of the same names):
    ffsubsync_amd.aligners            FFTAligner, MaxScoreAligner, FailedToFindAlignmentException
    ffsubsync_amd.golden_section_search.gss
    ffsubsync_amd.speech_transformers _make_energy_detector, PCMSpeechTransformer,
                                      ComputeSpeechFrameBoundariesMixin
    ffsubsync_amd.batch               BatchAligner / DeviceBatch (throughput path, multi-GPU sharding)
    ffsubsync_amd.install()           patch an importable ``ffsubsync`` to use the GPU aligner
"""
from .aligners import (  # noqa: F401
    MAX_FRAMERATE_RATIO,
    MIN_FRAMERATE_RATIO,
    FailedToFindAlignmentException,
    FFTAligner,
    MaxScoreAligner,
)

__version__ = "0.1.0"


def install() -> None:
    """Swap the GPU aligner into an importable ffsubsync: the caller binds the classes by name
    (ffsubsync/ffsubsync.py:14), the same seam tests/test_quality_gate.py:98-102 patches."""
    import ffsubsync.aligners as ref_aligners
    import ffsubsync.ffsubsync as ref_main

    for mod in (ref_aligners, ref_main):
        mod.FFTAligner = FFTAligner
        mod.MaxScoreAligner = MaxScoreAligner
    # keep `except FailedToFindAlignmentException` clauses in the caller working
    global FailedToFindAlignmentException
    from . import aligners as _al

    _al.FailedToFindAlignmentException = ref_aligners.FailedToFindAlignmentException
    FailedToFindAlignmentException = ref_aligners.FailedToFindAlignmentException
