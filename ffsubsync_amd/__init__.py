"""ffsubsync_amd -- MI355X (gfx950) implementation of ffsubsync's alignment hot path.

Public surface (mirrors the reference modules of the same names):
    ffsubsync_amd.aligners            FFTAligner, MaxScoreAligner, FailedToFindAlignmentException
    ffsubsync_amd.golden_section_search.gss
    ffsubsync_amd.speech_transformers _make_energy_detector / _make_auditok_detector (factory seam),
                                      PCMSpeechTransformer, ComputeSpeechFrameBoundariesMixin,
                                      assemble_sparse_reference, load_speech_batch
    ffsubsync_amd.batch               BatchAligner / DeviceBatch (throughput path, multi-GPU sharding)
    ffsubsync_amd.install()           patch an importable ``ffsubsync`` to use the GPU aligner + detector
"""
from .aligners import (  # noqa: F401
    MAX_FRAMERATE_RATIO,
    MIN_FRAMERATE_RATIO,
    FailedToFindAlignmentException,
    FFTAligner,
    MaxScoreAligner,
)

__version__ = "0.1.0"


def install(detectors: bool = True) -> None:
    """Swap the GPU path into an importable ffsubsync.

    Aligner: the caller binds the classes by name (ffsubsync/ffsubsync.py:14), the same seam
    tests/test_quality_gate.py:98-102 patches.  VAD (``detectors``): the reference's
    ``VideoSpeechTransformer`` picks its detector factory as a module attribute
    (speech_transformers.py:655-679); the auditok factory is replaced by the GPU frame-energy sweep +
    token smoothing, everything around it (ffmpeg pipe, embedded-subtitle shortcut, progress, the
    multi-segment thread pool) stays the reference's own code."""
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
    if detectors:
        try:
            import ffsubsync.speech_transformers as ref_st
        except Exception:  # the VAD side of ffsubsync is not importable here: aligner-only install
            return
        from .speech_transformers import install_detectors

        install_detectors(ref_st)
