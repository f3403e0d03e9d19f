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


def install(detectors: bool = True, device_rasters: bool = False) -> None:
    """Swap the GPU path into an importable ffsubsync.

    Aligner: the caller binds the classes by name (ffsubsync/ffsubsync.py:14), the same seam
    tests/test_quality_gate.py:98-102 patches.  VAD (``detectors``): the reference's
    ``VideoSpeechTransformer`` picks its detector factory as a module attribute
    (speech_transformers.py:655-679); the auditok factory is replaced by the GPU frame-energy sweep +
    token smoothing, everything around it (ffmpeg pipe, embedded-subtitle shortcut, progress, the
    multi-segment thread pool) stays the reference's own code.

    ``device_rasters``: the activity vectors between the pipelines and the aligner stay in HBM.  The
    ``speech_extract`` step that ``make_subtitle_speech_pipeline`` builds (speech_transformers.py:56-98, the
    class looked up in that module at call time) becomes :class:`DeviceSubtitleSpeechTransformer` -- interval
    lists go to the GPU, bit-packed rasters come out -- and the reference vector of the video / deserialised
    transformers is handed over as a bit-packed device copy made once per fitted vector, so that
    ``MaxScoreAligner.fit`` (ffsubsync.py:230-235) reaches ``ffs_align_batch`` with device rasters only instead of
    converting and uploading 46 MB of float64 per solve.  ``numpy.asarray`` of such a raster gives the reference's
    float64 vector back (``--serialize-speech`` keeps working)."""
    import ffsubsync.aligners as ref_aligners
    import ffsubsync.ffsubsync as ref_main

    # import everything that will be patched BEFORE patching anything: a failure half way (a native wheel's OSError, a
    # pkg_resources error from auditok ...) must not leave ffsubsync with the aligners swapped and the rest untouched
    ref_st = None
    if detectors or device_rasters:
        try:
            import ffsubsync.speech_transformers as ref_st  # noqa: F811
        except (ImportError, OSError) as exc:  # the VAD side of ffsubsync is not importable here: aligner-only install
            import logging

            logging.getLogger(__name__).warning("ffsubsync.speech_transformers is not importable (%s): aligner-only install", exc)
    for mod in (ref_aligners, ref_main):
        mod.FFTAligner = FFTAligner
        mod.MaxScoreAligner = MaxScoreAligner
    # keep `except FailedToFindAlignmentException` clauses in the caller working (already the case when ffsubsync
    # was importable at import time: the class is then bound in aligners.py)
    global FailedToFindAlignmentException
    from . import aligners as _al

    _al.FailedToFindAlignmentException = ref_aligners.FailedToFindAlignmentException
    FailedToFindAlignmentException = ref_aligners.FailedToFindAlignmentException
    if ref_st is None:
        return
    if detectors:
        from .speech_transformers import install_detectors

        install_detectors(ref_st)
    if device_rasters:
        from .subtitle_raster import install_device_rasters

        install_device_rasters(ref_st, ref_main)
