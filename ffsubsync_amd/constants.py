"""Constants of the alignment hot path (values fixed by the reference, ffsubsync/constants.py:7-18)."""
from typing import List

SAMPLE_RATE: int = 100  # constants.py:7 -- activity-vector frames per second
FRAMERATE_RATIOS: List[float] = [24.0 / 23.976, 25.0 / 23.976, 25.0 / 24.0]  # constants.py:9
DEFAULT_FRAME_RATE: int = 48000  # constants.py:11 -- PCM samples per second
DEFAULT_NON_SPEECH_LABEL: float = 0.0  # constants.py:12
DEFAULT_MAX_OFFSET_SECONDS: int = 60  # constants.py:18
DEFAULT_ENERGY_THRESHOLD_DB: float = 50.0  # speech_transformers.py:124 (AudioEnergyValidator)


def candidate_ratios() -> List[float]:
    """[1.0] + FRAMERATE_RATIOS + 1/FRAMERATE_RATIOS, the order try_sync builds its pipelines in
    (ffsubsync.py:131-141, 196-199)."""
    return [1.0] + list(FRAMERATE_RATIOS) + [1.0 / r for r in FRAMERATE_RATIOS]
