import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # libffsalign.so is a build product (git-ignored): build it once if this checkout has none yet
    lib = os.path.join(ROOT, "ffsubsync_amd", "libffsalign.so")
    if not os.path.exists(lib):
        import subprocess

        subprocess.run(["make", "-C", os.path.join(ROOT, "ffsubsync_amd", "csrc")], check=False)


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# Modules written against the TRANSFORM path (rounds 1-3): their "identical records" / kernel-instantiation tests must
# keep exercising the FFT kernels now that bit-packed two-level vectors default to the run-boundary path
# (FFS_ALGORITHM=auto).  tests/test_gpu_runs.py runs the same parity cases, and its own, on the new path.
_TRANSFORM_PATH_MODULES = {"test_gpu_parity", "test_gpu_round2", "test_gpu_round3", "test_gpu_headline", "test_gpu_float",
                           "test_gpu_raster"}


@pytest.fixture(autouse=True)
def _transform_path_for_its_own_tests(request, monkeypatch):
    if request.module.__name__.split(".")[-1] in _TRANSFORM_PATH_MODULES and "FFS_ALGORITHM" not in os.environ:
        monkeypatch.setenv("FFS_ALGORITHM", "fft")
    yield
