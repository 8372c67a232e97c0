import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def _bounded_threads():
    # the GPU box exposes every host core but gives one GPU a 16-core share: oversubscribing
    # torch-CPU (the oracle) makes it many times slower
    import os

    import torch
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, n)))


def pytest_configure(config):
    _bounded_threads()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return "cuda:0"


def needs_dev_build(precision: str = "") -> None:
    """Skip unless the loaded library is the dev build (``make -C deepemia_amd/csrc DEV=1`` + ``DEEPEMIA_DEV_LIB=1``): the product
    library computes f16x2 (default), f16 and exact f32 only."""
    from deepemia_amd import _lib
    if precision in ("", "f32x3", "f16x2r", "bf16x2", "bf16") and not _lib.is_dev_build():
        pytest.skip(f"precision {precision or '(non-default)'} needs the dev build of the library (DEEPEMIA_DEV_LIB=1)")
