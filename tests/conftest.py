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


def pytest_report_header(config):
    """Which build of the C-ABI library this run loads (``product`` unless DEEPEMIA_DEV_LIB=1) -- the GPU suite's record says so."""
    try:
        from deepemia_amd import _lib
        lib = _lib.load()
        return f"deepemia library: {_lib.LIB_PATH.name} (flavour {lib.demia_build_flavor().decode()}, arch {lib.demia_build_arch().decode()}, ABI v{lib.demia_abi_version()})"
    except Exception as e:          # a CPU box without the built library still runs the oracle tests
        return f"deepemia library: not loadable here ({type(e).__name__})"


def pytest_sessionstart(session):
    _progress("session start: " + pytest_report_header(session.config))      # (-q hides the header: the progress log keeps it)


def pytest_collection_modifyitems(config, items):
    # a GPU test that hangs must end by itself (a silent run is killed by the pool after seven minutes, with no record of where)
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900))


def _progress(line: str) -> None:
    try:
        out = ROOT / "gpurun_out"
        out.mkdir(exist_ok=True)
        with open(out / "pytest_progress.log", "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def pytest_runtest_logstart(nodeid, location):
    import time
    _progress(f"{time.strftime('%H:%M:%S')} start {nodeid}")


def pytest_runtest_logreport(report):
    import time
    if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
        _progress(f"{time.strftime('%H:%M:%S')} {report.outcome:7s} {report.nodeid} ({report.duration:.1f} s)")


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
