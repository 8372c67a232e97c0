"""``system_logger`` with the reference's name, format and handlers
(reference ``src/utils/logger_utils.py:44-63``): file ``<logs>/system_<ts>.log`` at DEBUG and
console at INFO.  The log directory is ``~/logs`` unless ``DEEPEMIA_LOG_DIR`` is set (the GPU
box's home is not writable in every harness)."""
from __future__ import annotations

import logging
import os
import tempfile
from datetime import datetime
from pathlib import Path

_FMT = "%(asctime)s [%(levelname)s] %(message)s"


def get_log_dir() -> Path:
    base = Path(os.environ.get("DEEPEMIA_LOG_DIR", str(Path.home() / "logs")))
    try:
        base.mkdir(parents=True, exist_ok=True)
        probe = base / ".w"
        probe.touch()
        probe.unlink()
    except OSError:
        base = Path(tempfile.gettempdir()) / "deepemia_logs"
        base.mkdir(parents=True, exist_ok=True)
    return base


system_logger = logging.getLogger("system")
if not system_logger.handlers:
    system_logger.setLevel(logging.DEBUG)
    LOG_DIR = get_log_dir()
    _fh = logging.FileHandler(LOG_DIR / f"system_{datetime.now():%Y-%m-%d_%H-%M-%S}.log", encoding="utf-8")
    _fh.setFormatter(logging.Formatter(_FMT))
    system_logger.addHandler(_fh)
    _ch = logging.StreamHandler()
    _ch.setFormatter(logging.Formatter(_FMT))
    _ch.setLevel(logging.INFO)
    system_logger.addHandler(_ch)


def set_console_log_level(level=logging.INFO) -> None:
    for h in system_logger.handlers:
        if isinstance(h, logging.StreamHandler) and not isinstance(h, logging.FileHandler):
            h.setLevel(level)


_HAS_GPU = None


def log_memory_usage(stage: str = "") -> None:
    """RSS and HBM use at one pipeline stage (reference ``logger_utils.py:66-95``)."""
    try:
        import psutil
        import torch

        rss = psutil.Process().memory_info().rss / 1024 ** 2
        global _HAS_GPU
        if _HAS_GPU is None:
            _HAS_GPU = bool(torch.cuda.is_available())       # (asked once: the call costs ~0.1 ms and this runs twice per image)
        if _HAS_GPU:
            system_logger.info(f"[{stage}] Memory - RAM: {rss:.1f}MB, GPU: {torch.cuda.memory_allocated() / 1024 ** 3:.2f}GB "
                               f"allocated / {torch.cuda.memory_reserved() / 1024 ** 3:.2f}GB reserved")
        else:
            system_logger.info(f"[{stage}] Memory - RAM: {rss:.1f}MB")
    except Exception as e:  # pragma: no cover
        system_logger.debug(f"[{stage}] Could not log memory usage: {e}")
