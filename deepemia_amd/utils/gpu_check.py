"""Device report (reference ``src/utils/gpu_check.py``).  On ROCm ``torch.cuda.*`` is the HIP
runtime; the MI355X path REQUIRES a device -- there is no CPU inference fallback to offer."""
from __future__ import annotations

import torch

from .logger_utils import system_logger


def log_device_info() -> None:
    if torch.cuda.is_available():
        i = torch.cuda.current_device()
        p = torch.cuda.get_device_properties(i)
        system_logger.info(f"GPU {i}: {p.name}, {p.total_memory / 1024 ** 3:.0f} GiB HBM, {p.multi_processor_count} CUs "
                           f"(HIP {torch.version.hip})")
    else:
        system_logger.warning("No HIP device visible")


def check_gpu_availability(require_gpu: bool = False, interactive: bool = True) -> bool:
    """True when a device is present.  The reference prompts y/n to continue on CPU
    (``gpu_check.py:18-91``); this build has no CPU path, so absence is reported and, when
    ``require_gpu`` is set, raised."""
    if torch.cuda.is_available():
        return True
    msg = "No MI355X / HIP device available: the deepEMIA hot path has no CPU fallback in this build"
    system_logger.error(msg)
    if require_gpu:
        raise RuntimeError(msg)
    return False
