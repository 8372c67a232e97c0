"""Thresholds shared with the reference (``src/utils/constants.py:19``)."""


class DefaultThresholds:
    MIN_CRYSTAL_SIZE = 2
