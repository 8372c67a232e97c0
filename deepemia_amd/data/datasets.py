"""Dataset registration kept as a drop-in (reference ``src/data/datasets.py:65-153,242-258``).

Only the parts the inference path consumes are reproduced: ``read_dataset_info`` and the
``thing_classes`` metadata that ``register_datasets`` attaches to ``<dataset>_train`` /
``<dataset>_test`` (``inference.py:599-603``).  The training dicts the reference loads and discards
(``inference.py:598,607``) are registered as a lazy no-op; the train/test split and the custom-JSON
-> Detectron2 dict conversion are training-side and out of scope.
"""
from __future__ import annotations

import json
from types import SimpleNamespace
from typing import Callable, Dict

from ..utils.logger_utils import system_logger


class _Metadata(SimpleNamespace):
    def set(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
        return self

    def get(self, key, default=None):
        return getattr(self, key, default)


class _MetadataCatalog:
    """``detectron2.data.MetadataCatalog`` work-alike: ``get(name)`` creates on first use."""

    def __init__(self):
        self._m: Dict[str, _Metadata] = {}

    def get(self, name: str) -> _Metadata:
        if name not in self._m:
            self._m[name] = _Metadata(name=name)
        return self._m[name]

    def list(self):
        return list(self._m)

    def remove(self, name):
        self._m.pop(name, None)


class _DatasetCatalog:
    """``detectron2.data.DatasetCatalog`` work-alike (lazy callables)."""

    def __init__(self):
        self._d: Dict[str, Callable] = {}

    def register(self, name: str, func: Callable) -> None:
        self._d[name] = func

    def get(self, name: str):
        if name not in self._d:
            raise KeyError(f"Dataset '{name}' is not registered! Available datasets are: {', '.join(self._d)}")
        return self._d[name]()

    def list(self):
        return list(self._d)

    def remove(self, name):
        self._d.pop(name, None)


MetadataCatalog = _MetadataCatalog()
DatasetCatalog = _DatasetCatalog()


def read_dataset_info(file_path) -> dict:
    """``{name: (img_dir, label_dir, [classes])}`` from ``dataset_info.json`` (``datasets.py:242-258``)."""
    with open(file_path, "r") as f:
        data = json.load(f)
    info = {k: tuple(v) if isinstance(v, list) else v for k, v in data.items()}
    system_logger.info(f"Dataset Info: {info}")
    return info


def register_datasets(dataset_info, dataset_name, test_size=0.2, dataset_format="json"):
    """Make ``MetadataCatalog.get(f"{dataset_name}_train").thing_classes`` available
    (``datasets.py:65-153``).  Raises ``ValueError`` for an unknown dataset / format, as the reference."""
    if dataset_format not in ("json", "coco"):
        raise ValueError(f"Unknown dataset_format: {dataset_format}")
    if dataset_name not in dataset_info:
        raise ValueError(f"Dataset '{dataset_name}' not found in dataset_info.")
    _img_dir, _label_dir, thing_classes = dataset_info[dataset_name]
    for split in ("train", "test"):
        DatasetCatalog.register(f"{dataset_name}_{split}", lambda: [])  # training dicts: not needed for inference
        MetadataCatalog.get(f"{dataset_name}_{split}").set(thing_classes=list(thing_classes))
    system_logger.info(f"Registered dataset '{dataset_name}' ({dataset_format}) with classes {list(thing_classes)}")
