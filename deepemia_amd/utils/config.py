"""YAML configuration singleton with per-dataset overrides -- same files, keys and merge rules as
the reference's ``src/utils/config.py:77-165`` so existing ``~/deepEMIA/config`` trees work as is.

Files:  ``<root>/config.yaml`` and ``<root>/datasets/<dataset>.yaml`` where ``<root>`` is
``~/deepEMIA/config`` (reference ``config.py:82,55``) unless ``DEEPEMIA_CONFIG_DIR`` points
elsewhere.  Dataset blocks are folded into the global dict as the reference does:

  ``inference_overrides``  -> deep-merged into ``inference_settings``
  ``scale_bar_roi``        -> ``scale_bar_rois[<dataset>]``
  ``scalebar_thresholds``  -> deep-merged
  ``scale_bar``            -> deep-merged (not in the reference: the label / calibration that replaces the OCR)
  ``spatial_constraints``  -> ``inference_settings.spatial_constraints[<dataset>]``
  ``rcnn_hyperparameters.best_R50 / best_R101`` -> ``rcnn_hyperparameters.best.R50 / R101``
"""
from __future__ import annotations

import copy
import os
from pathlib import Path
from typing import Any, Dict, Optional

import yaml

from .logger_utils import system_logger

_config: Optional[Dict[str, Any]] = None
_dataset_configs: Dict[str, Optional[Dict[str, Any]]] = {}


def config_root() -> Path:
    return Path(os.environ.get("DEEPEMIA_CONFIG_DIR", str(Path.home() / "deepEMIA" / "config")))


def reset_cache() -> None:
    global _config
    _config = None
    _dataset_configs.clear()


def deep_merge(base: Dict, override: Dict) -> Dict:
    out = dict(base)
    for k, v in override.items():
        out[k] = deep_merge(out[k], v) if isinstance(out.get(k), dict) and isinstance(v, dict) else v
    return out


def load_dataset_config(dataset_name: str) -> Optional[Dict[str, Any]]:
    if dataset_name in _dataset_configs:
        return _dataset_configs[dataset_name]
    f = config_root() / "datasets" / f"{dataset_name}.yaml"
    if not f.exists():
        system_logger.debug(f"No dataset-specific config found for '{dataset_name}'")
        return None
    try:
        cfg = yaml.safe_load(f.read_text())
    except Exception as e:
        system_logger.error(f"Error loading dataset config for '{dataset_name}': {e}")
        return None
    _dataset_configs[dataset_name] = cfg
    system_logger.info(f"Loaded dataset-specific config for '{dataset_name}'")
    return cfg


def get_config(dataset_name: str = None) -> Dict[str, Any]:
    global _config
    if _config is None:
        path = config_root() / "config.yaml"
        try:
            _config = yaml.safe_load(path.read_text())
        except FileNotFoundError:
            system_logger.error(f"Configuration file not found: {path}")
            raise
        system_logger.info(f"Loaded configuration from {path}")
    if dataset_name is None:
        return _config
    ds = load_dataset_config(dataset_name)
    if ds is None:
        return _config
    merged = copy.copy(_config)
    for k, v in _config.items():
        if isinstance(v, dict):
            merged[k] = deep_merge(v, {})
    if "inference_overrides" in ds:
        merged["inference_settings"] = deep_merge(merged.get("inference_settings", {}), ds["inference_overrides"])
    if "scale_bar_roi" in ds:
        merged.setdefault("scale_bar_rois", {})
        merged["scale_bar_rois"] = dict(merged["scale_bar_rois"])
        merged["scale_bar_rois"][dataset_name] = ds["scale_bar_roi"]
    if "scalebar_thresholds" in ds:
        merged["scalebar_thresholds"] = deep_merge(merged.get("scalebar_thresholds", {}), ds["scalebar_thresholds"])
    if "scale_bar" in ds:        # this build's extension (no OCR): label / text_center / um_per_pixel per dataset
        merged["scale_bar"] = deep_merge(merged.get("scale_bar", {}) or {}, ds["scale_bar"])
    if "spatial_constraints" in ds:
        inf = merged.setdefault("inference_settings", {})
        if "spatial_constraints" not in inf:
            inf["spatial_constraints"] = {}
        inf["spatial_constraints"] = dict(inf["spatial_constraints"])
        inf["spatial_constraints"][dataset_name] = ds["spatial_constraints"]
    if "rcnn_hyperparameters" in ds and "rcnn_hyperparameters" in merged:
        best = merged["rcnn_hyperparameters"].setdefault("best", {})
        for key in ("best_R50", "best_R101"):
            if key in ds["rcnn_hyperparameters"]:
                best[key.replace("best_", "")] = ds["rcnn_hyperparameters"][key]
    return merged
