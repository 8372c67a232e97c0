"""Spatial constraint filters (reference ``src/utils/spatial_constraints.py``), index-based.

Same names, rules, thresholds and removal order as the reference; the masks themselves stay on the
GPU bit-packed and are consulted through :class:`~deepemia_amd.utils.mask_algebra.DeviceMaskAlgebra`
(bounding boxes, areas and ``count_nonzero(m1 & m2)`` come from HIP reductions).
"""
from __future__ import annotations

import numpy as np
from typing import Dict, List, Sequence, Set

from .config import get_config
from .logger_utils import system_logger


def load_spatial_constraints(dataset_name=None) -> dict:
    """spatial_constraints.py:21-67 (same lookup order and defaults)."""
    default_config = {"enabled": False, "containment_rules": {}, "overlap_rules": {}, "containment_threshold": 0.95}
    try:
        config = get_config(dataset_name=dataset_name)
        spatial_config = None
        if "inference_overrides" in config:
            spatial_config = config["inference_overrides"].get("spatial_constraints")
        if spatial_config is None:
            spatial_config = config.get("inference_settings", {}).get("spatial_constraints", {})
        if spatial_config is None:
            spatial_config = config.get("spatial_constraints", {})
        if spatial_config is None:
            return default_config
        if dataset_name and dataset_name in spatial_config:
            spatial_config = spatial_config[dataset_name]
        result = {**default_config, **spatial_config}
        if result["enabled"]:
            system_logger.info(f"Spatial constraints ENABLED for '{dataset_name}'")
        return result
    except Exception as e:  # same behaviour as the reference: log and fall back to "disabled"
        system_logger.error(f"Error loading spatial constraints: {e}")
        return default_config


def bboxes_overlap(bbox1, bbox2) -> bool:
    """spatial_constraints.py:92-115; boxes are (y_min, x_min, y_max, x_max) or None."""
    if bbox1 is None or bbox2 is None:
        return False
    y1_min, x1_min, y1_max, x1_max = bbox1
    y2_min, x2_min, y2_max, x2_max = bbox2
    if x1_max < x2_min or x2_max < x1_min:
        return False
    if y1_max < y2_min or y2_max < y1_min:
        return False
    return True


def calculate_iou(alg, i: int, j: int) -> float:
    """spatial_constraints.py:118-153."""
    if not bboxes_overlap(alg.bbox_of(i), alg.bbox_of(j)):
        return 0.0
    intersection = alg.inter(i, j)
    if intersection == 0:
        return 0.0
    union = alg.union(i, j)
    if union == 0:
        return 0.0
    return intersection / union


def calculate_containment(alg, child: int, parent: int) -> float:
    """spatial_constraints.py:156-189."""
    if not bboxes_overlap(alg.bbox_of(child), alg.bbox_of(parent)):
        return 0.0
    child_area = int(alg.area[child])
    if child_area == 0:
        return 0.0
    return alg.inter(child, parent) / child_area


def filter_by_overlap_rules(alg, scores: Sequence[float], classes: Sequence[int], overlap_rules: dict,
                            indices: Sequence[int] = None) -> Set[int]:
    """spatial_constraints.py:192-277 -> removed indices (positions in ``indices``)."""
    removed: Set[int] = set()
    if not overlap_rules:
        return removed
    idx = list(range(alg.n)) if indices is None else list(indices)
    groups: Dict[int, List[int]] = {}
    for pos, cls in enumerate(classes):
        groups.setdefault(cls, []).append(pos)
    alg.prefetch_overlapping_pairs([[idx[p] for p in g] for c, g in groups.items() if c in overlap_rules])
    for cls, positions in groups.items():
        if cls not in overlap_rules:
            continue
        rule = overlap_rules[cls]
        allow_overlap = rule.get("allow_overlap", True)
        max_iou = rule.get("max_iou_threshold", 0.5)
        if allow_overlap and max_iou >= 0.9:
            continue
        order = sorted(positions, key=lambda p: scores[p], reverse=True)
        # box-overlap matrix of this class once (numpy) instead of one Python box test per pair: same pairs, same order
        bb = np.asarray([alg.bbox[idx[p]] for p in order], dtype=np.int64).reshape(-1, 4)
        okb = bb[:, 0] >= 0
        ov = ~((bb[:, None, 3] < bb[None, :, 1]) | (bb[None, :, 3] < bb[:, None, 1]) |
               (bb[:, None, 2] < bb[None, :, 0]) | (bb[None, :, 2] < bb[:, None, 0])) & okb[:, None] & okb[None, :]
        gone = np.zeros(len(order), dtype=bool)
        for a, p1 in enumerate(order):
            if gone[a]:
                continue
            cand = np.nonzero(ov[a, a + 1:] & ~gone[a + 1:])[0] + a + 1
            for b in cand:
                if calculate_iou(alg, idx[p1], idx[order[b]]) > max_iou:
                    gone[b] = True
                    removed.add(order[b])
    return removed


def filter_by_containment_rules(alg, scores: Sequence[float], classes: Sequence[int], containment_rules: dict,
                                containment_threshold: float = 0.95, indices: Sequence[int] = None) -> Set[int]:
    """spatial_constraints.py:280-398 -> removed indices (positions in ``indices``)."""
    removed: Set[int] = set()
    if not containment_rules:
        return removed
    idx = list(range(alg.n)) if indices is None else list(indices)
    by_class: Dict[int, List[int]] = {}
    for pos, cls in enumerate(classes):
        by_class.setdefault(cls, []).append(pos)
    for child_class, parent_class in containment_rules.items():
        if child_class not in by_class:
            continue
        if parent_class not in by_class:
            removed.update(by_class[child_class])
            continue
        parents = [p for p in by_class[parent_class] if p not in removed and alg.bbox_of(idx[p]) is not None]
        children = by_class[child_class]
        # child x parent box-overlap matrix once (numpy): the pairs the reference tests one by one
        cb = np.asarray([alg.bbox[idx[c]] for c in children], dtype=np.int64).reshape(-1, 4)
        pb = np.asarray([alg.bbox[idx[q]] for q in parents], dtype=np.int64).reshape(-1, 4)
        ov = ~((cb[:, None, 3] < pb[None, :, 1]) | (pb[None, :, 3] < cb[:, None, 1]) |
               (cb[:, None, 2] < pb[None, :, 0]) | (pb[None, :, 2] < cb[:, None, 0])) & (cb[:, 0] >= 0)[:, None]
        pi, pj = [], []
        for a, b in zip(*np.nonzero(ov)):
            if (idx[children[a]], idx[parents[b]]) not in alg._cache:
                pi.append(idx[children[a]])
                pj.append(idx[parents[b]])
        if pi:
            alg.intersections(pi, pj)
        for a, ch in enumerate(children):
            if ch in removed:
                continue
            if alg.bbox_of(idx[ch]) is None:
                removed.add(ch)
                continue
            max_containment = 0.0
            for b in np.nonzero(ov[a])[0]:
                p = parents[b]
                if p in removed:
                    continue
                c = calculate_containment(alg, idx[ch], idx[p])
                if c > max_containment:
                    max_containment = c
            if max_containment < containment_threshold:
                removed.add(ch)
    return removed


def apply_spatial_constraints_indices(alg, scores: Sequence[float], classes: Sequence[int], config: dict) -> List[int]:
    """spatial_constraints.py:401-460 with an already loaded constraint dict -> kept mask indices."""
    keep = list(range(alg.n))
    if alg.n == 0 or not config.get("enabled", False):
        return keep
    threshold = config.get("containment_threshold", 0.95)
    overlap_rules = config.get("overlap_rules", {})
    if overlap_rules:
        rem = filter_by_overlap_rules(alg, [scores[i] for i in keep], [classes[i] for i in keep], overlap_rules, keep)
        keep = [k for p, k in enumerate(keep) if p not in rem]
    containment_rules = config.get("containment_rules", {})
    if containment_rules:
        rem = filter_by_containment_rules(alg, [scores[i] for i in keep], [classes[i] for i in keep], containment_rules,
                                          threshold, keep)
        keep = [k for p, k in enumerate(keep) if p not in rem]
    removed = alg.n - len(keep)
    if removed > 0:
        system_logger.info(f"Spatial constraints removed {removed} instances ({alg.n} -> {len(keep)})")
    return keep


def apply_spatial_constraints(alg, scores, classes, dataset_name=None) -> List[int]:
    """Reference entry point (loads the dataset's constraint block, then filters)."""
    if alg.n == 0:
        return []
    return apply_spatial_constraints_indices(alg, scores, classes, load_spatial_constraints(dataset_name))
