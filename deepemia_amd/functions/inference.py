"""Tiled Mask R-CNN inference + morphometrics: the reference's ``run_inference`` pipeline
(``src/functions/inference.py:499-1350``) re-built around device-resident data.

Same entry point, config keys, decision rules (quirks included, SURVEY.md section 8 notes N1-N8) and output
files as the reference; what differs is where the data lives and how the work is batched:

* the image, its tiles and every mask stay in HBM; masks are bit-packed ``[M, H, W/32]``;
* all tiles of an image go through the network in ONE batched forward per model, and the output of
  a (model, image|tile) pair is computed once and reused by every class of the class loop (the
  reference re-runs the same forward once per class and only filters the result,
  ``inference.py:789,1411,1519``) -- identical results, ``num_classes`` times fewer forwards;
* hole filling, cross morphology, overlap removal, component tests, IoU / containment counts,
  contour tracing and the measurement reductions are HIP kernels (``deepemia_amd/csrc``); the
  greedy keep/remove loops (sequential by definition, a few hundred integers) run on the host
  over the kernels' integer tables.
"""
from __future__ import annotations

import csv
import math
import os
import time
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ..data.datasets import DatasetCatalog, MetadataCatalog, read_dataset_info, register_datasets
from ..data.models import choose_and_use_model, get_trained_model_paths
from .. import _lib as _L
from .. import parallel
from ..maskset import MaskOps
from ..utils.config import get_config
from ..utils.logger_utils import log_memory_usage, system_logger
from ..utils.mask_algebra import DeviceMaskAlgebra
from ..utils.measurements import contrast_percentiles
from ..utils.mask_utils import (mask_crops, postprocess_masks_device, postprocess_masks_universal_device,
                                process_masks_device, rle_crop_launch, rle_encoding_packed, rle_text_from_payload,
                                rle_text_packed)
from ..utils.spatial_constraints import apply_spatial_constraints_indices, load_spatial_constraints

CSV_HEADER = ["Instance_ID", "Class", "Class_Name", "Major axis length", "Minor axis length", "Eccentricity", "C. Length",
              "C. Width", "Circular eq. diameter", "Aspect ratio", "Circularity", "Chord length", "Ferret diameter",
              "Roundness", "Sphericity", "Contrast d10", "Contrast d50", "Contrast d90", "Detected scale bar", "File name"]
# measurement vector layout of demia_contour_measure -> CSV column order (inference.py:1209-1230)
_VAL = {"major": 0, "minor": 1, "ecc": 2, "Length": 3, "Width": 4, "CircularED": 5, "Aspect": 6, "Circularity": 7,
        "Chords": 8, "Feret": 9, "Roundness": 10, "Sphericity": 11}
CLASS_COLORS = [(0, 255, 0), (255, 0, 0), (0, 0, 255), (255, 255, 0), (255, 0, 255), (0, 255, 255), (128, 0, 128), (255, 165, 0)]


def is_image_file(filename: str) -> bool:
    return filename.lower().endswith((".tif", ".tiff", ".png", ".jpg", ".jpeg", ".bmp", ".gif"))


def get_image_folder_path(base_path: Optional[Path] = None) -> str:
    """``inference.py:370-404``: ``<root>/DATASET/INFERENCE`` or its ``UPLOAD`` sub-folder."""
    if base_path is None:
        root = Path(get_config()["paths"].get("local_dataset_root", "~")).expanduser()
        base_path = root / "DATASET" / "INFERENCE"
    inference_path = str(base_path)
    upload_path = os.path.join(inference_path, "UPLOAD")
    if os.path.isdir(inference_path) and any(os.path.isfile(os.path.join(inference_path, f)) for f in os.listdir(inference_path)):
        return inference_path
    if os.path.exists(upload_path) and any(os.path.isfile(os.path.join(upload_path, f)) for f in os.listdir(upload_path)):
        return upload_path
    raise FileNotFoundError("No images found in INFERENCE or INFERENCE/UPLOAD folders.")


def imread_bgr(path: str) -> Optional[np.ndarray]:
    """``cv2.imread`` stand-in: 8-bit, 3-channel, BGR; ``None`` when the file cannot be decoded."""
    from PIL import Image

    try:
        with Image.open(path) as im:
            if im.mode in ("I;16", "I;16B", "I;16L"):
                # cv2.imread(IMREAD_COLOR) reduces 16-bit samples to 8 bits by 1/256 (convertTo: round half to even)
                a = np.rint(np.asarray(im).astype(np.float64) / 256.0)
                rgb = np.repeat(np.clip(a, 0, 255).astype(np.uint8)[:, :, None], 3, axis=2)
            elif im.mode == "I":
                a = np.asarray(im).astype(np.float64)
                a = np.rint(a / 256.0) if a.max() > 255 else a
                rgb = np.repeat(np.clip(a, 0, 255).astype(np.uint8)[:, :, None], 3, axis=2)
            else:
                rgb = np.asarray(im.convert("RGB"))
    except Exception:
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


_scale_bar_warned = False


def get_scalebar_roi_for_dataset(dataset_name: Optional[str] = None) -> dict:
    """``scalebar_ocr.py:28-70``: ``scale_bar_rois[<dataset>]``, else ``scale_bar_rois.default``, else the built-in ROI."""
    default_roi = {"x_start_factor": 0.7, "y_start_factor": 0.05, "width_factor": 1, "height_factor": 0.05}
    try:
        rois = get_config(dataset_name=dataset_name).get("scale_bar_rois", {}) or {}
        if dataset_name and dataset_name in rois:
            return rois[dataset_name]
        return rois.get("default", default_roi)
    except Exception as e:                                                   # same fallback as the reference
        system_logger.error(f"Error loading scale bar ROI config: {e}")
        return default_roi


def _scale_bar_settings(dataset_name):
    try:
        return (get_config(dataset_name=dataset_name) if dataset_name else get_config()).get("scale_bar", {}) or {}
    except Exception:
        return {}


def scale_bar_needs_image(dataset_name=None) -> bool:
    """True when the bar's LINE has to be found in the image (a label is configured, the calibration is not)."""
    cfg = _scale_bar_settings(dataset_name)
    return os.environ.get("DEEPEMIA_UM_PER_PIXEL", cfg.get("um_per_pixel")) is None and cfg.get("label") is not None


def detect_scale_bar(image, roi_config=None, intensity_threshold=200, proximity_threshold=50, dataset_name=None,
                     draw_debug=False) -> Tuple[str, float]:
    """``scalebar_ocr.py:72-364`` without EasyOCR (out of scope).  Three cases:

    * ``scale_bar.um_per_pixel`` (or ``DEEPEMIA_UM_PER_PIXEL``): the calibration is configured, nothing is detected;
    * ``scale_bar.label`` (+ optional ``scale_bar.text_center``): the label the OCR would have read is configured and the
      bar's LINE is found as the reference finds it -- Canny 50/150, HoughLinesP, horizontal / margin / brightness /
      proximity filters, collinear merge, longest survivor (``deepemia_amd/utils/scalebar.py``);
      ``um_pix = label / length``; thresholds from ``scalebar_thresholds`` with the reference's override rule (:102-115);
      ``draw_debug`` draws ROI, candidates and the selected line into ``image`` (for ``<img>_scalebar_debug.png``);
    * neither: the reference's own fallback when the OCR finds nothing, ``("0", 1.0)`` (:362-364), logged once."""
    global _scale_bar_warned
    cfg = _scale_bar_settings(dataset_name)
    um = os.environ.get("DEEPEMIA_UM_PER_PIXEL", cfg.get("um_per_pixel"))
    if um is not None:
        um = float(um)
        if not (um > 0.0 and math.isfinite(um)):
            raise ValueError(f"um_per_pixel must be a positive number, got {um!r}")
        # the `Detected scale bar` column holds the label's DIGITS in every mode (scalebar_ocr.py:163-166 keeps the digits of
        # the OCR text; the line-detection branch below does the same)
        psum = "".join(ch for ch in str(cfg.get("label", "0")) if ch.isdigit()) or "0"
        if not _scale_bar_warned:
            system_logger.info(f"Scale bar: using the configured calibration {um} um/pixel (label {psum!r}); no OCR")
            _scale_bar_warned = True
        return psum, um
    if cfg.get("label") is not None and image is not None:
        from ..utils.scalebar import find_scale_bar_line

        if roi_config is None:
            roi_config = get_scalebar_roi_for_dataset(dataset_name)
        merge_gap, min_line_length, edge_margin_factor = 15, 30, 0.1
        try:
            th = get_config(dataset_name=dataset_name).get("scalebar_thresholds", {}) or {}
            if "intensity" in th and intensity_threshold == 200:
                intensity_threshold = th["intensity"]
            if "proximity" in th and proximity_threshold == 50:
                proximity_threshold = th["proximity"]
            merge_gap = th.get("merge_gap", 15)
            min_line_length = th.get("min_line_length", 30)
            edge_margin_factor = th.get("edge_margin_factor", 0.1)
        except Exception as e:
            system_logger.warning(f"Could not load thresholds from config: {e}")
        res = find_scale_bar_line(image, roi_config, cfg["label"], cfg.get("text_center"), intensity_threshold, proximity_threshold,
                                  merge_gap, min_line_length, edge_margin_factor)
        if draw_debug:
            _draw_scale_bar_debug(image, res)
        if res["line"] is not None:
            system_logger.info(f"Detected scale bar: {res['psum']} units, {res['length']:.2f} pixels, {res['um_pix']:.4f} units/pixel")
        else:
            system_logger.warning("No scale bar line detected near the configured label position.")
        return res["psum"], res["um_pix"]
    if not _scale_bar_warned:
        system_logger.warning("Scale bar not read (EasyOCR is out of scope): um_pix = 1.0, every length / area column of "
                              "measurements_results.csv is in PIXELS; set scale_bar.label, scale_bar.um_per_pixel or DEEPEMIA_UM_PER_PIXEL")
        _scale_bar_warned = True
    return "0", 1.0


def _draw_scale_bar_debug(image: np.ndarray, res: dict) -> None:
    """The debug drawing of ``scalebar_ocr.py:139-143,289-300,351-356`` (BGR colours as there: ROI green, candidates cyan, near
    the ROI edge grey, the selected line red) into ``image`` in place; Pillow rasterises, not OpenCV (no pixel parity)."""
    from PIL import Image, ImageDraw

    pil = Image.fromarray(np.ascontiguousarray(image[..., ::-1]))
    d = ImageDraw.Draw(pil)
    x0, y0, x1, y1 = res["roi"]
    d.rectangle([x0, y0, x1 - 1, y1 - 1], outline=(0, 255, 0), width=2)
    for k, sg in enumerate(res["segments"]):
        col = (128, 128, 128) if sg["near_edge"] else (0, 255, 255)
        d.line([x0 + sg["x1"], y0 + sg["y1"], x0 + sg["x2"], y0 + sg["y2"]], fill=col, width=1)
        d.text((x0 + sg["x1"], y0 + sg["y1"] - 12), f"M{k}: {sg['length']:.0f}px, I:{sg['intensity']:.0f}, D:{sg['dist_to_text']:.0f}", fill=col)
    if res["line"] is not None:
        d.line(list(res["line"]), fill=(255, 0, 0), width=3)
        d.text((res["line"][0], res["line"][1] - 24), f"SELECTED: {res['length']:.0f}px", fill=(255, 0, 0))
    else:
        d.text((x0, y0 + 30), "SCALE BAR DETECTION FAILED", fill=(255, 0, 0))
    image[...] = np.asarray(pil)[..., ::-1]


def calculate_image_quality_score(image: np.ndarray) -> float:
    """``inference.py:256-285`` with OpenCV's BGR->gray fixed-point weights (B 1868, G 9617, R 4899 >> 14)."""
    if image.ndim == 3:
        b, g, r = (image[:, :, i].astype(np.int64) for i in range(3))
        gray = ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)
    else:
        gray = image
    return float(np.clip(0.4 * (np.mean(gray) / 255.0) + 0.6 * (np.std(gray) / 128.0), 0.0, 1.0))


def get_confidence_threshold(image, target_class, small_classes, global_config) -> float:
    """``inference.py:288-362``: reads the GLOBAL config (not the dataset override), as the reference does."""
    inf = global_config.get("inference_settings", {})
    ccfg = inf.get("class_specific_settings", {}).get(f"class_{target_class}", {})
    base = ccfg.get("confidence_threshold", 0.3 if target_class in small_classes else 0.5)
    if inf.get("confidence_mode", "auto") == "manual":
        return base
    q = calculate_image_quality_score(image)
    if q < 0.3:
        return base * 0.7
    if q < 0.5:
        return base * 0.85
    return base


def determine_small_classes(class_avg_sizes: Dict[int, float], threshold_percentile=50) -> set:
    """``inference.py:1709-1736``."""
    if not class_avg_sizes:
        return set()
    thr = np.percentile(list(class_avg_sizes.values()), threshold_percentile)
    return {c for c, s in class_avg_sizes.items() if s <= thr}


class _Detections:
    """One predictor call's result, device-resident (masks packed) + small host tables."""

    def __init__(self, packed: torch.Tensor, scores: np.ndarray, classes: np.ndarray, hw: Tuple[int, int],
                 bbox: Optional[torch.Tensor] = None):
        self.packed, self.scores, self.classes, self.hw = packed, scores, classes, hw
        self.bbox = bbox      # [n, 4] int32 on the device: the paste boxes (supersets of the tight mask boxes) or None
        # the forward's whole mask / box tables ([B * D, H, W/32], [B * D, 4]) and this call's rows in them: lets a
        # batched stage gather the masks of MANY tiles with one launch
        self.base: Optional[torch.Tensor] = None
        self.base_bbox: Optional[torch.Tensor] = None
        self.base_idx: Optional[np.ndarray] = None
        self.owner = None     # (engine, RawDetections) when `base` are a replayed forward's own planes read in place


class _PassTables:
    """Pixel counts and tight boxes of a finished class pass (host copies): what the later stages read of it."""

    def __init__(self, area: np.ndarray, bbox: np.ndarray):
        self.area, self.bbox = area, bbox


class EmptyEnsembleTypeError(ValueError):
    """Reference behaviour N4: ``np.array([]) + [masks...]`` raises and the image is skipped."""


class PeerImageFailure(RuntimeError):
    """Multi-GPU: some rank's local passes of an image failed; raised on EVERY rank after the image's exchange, so all
    ranks skip the image together (reference semantics per image: log, skip, continue -- inference.py:928-931)."""


class OutputWriteError(RuntimeError):
    """Rank 0's host-side output section (CSV / overlay writers, after the last collective) failed -- the one kind of
    failure ``main.py`` reports to the other ranks through its status all-reduce instead of ending the process."""


class InferencePipeline:
    def __init__(self, predictors: Sequence, dataset_name: str, inf_settings: dict, global_config: dict):
        self.predictors = list(predictors)
        self.dataset_name = dataset_name
        self.inf = inf_settings
        self.gcfg = global_config
        self.dev = self.predictors[0].engine.device
        import threading
        self._tls = threading.local()        # every host thread that walks images gets a MaskOps of its own (frame width, pools, upload stream)
        self._cache_lock = threading.RLock()
        l4 = global_config.get("l4_performance_optimizations", {})
        self.parallel_mask_processing = l4.get("enable_parallel_mask_processing", True)
        gens = global_config.get("inference_settings", {}).get("ensemble_settings", {})
        # N3: weights always come from the import-time GLOBAL config, dict order R50, R101
        self.ensemble_weights = list(gens.get("weights", {"R50": 0.6, "R101": 0.4}).values())
        self.class_specific_settings = inf_settings.get("class_specific_settings", {})
        # f4, flagged NON-parity: `merge_mode: soft_nms` replaces the 0.4 hard merge of full-image + tile results
        self.merge_mode = str(inf_settings.get("merge_mode", "smart"))
        snm = inf_settings.get("soft_nms", {}) or {}
        self.soft_nms_sigma = float(snm.get("sigma", 0.5))
        self.soft_nms_score_threshold = float(snm.get("score_threshold", 0.001))
        if self.merge_mode not in ("smart", "soft_nms"):
            raise ValueError(f"inference_settings.merge_mode must be 'smart' or 'soft_nms', got {self.merge_mode!r}")
        if self.merge_mode == "soft_nms":
            system_logger.warning("merge_mode: soft_nms -- NON-PARITY mode: the reference has no soft-NMS (hard greedy dedup only)")
        # f4, flagged NON-parity: `multiscale_settings.enabled: true` makes the classes whose class_specific_settings carry
        # the reference's own (never read there) `use_multiscale: true` take their full-image pass through
        # run_adaptive_multiscale_inference (reference inference.py:1816-2064, dead code there: SURVEY N2)
        msc = inf_settings.get("multiscale_settings", {}) or {}
        self.multiscale_enabled = bool(msc.get("enabled", False))
        if self.multiscale_enabled:
            system_logger.warning("multiscale_settings.enabled -- NON-PARITY mode: the reference never reaches its multi-scale code")
        self._cache: Dict[Tuple[int, str], List[_Detections]] = {}
        # images per batched forward: l4_performance_optimizations.forward_batch_size / DEEPEMIA_FORWARD_BATCH (default 48: the res3-res5 layers fill the 256 CUs for several rounds, 26 GiB of activations; 16:
        # fills 256 CUs on the 50^2 feature maps, 8.6 GiB of activations per 2048^2-tile batch)
        self.forward_batch = max(1, int(os.environ.get("DEEPEMIA_FORWARD_BATCH", l4.get("forward_batch_size", 48))))
        self.use_graphs = os.environ.get("DEEPEMIA_GRAPHS", "1") == "1"     # hipGraph replay of repeated forward shapes
        self.graph_after = 2                                                   # ... from their second occurrence on
        self._shape_seen: Dict[tuple, int] = {}
        self.graph_slots, self.clone_graph_outputs = 1, True                  # (see forward_async)
        # tile-batch loops that consume a batch's masks before the next batch is processed (bench.py) keep the intermediate
        # and final mask sets in plane pools that stay zero outside per-slot boxes: a gather writes boxes, not 512-KiB planes.
        # The planes a call returns are then views of a pool, valid until the next call.
        self.pooled_planes = False
        self.last_batch_stats = None
        self.forward_calls = 0
        self.d2h_waits = 0                       # device-to-host waits of the post-processing (finish_forward + process_tile_batch)
        import torch.distributed as dist
        self.rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.exchange = parallel.ExchangeState()     # this job's agreed capacities: every rank builds its pipeline at the same point

    @property
    def ops(self) -> MaskOps:
        o = getattr(self._tls, "ops", None)
        if o is None:
            o = self._tls.ops = MaskOps(str(self.dev))
        return o

    # ------------------------------------------------------------------ predictor plumbing
    def forward_async(self, model_idx: int, images: torch.Tensor):
        """Enqueue the batched forward(s) on the CURRENT stream and return without synchronising: the handle
        (raw device results + an event) is turned into host tables later by :meth:`finish_forward`, so a caller
        can overlap this batch's network with the previous batch's post-processing on another stream."""
        pred = self.predictors[model_idx]
        raws = []
        step = self.forward_batch
        for b0 in range(0, images.shape[0], step):
            chunk = images[b0:b0 + step].contiguous()
            # hipGraph replay (two alternating output sets per shape: the previous batch may still be read by the
            # post-processing stream) where the caller asked for it; eager otherwise
            key = (model_idx,) + tuple(int(d) for d in chunk.shape[:3])
            self._shape_seen[key] = self._shape_seen.get(key, 0) + 1
            # a batch shape that comes back (the tiles of the next image, the next batch of a job) is worth a capture; a
            # one-off shape runs eagerly.  Same kernels in the same order either way: identical results.
            graphed = self.use_graphs and self._shape_seen[key] >= self.graph_after
            if graphed:
                # a replay writes into the graph's own output buffers, which a later replay of this shape overwrites:
                # the pipeline keeps detections for a whole image (and the first images' for the small-class statistics),
                # so it takes its own copy (one device-to-device copy of the packed masks, ~0.3 ms per 16 tiles).  A caller
                # that consumes a batch's detections before the forward after next (the tile-batch loop of bench.py:
                # `graph_slots = 2`, `clone_graph_outputs = False`) reads the graph's buffers in place.
                r = pred.engine.forward_graphed(chunk, slots=self.graph_slots)
                if self.clone_graph_outputs:
                    if r.bbox is not None:       # the planes are zero outside the paste boxes: read the boxes, write each plane once
                        flat = r.packed.view((-1,) + tuple(r.packed.shape[2:]))
                        idx = torch.arange(flat.shape[0], dtype=torch.int64, device=self.dev)
                        packed = self.ops.gather_regions(flat, idx, r.bbox.view(-1, 4)).view_as(r.packed)
                    else:
                        packed = r.packed.clone()
                    r = type(r)(r.boxes.clone(), r.scores.clone(), r.classes.clone(), r.valid.clone(), r.count.clone(), packed,
                                r.height, r.width, None if r.bbox is None else r.bbox.clone())
                elif getattr(r, "_slot", None) is not None:
                    # the graph's own planes, read in place: consumers may only READ them (they stay zero outside the boxes the
                    # next replay knows about), and they tell the engine when their last read is enqueued (release_outputs)
                    self.ops.protect(r.packed)
                    r._engine = pred.engine
                raws.append(r)
            else:
                raws.append(pred.engine.forward(chunk))
            self.forward_calls += 1
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        return raws, ev, (int(images.shape[1]), int(images.shape[2]))

    def finish_forward(self, handle) -> List[_Detections]:
        return self.finish_forwards([handle])[0]

    def finish_forwards(self, handles) -> List[List[_Detections]]:
        """The small tables of one or MANY enqueued forwards (an ensemble's models) in ONE device-to-host copy -- the wait
        for the forwards themselves."""
        cur = torch.cuda.current_stream(self.dev)
        parts, shapes = [], []
        for raws, ev, _ in handles:
            cur.wait_event(ev)
            for raw in raws:
                # the forward allocated these on ITS stream; they are consumed on this one: tell the caching allocator, or a
                # later forward could be handed the blocks while kernels of this stream still read them
                for t in (raw.packed, raw.bbox, raw.scores, raw.classes, raw.valid, raw.count, raw.boxes):
                    if t is not None:
                        t.record_stream(cur)
                parts += [raw.count.reshape(-1), raw.valid.reshape(-1).to(torch.int32), raw.scores.reshape(-1).view(torch.int32),
                          raw.classes.reshape(-1)]
                shapes.append((int(raw.count.shape[0]), int(raw.scores.shape[1])))
        tab_all = torch.cat(parts).cpu().numpy()           # ONE copy for all forwards' tables
        self.d2h_waits += 1
        outs: List[List[_Detections]] = []
        pos, si = 0, 0
        for raws, _, (h, w) in handles:
            out: List[_Detections] = []
            for raw in raws:
                nb, nd = shapes[si]
                si += 1
                tab = tab_all[pos:pos + nb + 3 * nb * nd]
                pos += nb + 3 * nb * nd
                counts = tab[:nb]
                valid = tab[nb:nb + nb * nd].reshape(nb, nd).astype(bool)
                scores = tab[nb + nb * nd:nb + 2 * nb * nd].view(np.float32).reshape(nb, nd)
                classes = tab[nb + 2 * nb * nd:].reshape(nb, nd).astype(np.int64)
                # ONE view object per forward: detections of the same forward are recognised by `base is base`
                base = None if raw.bbox is None else raw.packed.view((-1,) + tuple(raw.packed.shape[2:]))
                base_bbox = None if raw.bbox is None else raw.bbox.view(-1, 4)
                for b in range(raw.count.shape[0]):
                    n = int(counts[b])
                    sel = np.nonzero(valid[b, :n])[0]
                    if len(sel) == n:
                        packed, bbox = raw.packed[b, :n], (None if raw.bbox is None else raw.bbox[b, :n])
                    else:
                        si_ = torch.from_numpy(sel).to(self.dev)
                        packed, bbox = raw.packed[b, si_], (None if raw.bbox is None else raw.bbox[b, si_])
                    det = _Detections(packed, scores[b, :n][sel], classes[b, :n][sel], (h, w), bbox)
                    if base is not None:
                        det.base, det.base_bbox = base, base_bbox
                        det.base_idx = (b * int(raw.packed.shape[1]) + sel).astype(np.int64)
                    if getattr(raw, "_engine", None) is not None:
                        det.owner = (raw._engine, raw)
                    out.append(det)
            outs.append(out)
        return outs

    def _predict_batches(self, model_ids: Sequence[int], key: str, images: torch.Tensor) -> List[List[_Detections]]:
        """:meth:`_predict_batch` for several models at once: the forwards that are not cached yet are all enqueued first and
        their tables come over in ONE wait."""
        with self._cache_lock:
            todo = [m for m in model_ids if (m, key) not in self._cache]
            if todo:
                for m, d in zip(todo, self.finish_forwards([self.forward_async(m, images) for m in todo])):
                    self._cache[(m, key)] = d
            return [self._cache[(m, key)] for m in model_ids]

    def _release_forward_outputs(self, dets: Sequence[_Detections]) -> None:
        """Every read of the forwards' own output planes behind ``dets`` is enqueued on the current stream: the replays that
        overwrite those planes may run once this point of the stream is reached (``MaskRCNNEngine.release_outputs``).
        Detections that alias a forward's planes must not be found in the cache afterwards (a later hit would read planes a
        later replay has overwritten): their cache entries go with the release."""
        seen, released = set(), set()
        for det in dets:
            if det.owner is not None:
                released.add(id(det))
                if id(det.owner[1]) not in seen:
                    seen.add(id(det.owner[1]))
                    det.owner[0].release_outputs(det.owner[1])
        if released:
            for ck in [ck for ck, v in self._cache.items() if any(id(d) in released for d in v)]:
                del self._cache[ck]

    def _predict_batch(self, model_idx: int, key: str, images: torch.Tensor) -> List[_Detections]:
        """Forward a batch of equally sized images once per (model, key); every class reuses it."""
        ck = (model_idx, key)
        with self._cache_lock:              # (a miss runs the engine: one thread at a time)
            if ck not in self._cache:
                self._cache[ck] = self.finish_forward(self.forward_async(model_idx, images))
            return self._cache[ck]

    def clear_cache(self) -> None:
        self._cache.clear()

    # ------------------------------------------------------------------ a12
    def _greedy_dedup(self, packed: torch.Tensor, thr: float) -> List[int]:
        """``inference.py:1451-1459``: keep mask i unless IoU(mask_i, kept_j) > thr for a kept j."""
        n = int(packed.shape[0])
        alg = DeviceMaskAlgebra(self.ops, packed)
        alg.prefetch_overlapping_pairs()
        return self._greedy_keep(alg, range(n), thr)

    # ------------------------------------------------------------------ a6 + a9 + a11 + a12
    def _single_model_class_pass(self, det: _Detections, target_class: int, small_classes, confidence_threshold, iou_threshold):
        sel = np.nonzero((det.classes == target_class))[0]
        sel = sel[det.scores[sel] >= confidence_threshold]
        if len(sel) == 0:
            return None, [], []
        scores = det.scores[sel]
        si = torch.from_numpy(sel).to(self.dev)
        masks = det.packed[si].contiguous()
        hint = None if det.bbox is None else det.bbox[si].contiguous()
        is_small = target_class in small_classes
        ccfg = self.class_specific_settings.get(f"class_{target_class}", {})
        min_size = ccfg.get("min_size", 5 if is_small else 25)
        processed = postprocess_masks_device(self.ops, masks, scores, min_crys_size=min_size, bbox=hint)
        if processed is None or processed.shape[0] == 0:
            return None, [], []
        if processed.shape[0] > 2 and self.parallel_mask_processing:
            processed = process_masks_device(self.ops, processed)
        thr = 0.5 if is_small else iou_threshold
        keep = self._greedy_dedup(processed, thr)
        kp = processed[torch.tensor(keep, dtype=torch.long, device=self.dev)].contiguous()
        return kp, [scores[i] for i in keep], [target_class] * len(keep)

    # ------------------------------------------------------------------ a10 + a14 (ensemble branch)
    def _ensemble_class_pass(self, dets: Sequence[_Detections], target_class, small_classes, conf_threshold, iou_threshold):
        all_masks, all_scores = [], []
        is_small = target_class in small_classes
        hw = dets[0].hw
        for det, weight in zip(dets, self.ensemble_weights):
            if len(det.scores) == 0:
                continue
            sel = np.nonzero((det.classes == target_class) & (det.scores >= conf_threshold))[0]
            if len(sel) == 0:
                continue
            si = torch.from_numpy(sel).to(self.dev)
            masks = det.packed[si].contiguous()
            hint = None if det.bbox is None else det.bbox[si].contiguous()
            kept_masks, kept_idx = postprocess_masks_universal_device(self.ops, masks, hw, is_small, bbox=hint)
            if len(kept_idx) == 0:
                continue
            all_masks.append(kept_masks)
            all_scores.extend(float(det.scores[sel][i]) * weight for i in kept_idx)
        if not all_masks:
            return "EMPTY_NDARRAY", [], []   # the reference returns three empty ndarrays here (N4)
        packed = torch.cat(all_masks, dim=0)
        return self.deduplicate_masks_smart(packed, all_scores, [target_class] * len(all_scores), iou_threshold)

    # ------------------------------------------------------------------ a14
    def deduplicate_masks_smart(self, packed: Optional[torch.Tensor], scores: Sequence[float], classes: Sequence[int],
                                iou_threshold: float = 0.4, with_tables: bool = False, all_pairs: bool = False):
        """``inference.py:2552-2677`` bug-for-bug (N6); see oracle/postproc_ref.py for the dense twin.

        ONE device-to-host wait: pixel counts and tight boxes, the contour trace (first-contour perimeters for the compactness
        rule) and the same-class pair counts (``demia_mask_pair_matrix`` over the class runs) are enqueued back to back and
        fetched together; step 2 is one native call (``demia_host_dedup_smart``, N6 literally).  The CLI loop calls this
        three times per image (two 0.4 merges + the 0.7 cross-class pass): the host-loop version below -- five waits and
        interpreted greedy loops, kept as its checker and for class lists that are not contiguous runs -- was 12 ms of a
        70-ms image.  ``with_tables``: also return (area, bbox) of the kept masks (host arrays)."""
        if packed is None or packed.shape[0] == 0:
            return (None, [], [], None) if with_tables else (None, [], [])
        out = self.deduplicate_masks_smart_segments(packed, scores, classes, [(0, int(packed.shape[0]))], iou_threshold, all_pairs=all_pairs)[0]
        return out if with_tables else out[:3]

    def deduplicate_masks_smart_segments(self, packed: torch.Tensor, scores: Sequence[float], classes: Sequence[int],
                                         segments: Sequence[Tuple[int, int]], iou_threshold: float, all_pairs: bool = False):
        """:meth:`deduplicate_masks_smart` for SEVERAL independent calls at once -- ``segments`` = [start, end) of each call's masks in
        ``packed`` (the two per-class 0.4 merges of an image) -- with ONE device-to-host wait for all of them: one reduction, one contour
        trace and one pair matrix over the (segment, class) runs, one native call with a "tile" per segment (every segment is
        filtered with its own local indices, N6 included: exactly what separate calls do).  Returns per segment
        (masks, scores, classes, (area, bbox, inter)) -- (None, [], [], None) for an empty result.  ``all_pairs``: the pair matrix
        covers EVERY pair of a segment, not only the same-class ones the filter itself reads, and ``inter`` is the survivors'
        symmetric intersection matrix (else None): the spatial constraints that follow the 0.7 pass of an image then need no
        device work of their own (``DeviceMaskAlgebra.preload``)."""
        empty = (None, [], [], None)
        n = int(packed.shape[0])
        if n == 0:
            return [empty for _ in segments]
        cl = np.asarray([int(c) for c in classes], dtype=np.int32)
        run_first = np.zeros(n, dtype=np.int32)
        run_count = np.zeros(n, dtype=np.int32)
        general = False
        for s0, s1 in segments:
            if s1 <= s0:
                continue
            seg = cl[s0:s1]
            change = np.nonzero(np.diff(seg))[0] + 1
            starts = np.concatenate(([0], change))
            ends = np.concatenate((change, [s1 - s0]))
            if len(set(seg[starts].tolist())) != len(starts):      # a class in two separate runs: the general host-loop version
                general = True
                break
            for a_, b_ in zip(starts, ends):
                run_first[s0 + a_:s0 + b_] = s0 + a_
                run_count[s0 + a_:s0 + b_] = b_ - a_
            if all_pairs:
                run_first[s0:s1] = s0
                run_count[s0:s1] = s1 - s0
        if general:
            out = []
            for s0, s1 in segments:
                if s1 <= s0:
                    out.append(empty)
                    continue
                m, s_, c_ = self.deduplicate_masks_smart_hostloops(packed[s0:s1].contiguous(), list(scores[s0:s1]), list(classes[s0:s1]), iou_threshold)
                if m is None:
                    out.append(empty)
                else:
                    a_, b_ = self.ops.area_bbox(m)
                    out.append((m, s_, c_, (a_.cpu().numpy().astype(np.int64), b_.cpu().numpy().astype(np.int64), None)))
            return out
        ops = self.ops
        packed = packed.contiguous()
        area_d, bbox_d = ops.area_bbox(packed)
        cset = ops.trace(packed, max_contours=256, bbox=bbox_d, max_points=int(min(4096 * n + (1 << 16), 1 << 26)))
        ld = max(1, int(run_count.max()))
        I = ops.pair_matrix(packed, bbox_d, run_first, np.maximum(run_count, 1), None, ld)
        extra = [area_d.to(torch.int32), bbox_d, I]
        try:
            area_h, bbox_h, I_h = cset.fetch(extra=extra)
        except _L.HipKernelError as e:
            if "overflow" not in str(e):
                raise
            cset = ops.trace(packed, max_contours=256, bbox=bbox_d, total_area=int(area_d.sum().item()))
            area_h, bbox_h, I_h = cset.fetch(extra=extra)
        self.d2h_waits += 1
        area = np.ascontiguousarray(area_h, dtype=np.int64)
        bbox = np.ascontiguousarray(bbox_h.reshape(n, 4), dtype=np.int64)
        per0 = cset.first_contour_perimeter()
        ok = (bbox[:, 0] >= 0) & ~((per0 > 0) & ((4 * np.pi * area) / np.where(per0 > 0, per0, 1.0) ** 2 < 0.15))
        per_seg = [np.nonzero(ok[s0:s1])[0] + s0 for s0, s1 in segments]
        items = np.ascontiguousarray(np.concatenate(per_seg) if per_seg else np.zeros(0), dtype=np.int32)
        if len(items) == 0:
            return [empty for _ in segments]
        T = len(segments)
        tile_off = np.concatenate(([0], np.cumsum([len(k) for k in per_seg]))).astype(np.int32)
        sc_items = np.ascontiguousarray(np.asarray([scores[i] for i in items], dtype=np.float64))
        cl_items = np.ascontiguousarray(cl[items])
        keep_out = np.zeros(len(items), dtype=np.int32)
        keep_cnt = np.zeros(T, dtype=np.int32)
        I_c = np.ascontiguousarray(I_h.reshape(n, ld), dtype=np.int32)
        _L.check(ops.lib.demia_host_dedup_smart(I_c.ctypes.data, ld, run_first.ctypes.data, area.ctypes.data, bbox.ctypes.data,
                                                items.ctypes.data, sc_items.ctypes.data, cl_items.ctypes.data, tile_off.ctypes.data, T,
                                                float(iou_threshold), keep_out.ctypes.data, keep_cnt.ctypes.data), "demia_host_dedup_smart")
        gls = [per_seg[t][keep_out[tile_off[t]:tile_off[t] + keep_cnt[t]]] for t in range(T)]
        flat = np.concatenate(gls) if gls else np.zeros(0, dtype=np.int64)
        if len(flat) == 0:
            return [empty for _ in segments]
        sel = ops.upload(flat.astype(np.int64))
        kept_all = ops.gather_regions(packed, sel, bbox_d.index_select(0, sel))
        out, pos = [], 0
        for gl in gls:
            k = len(gl)
            if k == 0:
                out.append(empty)
            else:
                inter = None
                if all_pairs:
                    s0 = int(run_first[gl[0]])
                    lo, hi = np.minimum(gl[:, None], gl[None, :]), np.maximum(gl[:, None], gl[None, :])
                    inter = I_c[lo, np.minimum(hi - s0, ld - 1)].astype(np.int64)      # row i, column j - first[i] holds |i & j| for j > i
                    d_ = np.arange(k)
                    inter[d_, d_] = area[gl]
                out.append((kept_all[pos:pos + k], [scores[i] for i in gl], [classes[i] for i in gl], (area[gl], bbox[gl], inter)))
            pos += k
        return out

    def deduplicate_masks_smart_hostloops(self, packed: Optional[torch.Tensor], scores: Sequence[float], classes: Sequence[int],
                                          iou_threshold: float = 0.4):
        """The host-loop version of :meth:`deduplicate_masks_smart` (general class order; its checker)."""
        if packed is None or packed.shape[0] == 0:
            return None, [], []
        alg = DeviceMaskAlgebra(self.ops, packed)
        cont = self.ops.contours(packed, max_contours=256, measure=False)
        keep0 = []
        for idx in range(alg.n):
            if alg.bbox[idx, 0] < 0:
                continue  # empty mask
            if len(cont[idx]) > 0:
                per = cont[idx][0]["perimeter"]
                if per > 0 and (4 * np.pi * int(alg.area[idx])) / (per ** 2) < 0.15:
                    continue
            keep0.append(idx)
        if not keep0:
            return None, [], []
        scores = [scores[i] for i in keep0]
        classes = [classes[i] for i in keep0]
        # stored as (y_min, y_max, x_min, x_max) ... (inference.py:2635)
        bb = [(int(alg.bbox[i, 0]), int(alg.bbox[i, 2]), int(alg.bbox[i, 1]), int(alg.bbox[i, 3])) for i in keep0]

        def overlap_literal(b1, b2):  # ... unpacked as (y_min, x_min, y_max, x_max) (inference.py:2685)
            y1_min, x1_min, y1_max, x1_max = b1
            y2_min, x2_min, y2_max, x2_max = b2
            if x1_max < x2_min or x2_max < x1_min:
                return False
            if y1_max < y2_min or y2_max < y1_min:
                return False
            return True

        alg.prefetch_overlapping_pairs([[keep0[i] for i in range(len(keep0)) if classes[i] == c] for c in set(classes)])
        keep = self._dedup_smart_order(alg, keep0, scores, classes, bb, iou_threshold)
        sel = torch.tensor([keep0[i] for i in keep], dtype=torch.long, device=self.dev)
        return packed[sel].contiguous(), [scores[i] for i in keep], [classes[i] for i in keep]

    # ------------------------------------------------------------------ a1 + a2
    def _make_tiles(self, image_dev: torch.Tensor, tile_size: int, overlap_ratio: float):
        h, w = int(image_dev.shape[0]), int(image_dev.shape[1])
        stride = int(tile_size * (1 - overlap_ratio))
        offs = [(x, y) for y in range(0, h, stride) for x in range(0, w, stride)]
        tiles = torch.zeros((len(offs), tile_size, tile_size, 3), dtype=torch.uint8, device=self.dev)
        for i, (x, y) in enumerate(offs):
            ye, xe = min(y + tile_size, h), min(x + tile_size, w)
            tiles[i, : ye - y, : xe - x] = image_dev[y:ye, x:xe]
        return tiles, offs

    @staticmethod
    def _tile_offsets(h: int, w: int, tile_size: int, overlap_ratio: float):
        stride = int(tile_size * (1 - overlap_ratio))
        return [(x, y) for y in range(0, h, stride) for x in range(0, w, stride)]

    def _tiles_key(self, image_key: str, tile_size, overlap_ratio, upscale_factor) -> str:
        return f"{image_key}|tiles{tile_size}/{overlap_ratio}/{upscale_factor}/{self.rank}of{self.world}"

    def _my_tiles(self, image_dev: torch.Tensor, tile_size: int, overlap_ratio: float, upscale_factor: float) -> Optional[torch.Tensor]:
        """THIS rank's tiles of an image (a1, tile ``t % world == rank``), upscaled (a2) -- the input of the tile forward."""
        tiles, offs = self._make_tiles(image_dev, tile_size, overlap_ratio)
        mine = parallel.shard_indices(len(offs), self.rank, self.world)
        if not mine:
            return None
        my_tiles = tiles if len(mine) == len(offs) else tiles[torch.tensor(mine, dtype=torch.long, device=self.dev)]
        uh = uw = int(tile_size * upscale_factor)
        if (uh, uw) != (tile_size, tile_size):
            my_tiles = self.predictors[0].engine.resize_linear_u8(my_tiles, uh, uw)
        return my_tiles

    # ------------------------------------------------------------------ forwards batched ACROSS images (the CLI loop)
    def prefetch_images(self, items: Sequence[Tuple[str, torch.Tensor]], model_ids: Sequence[int], tile_size: int, overlap_ratio: float,
                        upscale_factor: float):
        """Enqueue, on the CURRENT stream, the standard forwards of a GROUP of images -- every image's full-image pass
        (rank 0; ``inference.py:789``) and this rank's tiles (``inference.py:2365-2449``) -- batched ACROSS the images: the
        full frames of equally sized images go through the network as one batch, and so do all their tiles (a folder of 2048^2
        micrographs with nine tiles each: 45 tiles per forward instead of nine; the res3-res5 layers fill the 256 CUs for several
        rounds only from ~32 tiles up).  Returns a handle for :meth:`finish_prefetch`, which files the results in the cache
        under exactly the keys the per-image passes ask for -- they then run unchanged and find every forward done.  Nothing
        is waited for here: the caller enqueues the NEXT group before it post-processes the current one.
        Results are bit-identical to image-by-image forwards: every stage of the network is batch-invariant (per-image scale
        groups; ``test_f16x2_forward_is_batch_invariant``)."""
        plan = []            # (model, [(cache key, images in the batch)], handle)
        by_shape: Dict[tuple, list] = {}
        for name, img in items:
            by_shape.setdefault(tuple(int(d) for d in img.shape[:2]), []).append((name, img))
        for (h, w), group in by_shape.items():
            full_keys = [(name + "|full", 1) for name, _ in group] if self.rank == 0 else []
            full = torch.stack([img for _, img in group]) if full_keys else None
            tkeys, tparts = [], []
            for name, img in group:
                t = self._my_tiles(img, tile_size, overlap_ratio, upscale_factor)
                if t is not None:
                    tkeys.append((self._tiles_key(name, tile_size, overlap_ratio, upscale_factor), int(t.shape[0])))
                    tparts.append(t)
            tiles = (tparts[0] if len(tparts) == 1 else torch.cat(tparts)) if tparts else None
            for m in model_ids:
                for keys, batch in ((full_keys, full), (tkeys, tiles)):
                    todo = [(k, n) for k, n in keys if (m, k) not in self._cache]
                    if batch is None or not todo:
                        continue
                    if len(todo) != len(keys):      # (some already cached: forward the missing ones only)
                        pos, sel = 0, []
                        for k, n in keys:
                            if (m, k) not in self._cache:
                                sel.extend(range(pos, pos + n))
                            pos += n
                        batch_m = batch[torch.tensor(sel, dtype=torch.long, device=self.dev)]
                    else:
                        batch_m = batch
                    plan.append((m, todo, self.forward_async(m, batch_m)))
        return plan

    def finish_prefetch(self, plan) -> None:
        """ONE device-to-host wait for the tables of all forwards of a :meth:`prefetch_images` group; the per-image detection
        lists go into the cache."""
        if not plan:
            return
        for (m, keys, _), dets in zip(plan, self.finish_forwards([h for _, _, h in plan])):
            pos = 0
            for k, n in keys:
                self._cache[(m, k)] = dets[pos:pos + n]
                pos += n

    def drop_cached(self, image_key: str) -> None:
        """Forget the cached forwards of ONE image (the CLI loop calls it when the image is done; forwards of the images
        ahead stay)."""
        with self._cache_lock:
            for ck in [ck for ck in self._cache if ck[1] == image_key or ck[1].startswith(image_key + "|")]:
                del self._cache[ck]

    # ------------------------------------------------------------------ a13 + the tile pipeline
    def _tile_pipeline_local(self, model_ids: Sequence[int], image_key: str, image_dev: torch.Tensor, target_class,
                             small_classes, confidence_threshold, tile_size=512, overlap_ratio=0.1, upscale_factor=2.0,
                             iou_threshold=0.7, edge_filter_enabled=True):
        """``inference.py:2299-2460`` for one class, THIS RANK'S share: the full-image pass (rank 0) and the tiles
        ``t % world == rank``, each through the class pass, the nearest resize back to tile scale, the edge filter and the
        paste into the global frame.  Returns ``(full_masks, full_scores, full_classes, tile_masks, tile_scores,
        tile_classes, tile_units)`` -- no communication here."""
        h, w = int(image_dev.shape[0]), int(image_dev.shape[1])
        self.ops.set_frame_width(w)
        ensemble = len(model_ids) > 1

        def class_pass(dets_per_model):
            if ensemble:
                return self._ensemble_class_pass(dets_per_model, target_class, small_classes, confidence_threshold, iou_threshold)
            # the batched pass over ONE image: everything enqueued, one wait, the greedy loop native (same keep list as
            # `_single_model_class_pass`, its tile-by-tile checker: test_batched_tile_pipeline_equals_tile_by_tile)
            big, res, tabs = self._single_class_pass_batched(dets_per_model[:1], target_class, small_classes, confidence_threshold, iou_threshold)
            kept, sc = res[0]
            if big is None or not kept:
                return None, [], []
            return self.ops.gather_regions(big, kept, tabs.bbox[kept]), list(sc), [target_class] * len(kept)

        rank, world = self.rank, self.world
        offs = self._tile_offsets(h, w, tile_size, overlap_ratio)
        uh, uw = int(tile_size * upscale_factor), int(tile_size * upscale_factor)
        # unit 0 = the full-image pass (rank 0), unit 1 + t = tile t (rank t % world): SURVEY.md section 8(e)
        mine = parallel.shard_indices(len(offs), rank, world)
        full_masks, full_scores, full_classes = None, [], []
        if rank == 0:
            if self.uses_multiscale(target_class):      # flagged NON-parity mode (f4): the full-image pass at several scales
                full_masks, full_scores, full_classes = self.run_adaptive_multiscale_inference(
                    list(model_ids), image_key, image_dev, target_class, confidence_threshold, small_classes, iou_threshold)
                self.ops.set_frame_width(w)
            else:
                full = [self._predict_batch(m, image_key + "|full", image_dev[None])[0] for m in model_ids]
                full_masks, full_scores, full_classes = class_pass(full)
        tile_masks, tile_scores, tile_classes, tile_units = [], [], [], []
        if mine:
            tkey = self._tiles_key(image_key, tile_size, overlap_ratio, upscale_factor)
            # (the tiles are cut only if some model's forward over them is not in the cache yet: `prefetch_images` has usually
            # run them already, batched with the tiles of the neighbouring images)
            my_tiles = None if all((m, tkey) in self._cache for m in model_ids) else self._my_tiles(image_dev, tile_size, overlap_ratio, upscale_factor)
            tile_dets = [self._predict_batch(m, tkey, my_tiles) for m in model_ids]
        edge = int(tile_size * overlap_ratio / 2)
        if mine:
            # a6 + a9..a12 (or a10 + a14) for ALL of this rank's tiles with one launch per kernel, then a13 likewise: one
            # nearest-resize launch to tile scale, one bbox reduction for the edge test, one paste into the global frame
            self.ops.set_frame_width(uw)           # the tile masks' own frame: its right border is pixel uw - 1
            if ensemble:
                big, res, pass_tabs = self._ensemble_class_pass_batched(tile_dets, target_class, small_classes, confidence_threshold, iou_threshold)
                pass_tabs = None          # (a DeviceMaskAlgebra there)
            else:
                big, res, pass_tabs = self._single_class_pass_batched(tile_dets[0], target_class, small_classes, confidence_threshold, iou_threshold)
            tm_, ts_, tc_, tu_ = self._place_tile_results(big, res, pass_tabs, mine, offs, tile_size, uh, uw, h, w, edge, edge_filter_enabled, target_class)
            tile_masks += tm_
            tile_scores += ts_
            tile_classes += tc_
            tile_units += tu_
            self.ops.set_frame_width(w)
        return full_masks, full_scores, full_classes, tile_masks, tile_scores, tile_classes, tile_units

    def _place_tile_results(self, big, res, pass_tabs, mine, offs, tile_size, uh, uw, h, w, edge, edge_filter_enabled, target_class):
        """a13 for one class: the kept masks of this rank's tiles (``res[k]`` = (indices into ``big``, scores) of tile ``mine[k]``) back at
        tile scale (nearest resize; the identity when the tiles were not upscaled), the edge filter on their boxes, the paste into the
        global frame.  Returns ([global masks], scores, classes, unit ids)."""
        tile_masks, tile_scores, tile_classes, tile_units = [], [], [], []
        src, xo, yo, un, sc_all = [], [], [], [], []
        for k, t in enumerate(mine):
            kept, sc = res[k]
            src.extend(kept)
            xo.extend([offs[t][0]] * len(kept))
            yo.extend([offs[t][1]] * len(kept))
            un.extend([1 + t] * len(kept))
            sc_all.extend(sc)
        if src:
            n = len(src)
            tabs_ = pass_tabs
            if (uh, uw) == (tile_size, tile_size) and tabs_ is not None:
                # tiles were not upscaled: the nearest resize back to tile scale is the identity, and the tight boxes the
                # class pass reduced are the edge filter's boxes -- one gather of the regions, no launch + wait for boxes
                bb = np.asarray(tabs_.bbox)[src]
                small = self.ops.gather_regions(big, src, bb)
                self.ops.set_frame_width(tile_size)
            else:
                tm = big[torch.tensor(src, dtype=torch.long, device=self.dev)].contiguous()
                small = self.ops.place_tiles(tm, [0] * n, [0] * n, tile_size, tile_size, tile_size, tile_size, src_w=uw)
                self.ops.set_frame_width(tile_size)
                bb = None
            keep = list(range(n))
            if edge_filter_enabled:
                if bb is None:
                    _, bb = self.ops.area_bbox(small)
                    bb = bb.cpu().numpy()
                keep = [i for i in range(n) if not (bb[i, 0] < 0 or bb[i, 0] < edge or bb[i, 2] > tile_size - edge
                                                    or bb[i, 1] < edge or bb[i, 3] > tile_size - edge)]
            if keep:
                sel = torch.tensor(keep, dtype=torch.long, device=self.dev)
                glob = self.ops.place_tiles(small[sel].contiguous(), [xo[i] for i in keep], [yo[i] for i in keep],
                                            tile_size, tile_size, h, w, src_w=tile_size)
                tile_masks.append(glob)
                tile_scores.extend(sc_all[i] for i in keep)
                tile_classes.extend([target_class] * len(keep))
                tile_units.extend(un[i] for i in keep)
        return tile_masks, tile_scores, tile_classes, tile_units

    # ------------------------------------------------------------------ f4: flagged NON-parity modes
    def soft_nms_merge(self, packed: Optional[torch.Tensor], scores: Sequence[float], classes: Sequence[int],
                       sigma: float = 0.5, score_threshold: float = 0.001):
        """Gaussian soft-NMS on MASK IoU (Bodla et al. 2017), per class -- a merge mode that exists in north_star but NOT
        in the reference (its live path only has hard NMS / hard greedy dedup), so it is opt-in
        (``inference_settings.merge_mode: soft_nms``) and claims no parity with the reference.  Repeatedly the remaining
        mask with the highest score is kept and every other remaining mask of its class has its score multiplied by
        ``exp(-IoU^2 / sigma)``; masks whose score falls below ``score_threshold`` are dropped.  Ties: lower index first.
        IoUs come from ONE pair-intersection launch over the box-overlapping pairs.  Returns (masks, scores, classes) in
        the order of selection, scores decayed."""
        if packed is None or packed.shape[0] == 0:
            return None, [], []
        alg = DeviceMaskAlgebra(self.ops, packed)
        n = alg.n
        sc = np.asarray(scores, dtype=np.float64).copy()
        cl = np.asarray(classes)
        alg.prefetch_overlapping_pairs([[i for i in range(n) if cl[i] == c] for c in sorted(set(cl.tolist()))])
        alive = sc >= score_threshold
        order = []
        while alive.any():
            cand = np.nonzero(alive)[0]
            i = int(cand[np.argmax(sc[cand])])           # first maximum = lower index on ties
            order.append(i)
            alive[i] = False
            for j in np.nonzero(alive & (cl == cl[i]))[0]:
                v = alg.iou(i, int(j))
                if v > 0.0:
                    sc[j] *= math.exp(-(v * v) / sigma)
                    if sc[j] < score_threshold:
                        alive[j] = False
        sel = torch.tensor(order, dtype=torch.long, device=self.dev)
        return packed[sel].contiguous(), [float(sc[i]) for i in order], [int(cl[i]) for i in order]

    def uses_multiscale(self, target_class: int) -> bool:
        return self.multiscale_enabled and bool(self.class_specific_settings.get(f"class_{target_class}", {}).get("use_multiscale", False))

    def run_adaptive_multiscale_inference(self, model_idx, image_key: str, image_dev: torch.Tensor, target_class: int,
                                          confidence_threshold: float = 0.3, small_classes=frozenset(), iou_threshold: float = 0.7):
        """The semantics of ``inference.py:1833-1984`` (``run_adaptive_multiscale_inference``) + ``:1986-2064``
        (``process_single_scale``) as a flagged NON-parity mode: that code is never reached from the reference's
        ``run_inference``, and its per-scale step is the iterative masking loop, which this build replaces by ONE
        class pass per scale (a6 + a9..a12).  Kept from the reference: the baseline scales 0.7 / 1.0 / 1.5, the 10 % benefit
        rule that unlocks 2.0 / 2.5 and 0.5 / 0.6, the 5 % low-yield stop, ``cv2.resize(INTER_LINEAR)`` of the image to
        ``int(h s) x int(w s)``, the scale-invariant minimum size (``max(3, int(A 5e-6))`` / ``max(25, int(A 1e-4))`` of the
        ORIGINAL area, times s^2), ``INTER_NEAREST`` back to the original frame and the greedy cross-scale dedup at mask IoU
        0.4 in descending score order.  Everything runs on the device: resize, predictor, class pass, nearest resize, IoUs.
        ``model_idx`` may be a list of model indices: the per-scale step is then the ensemble class pass (a10 + a14) on
        the scaled image (BASELINE configs[3]: ensemble + multi-scale)."""
        h, w = int(image_dev.shape[0]), int(image_dev.shape[1])
        model_ids = [int(model_idx)] if isinstance(model_idx, (int, np.integer)) else [int(m) for m in model_idx]
        eng = self.predictors[model_ids[0]].engine
        is_small = target_class in small_classes
        area0 = h * w
        base_min = max(3, int(area0 * 0.000005)) if is_small else max(25, int(area0 * 0.0001))

        def single_scale(scale: float):
            sh, sw = (h, w) if scale == 1.0 else (int(h * scale), int(w * scale))
            img = image_dev[None] if scale == 1.0 else eng.resize_linear_u8(image_dev[None].contiguous(), sh, sw)
            dets = [self._predict_batch(m, image_key + ("|full" if scale == 1.0 else f"|scale{scale}"), img)[0] for m in model_ids]
            self.ops.set_frame_width(sw)
            if len(model_ids) > 1:
                masks, sc, _ = self._ensemble_class_pass(dets, target_class, small_classes, confidence_threshold, iou_threshold)
            else:
                masks, sc, _ = self._single_model_class_pass(dets[0], target_class, small_classes, confidence_threshold, iou_threshold)
            if masks is None or isinstance(masks, str) or masks.shape[0] == 0:
                return None, []
            area, _ = self.ops.area_bbox(masks)
            keep = np.nonzero(area.cpu().numpy() >= int(base_min * (scale ** 2)))[0]
            if len(keep) == 0:
                return None, []
            masks = masks[torch.from_numpy(keep).to(self.dev)].contiguous()
            if scale != 1.0:
                masks = self.ops.place_tiles(masks, [0] * len(keep), [0] * len(keep), h, w, h, w, src_w=sw)
            return masks, [sc[i] for i in keep]

        parts, scores, found = [], [], {}

        def add(scale):
            m, sc = single_scale(scale)
            found[scale] = len(sc)
            return m, sc

        for scale in (0.7, 1.0, 1.5):
            m, sc = add(scale)
            if m is not None:
                parts.append(m)
                scores.extend(sc)
        base = found.get(1.0, 0)
        for unlocked, extra in ((found.get(1.5, 0) > base * 0.1, (2.0, 2.5)), (found.get(0.7, 0) > base * 0.1, (0.5, 0.6))):
            if not unlocked:
                continue
            for scale in extra:
                m, sc = add(scale)
                if len(sc) < base * 0.05:
                    break
                if m is not None:
                    parts.append(m)
                    scores.extend(sc)
        self.ops.set_frame_width(w)
        if not parts:
            return None, [], []
        packed = torch.cat(parts, dim=0)
        alg = DeviceMaskAlgebra(self.ops, packed)
        order = np.argsort(-np.asarray(scores, dtype=np.float64), kind="stable")
        alg.prefetch_overlapping_pairs([list(range(alg.n))])
        kept: List[int] = []
        for idx in order:
            if not any(alg.iou(int(idx), k) > 0.4 for k in kept):
                kept.append(int(idx))
        sel = torch.tensor(kept, dtype=torch.long, device=self.dev)
        return packed[sel].contiguous(), [scores[i] for i in kept], [target_class] * len(kept)

    def _merge_full_and_tiles(self, full_masks, full_scores, full_classes, tile_masks, tile_scores, tile_classes):
        """``inference.py:2452-2472``: full-image results + tile results -> ``deduplicate_masks_smart`` at 0.4 (N4 included)."""
        if isinstance(full_masks, str):        # N4: ndarray + list
            if tile_masks:
                raise EmptyEnsembleTypeError("operands could not be broadcast together (empty ensemble result + tile masks)")
            return None, [], []
        parts = ([full_masks] if full_masks is not None and full_masks.shape[0] else []) + list(tile_masks)
        if not parts:
            return None, [], []
        packed = torch.cat(parts, dim=0)
        if self.merge_mode == "soft_nms":        # flagged non-parity mode (f4)
            return self.soft_nms_merge(packed, list(full_scores) + list(tile_scores), list(full_classes) + list(tile_classes),
                                       self.soft_nms_sigma, self.soft_nms_score_threshold)
        return self.deduplicate_masks_smart(packed, list(full_scores) + list(tile_scores), list(full_classes) + list(tile_classes), 0.4)

    _encode_table = staticmethod(parallel.encode_instance_table)      # (a seam: the failure-agreement test makes it raise on one rank)

    def gather_and_merge(self, locals_by_class: Dict[int, tuple], hw: Tuple[int, int], ensemble_by_class: Dict[int, bool], status: int = 0):
        """The ONE exchange of the multi-GPU path, once per IMAGE: every rank contributes the class-tagged instance tables
        of all its local class passes (full-image pass on rank 0, its tiles), every rank receives the global table ordered
        by (unit id, local order) and runs the per-class 0.4 merges on it -- deterministic and identical everywhere.
        Rows of one class keep the reference's order (full image first, tiles in row-major order, detector order inside),
        because each rank appends its classes in the same order and the merge is stable.  Returns {class: (masks, scores,
        classes)}; a class whose merge would raise in the reference (N4) maps to an ``EmptyEnsembleTypeError`` instance.

        ``status`` != 0: this rank's local passes of the image FAILED (its ``locals_by_class`` is empty).  It still takes part
        in the exchange, and every rank -- the failed one included -- raises :class:`PeerImageFailure` right after it, so the
        image is skipped by all ranks together and the next image's exchange finds every rank at the same collective."""
        h, w = hw
        hdrs, pays = [], []
        try:
            # (building this rank's table allocates -- the concatenated full-frame masks, the crop -- and can fail like the
            # local passes can: it must not keep the rank out of the collective either)
            for cls, (fm, fs, fc, tm, ts, tc, tu) in (locals_by_class.items() if status == 0 else ()):
                empty_full = isinstance(fm, str)
                parts = ([fm] if (fm is not None and not empty_full and fm.shape[0]) else []) + list(tm)
                sc = ([] if (fm is None or empty_full) else list(fs)) + list(ts)
                cl = [cls] * len(sc)
                un = [0] * (len(sc) - len(ts)) + list(tu)
                if self.rank == 0 and empty_full:   # N4 marker travels too: unit -1 row of this class, no payload
                    mark = torch.zeros((1, parallel.HDR), dtype=torch.int32, device=self.dev)
                    mark[0, 0] = -1
                    mark[0, 1] = cls
                    mark[0, 4:8] = -1
                    hdrs.append(mark)
                if parts:
                    local = torch.cat(parts, dim=0)
                    a, b = self.ops.area_bbox(local)
                    hdr, pay = self._encode_table(local, sc, cl, un, b.cpu().numpy(), a.cpu().numpy())
                    hdrs.append(hdr)
                    pays.append(pay)
            hdr = torch.cat(hdrs, dim=0) if hdrs else torch.zeros((0, parallel.HDR), dtype=torch.int32, device=self.dev)
            pay = torch.cat(pays, dim=0) if pays else torch.zeros((0,), dtype=torch.int32, device=self.dev)
        except Exception as e:
            system_logger.error(f"Rank {self.rank}: building the instance table of an image failed: {e}", exc_info=True)
            status = 1
            hdr = torch.zeros((0, parallel.HDR), dtype=torch.int32, device=self.dev)
            pay = torch.zeros((0,), dtype=torch.int32, device=self.dev)
        gt = parallel.all_gather_instance_tables(hdr, pay, status=status, state=self.exchange)
        if bool((gt.status != 0).any()):
            raise PeerImageFailure(f"local passes failed on rank(s) {np.nonzero(gt.status)[0].tolist()}: every rank skips this image")
        packed_all, s_all, c_all, u_all = parallel.decode_instance_table(gt.header, gt.payload, h, w, self.dev, host_header=gt.host_header, offsets=gt.offsets)
        out = {}
        c_arr, u_arr = np.asarray(c_all, dtype=np.int64), np.asarray(u_all, dtype=np.int64)
        for cls in locals_by_class:
            rows = np.nonzero(c_arr == cls)[0]
            marker = rows[u_arr[rows] == -1]
            rows = rows[u_arr[rows] != -1]
            if len(marker):
                out[cls] = (EmptyEnsembleTypeError("operands could not be broadcast together (empty ensemble result + tile masks)")
                            if len(rows) else (None, [], []))
                continue
            if len(rows) == 0:
                out[cls] = (None, [], [])
                continue
            sel = torch.from_numpy(rows).to(self.dev)
            sc = [s_all[i] for i in rows]
            if not ensemble_by_class.get(cls, False):
                sc = [np.float32(v) for v in sc]      # single-model scores are the predictor's float32 values
            out[cls] = self.deduplicate_masks_smart(packed_all[sel].contiguous(), sc, [cls] * len(rows), 0.4)
        return out

    def tile_pipeline_all_classes(self, image_key: str, image_dev: torch.Tensor, class_params: Dict[int, Tuple[float, float]], small_classes,
                                  tile_size=512, overlap_ratio=0.1, upscale_factor=2.0, edge_filter_enabled=True):
        """``tile_based_inference_pipeline`` (``inference.py:2299-2485``) of ONE model for ALL classes of an image at once, in phases
        instead of class by class -- the same kernels on the same masks, TWO device-to-host waits instead of four per class:
        (1) the class passes of every class over the full image AND over the tiles are enqueued back to back and their tables
        fetched together; (2) after the placement of the tile results (a13) the per-class 0.4 merges run as the segments of one
        ``deduplicate_masks_smart_segments`` call.  ``class_params``: {class: (confidence threshold, IoU threshold)} in class-loop
        order.  Returns {class: (masks, scores, classes)} exactly as the per-class calls do (checked byte for byte on the CLI's
        files: ``test_grouped_padded_and_graphed_forwards_write_the_image_by_image_files`` runs both)."""
        assert self.world == 1 and self.merge_mode == "smart"
        h, w = int(image_dev.shape[0]), int(image_dev.shape[1])
        offs = self._tile_offsets(h, w, tile_size, overlap_ratio)
        uh = uw = int(tile_size * upscale_factor)
        mine = list(range(len(offs)))
        edge = int(tile_size * overlap_ratio / 2)
        full_det = self._predict_batch(0, image_key + "|full", image_dev[None])
        tkey = self._tiles_key(image_key, tile_size, overlap_ratio, upscale_factor)
        tile_dets = self._predict_batch(0, tkey, None if (0, tkey) in self._cache else self._my_tiles(image_dev, tile_size, overlap_ratio, upscale_factor))
        # ---- phase 1: every class pass enqueued, one wait --------------------------------------------------------------------
        handles = []
        for cls, (conf, _) in class_params.items():
            self.ops.set_frame_width(w)
            hf = self._single_class_pass_launch(full_det, cls, small_classes, conf)
            self.ops.set_frame_width(uw)           # the tile masks' own frame: its right border is pixel uw - 1
            ht = self._single_class_pass_launch(tile_dets, cls, small_classes, conf)
            handles.append((cls, hf, ht))
        live = [h_ for _, hf, ht in handles for h_ in (hf, ht) if h_ is not None]
        host = torch.cat([t_ for h_ in live for t_ in (h_["ncols"], h_["area"], h_["bbox"].reshape(-1), h_["I"].reshape(-1))]).cpu().numpy() if live else None
        if live:
            self.d2h_waits += 1
        pos = 0

        def tables(h_):
            nonlocal pos
            T, n, ld = h_["T"], h_["n"], h_["ld"]
            tabs = dict(ncols=host[pos:pos + T], area=host[pos + T:pos + T + n].astype(np.int64),
                        bbox=host[pos + T + n:pos + T + 5 * n].reshape(n, 4).astype(np.int64),
                        I=np.ascontiguousarray(host[pos + T + 5 * n:pos + T + 5 * n + n * ld]).reshape(n, ld))
            pos += T + 5 * n + n * ld
            return tabs
        parts, scores, classes, segments = [], [], [], []
        for cls, hf, ht in handles:
            iou_thr = class_params[cls][1]
            is_small = cls in small_classes
            s0 = len(scores)
            if hf is not None:
                big, res, tabs = self._single_class_pass_finish(hf, tables(hf), is_small, iou_thr)
                kept, sc = res[0]
                if kept:
                    self.ops.set_frame_width(w)
                    parts.append(self.ops.gather_regions(big, kept, tabs.bbox[kept]))
                    scores += list(sc)
                    classes += [cls] * len(kept)
            if ht is not None:
                big, res, tabs = self._single_class_pass_finish(ht, tables(ht), is_small, iou_thr)
                self.ops.set_frame_width(uw)
                tm_, ts_, tc_, _ = self._place_tile_results(big, res, tabs, mine, offs, tile_size, uh, uw, h, w, edge, edge_filter_enabled, cls)
                parts += tm_
                scores += ts_
                classes += tc_
            segments.append((s0, len(scores)))
        self.ops.set_frame_width(w)
        out = {cls: (None, [], []) for cls in class_params}
        if not parts:
            return out
        # ---- phase 2: the per-class 0.4 merges (inference.py:2452-2472) as the segments of one call, one wait --------------------
        merged = self.deduplicate_masks_smart_segments(torch.cat(parts, dim=0), scores, classes, segments, 0.4)
        for (cls, _, _), m in zip(handles, merged):
            out[cls] = m[:3]
        return out

    def tile_based_inference_pipeline(self, model_ids: Sequence[int], image_key: str, image_dev: torch.Tensor, target_class,
                                      small_classes, confidence_threshold, tile_size=512, overlap_ratio=0.1, upscale_factor=2.0,
                                      iou_threshold=0.7, edge_filter_enabled=True):
        """``inference.py:2299-2485`` for one class (with more than one rank: one exchange for this class; ``run_inference``
        batches the exchange over the classes of an image instead)."""
        loc = self._tile_pipeline_local(model_ids, image_key, image_dev, target_class, small_classes, confidence_threshold,
                                        tile_size, overlap_ratio, upscale_factor, iou_threshold, edge_filter_enabled)
        if self.world > 1:
            res = self.gather_and_merge({target_class: loc}, (int(image_dev.shape[0]), int(image_dev.shape[1])),
                                        {target_class: len(model_ids) > 1})[target_class]
            if isinstance(res, Exception):
                raise res
            return res
        return self._merge_full_and_tiles(*loc[:6])

    # ------------------------------------------------------------------ batch of independent tiles (configs[1] / [4])
    def process_tile_batch_unbatched(self, key: str, tiles: torch.Tensor, small_classes, class_thresholds: Dict[int, Tuple[float, float]],
                                     spatial_cfg: Optional[dict] = None, um_pix: float = 1.0, model_ids: Sequence[int] = (0,)):
        """Reference-shaped loop (tile by tile, class by class) -- kept as the checker of the batched version below."""
        dets = [self._predict_batch(m, key, tiles) for m in model_ids]
        out = []
        for t in range(tiles.shape[0]):
            parts, scores, classes = [], [], []
            for cls, (conf, iou_thr) in class_thresholds.items():
                if len(model_ids) > 1:
                    m, s, c = self._ensemble_class_pass([d[t] for d in dets], cls, small_classes, conf, iou_thr)
                else:
                    m, s, c = self._single_model_class_pass(dets[0][t], cls, small_classes, conf, iou_thr)
                if m is None or isinstance(m, str) or m.shape[0] == 0:
                    continue
                parts.append(m)
                scores.extend(s)
                classes.extend(c)
            packed = torch.cat(parts, dim=0) if parts else None
            packed, scores, classes = self.deduplicate_masks_smart(packed, scores, classes, iou_threshold=0.7)
            recs = []
            if packed is not None and packed.shape[0]:
                if spatial_cfg is not None and spatial_cfg.get("enabled", False):
                    keep = apply_spatial_constraints_indices(DeviceMaskAlgebra(self.ops, packed), scores, classes, spatial_cfg)
                    packed = packed[torch.tensor(keep, dtype=torch.long, device=self.dev)].contiguous()
                    scores, classes = [scores[i] for i in keep], [classes[i] for i in keep]
                recs = self.ops.contours(packed, max_contours=256, um_pix=um_pix) if packed.shape[0] else []
            out.append((packed, scores, classes, recs))
        return out

    def _gather_selected(self, dets: Sequence[_Detections], sels: Sequence[np.ndarray], pool_name: Optional[str] = None):
        """Our own copy of the selected masks of many detections sets (every later stage works on it in place) and
        their bbox hints: ONE gather per run of sets that share a forward's table.  ``pool_name`` (tile-batch loop with
        ``pooled_planes``): the copy goes into that plane pool -- boxes written, not planes."""
        dev = self.dev
        lens = [len(x) for x in sels]
        T = len(dets)
        pool = None
        if pool_name is not None and self.pooled_planes and all(d.base is not None for d in dets):
            pool = self.ops.pool(pool_name, sum(lens), int(dets[0].packed.shape[1]), int(dets[0].packed.shape[2]))
            packed = pool.planes[:sum(lens)]
        else:
            packed = torch.empty((sum(lens),) + tuple(dets[0].packed.shape[1:]), dtype=dets[0].packed.dtype, device=dev)
        bbox = torch.empty((sum(lens), 4), dtype=torch.int32, device=dev)
        have_hint, pos, t = True, 0, 0
        while t < T:
            if dets[t].base is not None:                       # the run of sets that share this forward's table
                t1 = t
                while t1 < T and dets[t1].base is dets[t].base:
                    t1 += 1
                gi = np.concatenate([dets[u].base_idx[sels[u]] for u in range(t, t1)])
                if len(gi):
                    # the masks of a forward are zero outside their paste boxes: copy the boxes, write the planes once
                    gt = self.ops.upload(gi)
                    torch.index_select(dets[t].base_bbox, 0, gt, out=bbox[pos:pos + len(gi)])
                    if pool is not None:
                        self.ops.gather_regions_pooled(dets[t].base, gt, bbox[pos:pos + len(gi)], pool, first=pos)
                    else:
                        self.ops.gather_regions(dets[t].base, gt, bbox[pos:pos + len(gi)], out=packed[pos:pos + len(gi)])
                pos += len(gi)
                t = t1
            else:
                if lens[t]:
                    si = torch.from_numpy(sels[t]).to(dev)
                    if dets[t].bbox is None:
                        have_hint = False
                        torch.index_select(dets[t].packed, 0, si, out=packed[pos:pos + lens[t]])
                    else:
                        torch.index_select(dets[t].bbox, 0, si, out=bbox[pos:pos + lens[t]])
                        self.ops.gather_regions(dets[t].packed.contiguous(), si, bbox[pos:pos + lens[t]], out=packed[pos:pos + lens[t]])
                pos += lens[t]
                t += 1
        if not have_hint:
            _, bbox = self.ops.area_bbox(packed)
        return packed, bbox

    def _ensemble_class_pass_batched(self, dets_per_model: Sequence[Sequence[_Detections]], target_class: int, small_classes,
                                     conf, iou_threshold):
        """a10 + a14 (``run_ensemble_inference``, ``inference.py:1464-1598``) for ONE class over MANY tiles: every model's
        selected masks of every tile go through ONE stage program (fill -> erosion [-> dilation]), one contour trace and
        one pair-count launch; the per-tile decisions (min size, compactness, smart dedup, N6 included) then run on the
        host over integers exactly as the tile-by-tile version does.  Returns (big, per tile (kept indices, scores), alg)."""
        T, dev = len(dets_per_model[0]), self.dev
        is_small = target_class in small_classes
        hw = dets_per_model[0][0].hw
        area_img = hw[0] * hw[1]
        min_size = max(3, int(area_img * 0.000005)) if is_small else max(25, int(area_img * 0.0001))
        flat_dets, sels, owner, weights = [], [], [], []
        for m, (dets, weight) in enumerate(zip(dets_per_model, self.ensemble_weights)):     # model-major: one gather per model
            for t, det in enumerate(dets):
                sel = np.nonzero((det.classes == target_class) & (det.scores >= conf))[0] if len(det.scores) else np.zeros((0,), dtype=np.int64)
                flat_dets.append(det)
                sels.append(sel)
                owner.append((t, m))
                weights.append(weight)
        empty = [([], []) for _ in range(T)]
        if sum(len(x) for x in sels) == 0:
            return None, empty, None
        packed, bbox = self._gather_selected(flat_dets, sels)
        area, bbox, _ = self.ops.program_(packed, ["fill", "erode"] if is_small else ["fill", "erode", "dilate"], bbox)
        alg = DeviceMaskAlgebra(self.ops, packed, area=area, bbox=bbox)
        cset = self.ops.trace(packed, max_contours=256, bbox=alg._bbox_dev, total_area=int(alg.area.sum()))
        per0 = cset.first_contour_perimeter()
        # per tile, in the reference's order (model by model): survivors of the min-size rule with their weighted scores
        tile_items: List[List[int]] = [[] for _ in range(T)]
        tile_scores: List[List[float]] = [[] for _ in range(T)]
        pos = 0
        by_owner = {}
        for (t, m), det, sel, wgt in zip(owner, flat_dets, sels, weights):
            by_owner[(t, m)] = (pos, det, sel, wgt)
            pos += len(sel)
        for t in range(T):
            for m in range(len(dets_per_model)):
                p0, det, sel, wgt = by_owner[(t, m)]
                for k in range(len(sel)):
                    if alg.area[p0 + k] >= min_size:
                        tile_items[t].append(p0 + k)
                        tile_scores[t].append(float(det.scores[sel][k]) * wgt)
        keep0_all, groups = [], []
        for t in range(T):
            k0 = []
            for j, idx in enumerate(tile_items[t]):
                if alg.bbox[idx, 0] < 0:
                    continue
                per = per0[idx]
                if per > 0 and (4 * np.pi * int(alg.area[idx])) / (per ** 2) < 0.15:
                    continue
                k0.append(j)
            keep0_all.append(k0)
            if len(k0) > 1:
                groups.append([tile_items[t][j] for j in k0])
        alg.prefetch_overlapping_pairs(groups)
        out = []
        for t in range(T):
            k0 = keep0_all[t]
            if not k0:
                out.append(([], []))
                continue
            gidx = [tile_items[t][j] for j in k0]
            scores = [tile_scores[t][j] for j in k0]
            classes = [target_class] * len(k0)
            bb = [(int(alg.bbox[i, 0]), int(alg.bbox[i, 2]), int(alg.bbox[i, 1]), int(alg.bbox[i, 3])) for i in gidx]
            keep = self._dedup_smart_order(alg, gidx, scores, classes, bb, iou_threshold)
            out.append(([gidx[i] for i in keep], [scores[i] for i in keep]))
        return packed, out, alg

    def _single_class_pass_batched(self, dets: Sequence[_Detections], target_class: int, small_classes, conf, iou_threshold):
        """a6 + a9 + a11 + a12 for ONE class over MANY tiles: everything enqueued at once (:meth:`_single_class_pass_launch`),
        ONE device-to-host wait for its tables, the greedy loop as one native call (:meth:`_single_class_pass_finish`).
        Returns (masks, per tile (indices into them, scores), tables with ``area`` / ``bbox``).  The host-loop version
        below is its checker."""
        h = self._single_class_pass_launch(dets, target_class, small_classes, conf)
        T = len(dets)
        if h is None:
            return None, [([], []) for _ in range(T)], None
        host = torch.cat([h["ncols"], h["area"], h["bbox"].reshape(-1), h["I"].reshape(-1)]).cpu().numpy()
        self.d2h_waits += 1
        n, ld = h["n"], h["ld"]
        tabs = dict(ncols=host[:T], area=host[T:T + n].astype(np.int64), bbox=host[T + n:T + 5 * n].reshape(n, 4).astype(np.int64),
                    I=np.ascontiguousarray(host[T + 5 * n:T + 5 * n + n * ld]).reshape(n, ld))
        big, res, tables = self._single_class_pass_finish(h, tabs, target_class in small_classes, iou_threshold)
        if not any(len(k) for k, _ in res):
            return None, res, None
        return big, res, tables

    def _single_class_pass_batched_hostloops(self, dets: Sequence[_Detections], target_class: int, small_classes, conf, iou_threshold):
        """a6 + a9 + a11 + a12 for ONE class over MANY tiles with one launch per kernel: the masks of all tiles are
        concatenated and carry a segment id (tile index); overlap removal and column counts are segment-aware, every
        other kernel is per mask anyway.  Returns per tile (index tensor into the returned big tensor, scores)."""
        T = len(dets)
        dev = self.dev
        sels = []
        for det in dets:
            sel = np.nonzero(det.classes == target_class)[0]
            sels.append(sel[det.scores[sel] >= conf])
        lens = [len(x) for x in sels]
        empty = [([], []) for _ in range(T)]
        if sum(lens) == 0:
            return None, empty, None
        packed, bbox = self._gather_selected(dets, sels)
        is_small = target_class in small_classes
        min_size = self.class_specific_settings.get(f"class_{target_class}", {}).get("min_size", 5 if is_small else 25)
        seg_np = np.repeat(np.arange(T, dtype=np.int32), lens)
        seg = torch.from_numpy(seg_np).to(dev)
        ncols = (self.ops.column_counts(packed, seg, T, bbox=bbox) > min_size).sum(dim=1).cpu().numpy()
        keep_idx, new_lens, start = [], [0] * T, 0
        for t in range(T):
            n = lens[t]
            if n:
                sc = dets[t].scores[sels[t]]
                if bool(sc.all()) < 0.5:
                    n_keep = 0                                   # `ori_score.all() < score_threshold` (mask_utils.py:59)
                else:
                    n_keep = n if ncols[t] >= n else int(ncols[t])  # the column-count truncation quirk (62-68)
                keep_idx.extend(range(start, start + n_keep))
                new_lens[t] = n_keep
            start += n
        if sum(new_lens) == 0:
            return None, empty, None
        if len(keep_idx) != packed.shape[0]:
            ki = torch.tensor(keep_idx, dtype=torch.long, device=dev)
            packed, bbox = packed[ki].contiguous(), bbox[ki].contiguous()
            seg_np = np.repeat(np.arange(T, dtype=np.int32), new_lens)
            seg = torch.from_numpy(seg_np).to(dev)
        # a9: fill holes -> closing, then the score-ordered overlap removal, then the component test; a11 (calls with
        # more than two masks): fill -> erosion -> dilation.  Two region programs + one overlap launch, all in place.
        _, bbox, _ = self.ops.program_(packed, ["fill", "dilate", "erode"], bbox)
        self.ops.overlap_prefix_(packed, seg, bbox)
        if self.parallel_mask_processing:
            active = torch.from_numpy((np.asarray(new_lens)[seg_np] > 2).astype(np.uint8)).to(dev)
            area, bbox, _ = self.ops.program_(packed, ["drop_multi", "gate", "fill", "erode", "dilate"], bbox, active)
        else:
            area, bbox, _ = self.ops.program_(packed, ["drop_multi"], bbox)
        closed = packed
        thr = 0.5 if is_small else iou_threshold
        bounds = np.concatenate(([0], np.cumsum(new_lens)))
        alg = DeviceMaskAlgebra(self.ops, closed, area=area, bbox=bbox,
                                blocks=[np.arange(bounds[t], bounds[t + 1]) for t in range(T)])
        alg.prefetch_overlapping_pairs([list(range(bounds[t], bounds[t + 1])) for t in range(T) if new_lens[t] > 1])
        out = []
        for t in range(T):
            kept = self._greedy_keep(alg, range(bounds[t], bounds[t + 1]), thr)
            sc = dets[t].scores[sels[t]] if lens[t] else []
            out.append((kept, [sc[i - bounds[t]] for i in kept]))
        return closed, out, alg

    @staticmethod
    def _row_bits(rows: np.ndarray) -> List[int]:
        """Boolean matrix -> one Python int per row (bit j = rows[i, j]): the greedy loops below then run on integer
        bit operations instead of one small numpy call per candidate."""
        if rows.shape[0] == 0:
            return []
        packed = np.packbits(rows, axis=1, bitorder="little")
        return [int.from_bytes(packed[i].tobytes(), "little") for i in range(rows.shape[0])]

    @staticmethod
    def _pair_tables(alg: DeviceMaskAlgebra, ka: np.ndarray, need: np.ndarray):
        """|a & b| and IoU (float64 ``inter / union``, 0 where the union is empty -- ``inference.py:422-435, 2697-2719``)
        for every pair of ``ka`` that ``need`` marks; pairs the algebra does not know yet are fetched with ONE launch."""
        miss = need & ~alg.known[np.ix_(ka, ka)]
        miss = np.triu(miss | miss.T, 1)
        if miss.any():
            ii, jj = np.nonzero(miss)
            alg.intersections(ka[ii], ka[jj])
        inter = alg.I[np.ix_(ka, ka)]
        area = alg.area[ka]
        union = area[:, None] + area[None, :] - inter
        iou = np.divide(inter, union, out=np.zeros(inter.shape, dtype=np.float64), where=union > 0)
        return inter, iou

    @staticmethod
    def _dedup_smart_order(alg: DeviceMaskAlgebra, k0: Sequence[int], scores, classes, bb, iou_threshold: float) -> List[int]:
        """Step 2 of ``deduplicate_masks_smart`` (``inference.py:2640-2671``): ``sorted_indices[idx+1:]`` sliced by MASK
        INDEX and the mixed-axis bbox pre-filter are kept literally (N6).  The pair conditions are evaluated as matrices
        once; the order-dependent part is the reference's loop on bit sets."""
        n = len(k0)
        if n == 0:
            return []
        k0a = np.asarray(k0, dtype=np.int64)
        cls = np.asarray(classes)
        b = np.asarray(bb, dtype=np.int64).reshape(-1, 4)       # stored (y_min, y_max, x_min, x_max) ...
        order = np.argsort(np.asarray(scores, dtype=np.float64), kind="stable")[::-1]
        # ... read as (y_min, x_min, y_max, x_max): b1 = row mask, b2 = column mask
        lit = ~((b[:, None, 3] < b[None, :, 1]) | (b[None, :, 3] < b[:, None, 1]) |
                (b[:, None, 2] < b[None, :, 0]) | (b[None, :, 2] < b[:, None, 0]))
        cand = (cls[:, None] == cls[None, :]) & lit
        inter, iou = InferencePipeline._pair_tables(alg, k0a, cand)
        hit = InferencePipeline._row_bits(cand & (inter > 0) & (iou > iou_threshold))
        after = [0] * n                                          # bit set of order[p + 1:] for p = 0 .. n-1
        acc = 0
        for p in range(n - 1, -1, -1):
            after[p] = acc
            acc |= 1 << int(order[p])
        removed = 0
        keep: List[int] = []
        for idx in order.tolist():
            if (removed >> idx) & 1:
                continue
            keep.append(idx)
            removed |= hit[idx] & after[idx]                     # `others = sorted_indices[idx + 1:]`: idx used as a position
        return keep

    @staticmethod
    def _greedy_keep(alg: DeviceMaskAlgebra, indices, thr: float) -> List[int]:
        """``inference.py:1451-1459`` over an index range (same integer counts, same float64 division as the scalar
        loop): keep mask i unless IoU(mask_i, kept_j) > thr for a mask kept before it."""
        idx = [int(i) for i in indices]
        n = len(idx)
        if n == 0:
            return []
        ka = np.asarray(idx, dtype=np.int64)
        _, iou = InferencePipeline._pair_tables(alg, ka, np.ones((n, n), dtype=bool))
        hit = InferencePipeline._row_bits(iou > thr)
        kept_bits, kept = 0, []
        for p in range(n):
            if hit[p] & kept_bits:
                continue
            kept_bits |= 1 << p
            kept.append(idx[p])
        return kept

    def process_tile_batch(self, key: str, tiles: torch.Tensor, small_classes, class_thresholds: Dict[int, Tuple[float, float]],
                           spatial_cfg: Optional[dict] = None, um_pix: float = 1.0, model_ids: Sequence[int] = (0,),
                           dets: Optional[List[_Detections]] = None):
        """The per-tile unit of work of the headline metric (see :meth:`process_tile_batch_hostloops` for the stages).
        Single-model calls take the THREE-WAIT path: the class passes of all classes are enqueued back to back with every
        data-dependent choice (the column-count truncation, the `> 2 masks` gate) made on the device, the pair counts come
        from ``demia_mask_pair_matrix`` (pair list made on the device), and the host waits once for the tables of all class
        passes, runs the greedy loops as ONE native call per class (``demia_host_greedy_keep``), enqueues the cross-class
        stage (gather, contour trace, measurements of every candidate, pair matrix) and waits once more before the native
        smart dedup (``demia_host_dedup_smart``).  With the forward's own result that is three device-to-host waits per
        batch, whatever its size.  Same keep lists, masks and records as the host-loop version, which stays as its checker
        (``tests/test_gpu_parity_maskops.py``)."""
        ensemble = len(model_ids) > 1
        ops, dev, lib = self.ops, self.dev, self.ops.lib
        ops.set_frame_width(int(tiles.shape[2]))
        if ensemble:
            # the ensemble (a10 + a14 per class, inference.py:1464-1598) on the same three waits: (1) the tables of ALL models'
            # forwards, (2) the class passes of all classes -- one gather per model, one stage program per class, ONE contour
            # trace and ONE pair matrix over (class, tile) runs, fetched together -- and the native smart dedup per class,
            # (3) the cross-class stage below, shared with the single-model path
            # (``dets``: the caller's own forwards, one list per model, e.g. a software-pipelined loop; else through the cache)
            dets_per_model = list(dets) if dets is not None else self._predict_batches(model_ids, key, tiles)
            T = len(dets_per_model[0])
            out = [(None, [], [], []) for _ in range(T)]
            self.last_batch_stats = [(np.zeros((0,), dtype=np.int64), np.zeros((0, 4), dtype=np.int64)) for _ in range(T)]
            h = self._ensemble_passes_launch(dets_per_model, class_thresholds, small_classes)
            self._release_forward_outputs([d_ for dm in dets_per_model for d_ in dm])     # the gathers above were the last reads
            if h is None:
                return out
            passes = self._ensemble_passes_finish(h, class_thresholds, small_classes)
        else:
            if dets is None:
                dets = self._predict_batch(model_ids[0], key, tiles)
            T = len(dets)
            out = [(None, [], [], []) for _ in range(T)]
            self.last_batch_stats = [(np.zeros((0,), dtype=np.int64), np.zeros((0, 4), dtype=np.int64)) for _ in range(T)]
            # ---- class passes: enqueue all, wait once ---------------------------------------------------------------
            handles = [self._single_class_pass_launch(dets, cls, small_classes, conf) for cls, (conf, _) in class_thresholds.items()]
            self._release_forward_outputs(dets)       # the gathers of the class passes were the last reads of the forward's planes
            live = [h for h in handles if h is not None]
            if not live:
                return out
            host = torch.cat([t_ for h in live for t_ in (h["ncols"], h["area"], h["bbox"].reshape(-1), h["I"].reshape(-1))]).cpu().numpy()
            self.d2h_waits += 1
            passes, pos = [], 0
            for (cls, (_, iou_thr)), h in zip(class_thresholds.items(), handles):
                if h is None:
                    continue
                n, ld = h["n"], h["ld"]
                tabs = dict(ncols=host[pos:pos + T], area=host[pos + T:pos + T + n].astype(np.int64),
                            bbox=host[pos + T + n:pos + T + 5 * n].reshape(n, 4).astype(np.int64),
                            I=np.ascontiguousarray(host[pos + T + 5 * n:pos + T + 5 * n + n * ld]).reshape(n, ld))
                pos += T + 5 * n + n * ld
                big, res, calg = self._single_class_pass_finish(h, tabs, cls in small_classes, iou_thr)
                if any(len(k) for k, _ in res):
                    passes.append((cls, big, res, calg))
        total = sum(len(k) for _, _, res, _ in passes for k, _ in res)
        if total == 0:
            return out
        # ---- what survives, gathered once per class into `allp` (class-major; a (class, tile) run is contiguous) -------
        H_, wpr_ = int(passes[0][1].shape[1]), int(passes[0][1].shape[2])
        apool = ops.pool("all", total, H_, wpr_) if self.pooled_planes else None
        allp = apool.planes[:total] if apool is not None else torch.empty((total, H_, wpr_), dtype=passes[0][1].dtype, device=dev)
        tile_items: List[List[int]] = [[] for _ in range(T)]
        scores_all = np.zeros(total, dtype=np.float64)
        classes_all = np.zeros(total, dtype=np.int32)
        run_first = np.zeros(total, dtype=np.int32)
        run_count = np.zeros(total, dtype=np.int32)
        area_parts, bbox_parts = [], []
        off = 0
        for cls, big, res, calg in passes:
            src = [i for kept, _ in res for i in kept]
            p0 = off
            for t, (kept, sc) in enumerate(res):
                k = len(kept)
                tile_items[t].extend(range(p0, p0 + k))
                scores_all[p0:p0 + k] = sc
                run_first[p0:p0 + k] = p0
                run_count[p0:p0 + k] = k
                p0 += k
            classes_all[off:off + len(src)] = cls
            if apool is not None:
                ops.gather_regions_pooled(big, src, calg.bbox[src], apool, first=off, grow=0)
            else:
                ops.gather_regions(big, src, calg.bbox[src], out=allp[off:off + len(src)])   # tight boxes of the class pass
            area_parts.append(calg.area[src])
            bbox_parts.append(calg.bbox[src])
            off += len(src)
        area_all = np.concatenate(area_parts)
        bbox_all = np.ascontiguousarray(np.concatenate(bbox_parts))
        bbox_dev = ops.upload(bbox_all.astype(np.int32))
        # ---- cross-class stage (a14 at 0.7): trace + measure every candidate + same-class pair counts, ONE wait -------
        cset = ops.trace(allp, max_contours=256, bbox=bbox_dev, total_area=int(area_all.sum()))
        cset.launch_measure(um_pix, slots=4)
        ld2 = int(run_count.max())
        I2 = ops.pair_matrix(allp, bbox_dev, run_first, run_count, None, ld2)
        (I2h,) = cset.fetch(extra=[I2], with_points=True)
        self.d2h_waits += 1
        per0 = cset.first_contour_perimeter()
        ok = (bbox_all[:, 0] >= 0) & ~((per0 > 0) & ((4 * np.pi * area_all) / np.where(per0 > 0, per0, 1.0) ** 2 < 0.15))
        keep0_all = [[i for i in tile_items[t] if ok[i]] for t in range(T)]
        items = np.asarray([i for k0 in keep0_all for i in k0], dtype=np.int32)
        tile_off = np.concatenate(([0], np.cumsum([len(k0) for k0 in keep0_all]))).astype(np.int32)
        keep_out = np.zeros(max(len(items), 1), dtype=np.int32)
        keep_cnt = np.zeros(T, dtype=np.int32)
        if len(items):
            sc_items = np.ascontiguousarray(scores_all[items])
            cl_items = np.ascontiguousarray(classes_all[items])
            I2c = np.ascontiguousarray(I2h, dtype=np.int32)
            _L.check(lib.demia_host_dedup_smart(I2c.ctypes.data, ld2, run_first.ctypes.data, area_all.ctypes.data, bbox_all.ctypes.data,
                                                items.ctypes.data, sc_items.ctypes.data, cl_items.ctypes.data, tile_off.ctypes.data, T, 0.7,
                                                keep_out.ctypes.data, keep_cnt.ctypes.data), "demia_host_dedup_smart")
        final_idx: List[List[int]] = []
        alg = None
        for t in range(T):
            k0 = keep0_all[t]
            keep = keep_out[tile_off[t]:tile_off[t] + keep_cnt[t]].tolist()
            gl = [k0[i] for i in keep]
            sc, cl = [scores_all[i] for i in gl], [int(classes_all[i]) for i in gl]
            sc = [float(v) for v in sc] if ensemble else [self._score_type(v) for v in sc]     # ensemble scores: f64 products
            if gl and spatial_cfg is not None and spatial_cfg.get("enabled", False):
                if alg is None:                 # the containment / overlap rules ask for arbitrary pairs: the general algebra
                    alg = DeviceMaskAlgebra(ops, allp, area=area_all, bbox=bbox_all, blocks=tile_items)
                kk = apply_spatial_constraints_indices(alg.view(gl), sc, cl, spatial_cfg)
                gl, sc, cl = [gl[i] for i in kk], [sc[i] for i in kk], [cl[i] for i in kk]
            final_idx.append(gl)
            out[t] = (None, sc, cl, [])
        flat = [i for gl in final_idx for i in gl]
        if not flat:
            return out
        if self.pooled_planes:
            finalp = ops.gather_regions_pooled(allp, flat, bbox_all[flat], ops.pool("final", len(flat), H_, wpr_), grow=0)
        else:
            finalp = ops.gather_regions(allp, flat, bbox_all[flat])
        recs = cset.records(um_pix=um_pix, measure=True, select=flat)
        self.d2h_waits += cset.blocking_point_copies       # (0 in the steady state: the points came with the fetch above)
        pos = 0
        self.last_batch_stats = [(area_all[final_idx[t]], bbox_all[final_idx[t]]) for t in range(T)]
        for t in range(T):
            n = len(final_idx[t])
            if n:
                out[t] = (finalp[pos:pos + n], out[t][1], out[t][2], recs[pos:pos + n])
            pos += n
        return out

    # ---- ensemble class passes of ALL classes over many tiles, in two halves like the single-model pass below
    def _ensemble_passes_launch(self, dets_per_model: Sequence[Sequence[_Detections]], class_thresholds, small_classes):
        ops = self.ops
        M, T = len(dets_per_model), len(dets_per_model[0])
        cls_list = list(class_thresholds.items())
        zero = np.zeros((0,), dtype=np.int64)
        sel = {}
        for ci, (cls, (conf, _)) in enumerate(cls_list):
            for t in range(T):
                for m in range(M):
                    det = dets_per_model[m][t]
                    sel[(ci, t, m)] = np.nonzero((det.classes == cls) & (det.scores >= conf))[0] if len(det.scores) else zero
        # gather MODEL-major (one launch per model: the sets of a model share its forward's table) ...
        flat_dets, flat_sels, mm_off, pos = [], [], {}, 0
        for m in range(M):
            for ci in range(len(cls_list)):
                for t in range(T):
                    flat_dets.append(dets_per_model[m][t])
                    flat_sels.append(sel[(ci, t, m)])
                    mm_off[(ci, t, m)] = pos
                    pos += len(sel[(ci, t, m)])
        n = pos
        if n == 0:
            return None
        packed_mm, bbox_mm = self._gather_selected(flat_dets, flat_sels)
        # ... then ONE permuting gather into (class, tile, model) order: a (class, tile) run is contiguous and holds the
        # reference's order inside a tile (model by model, detector order: inference.py:1510-1540)
        perm = np.empty(n, dtype=np.int64)
        scores = np.empty(n, dtype=np.float64)
        run_first = np.empty(n, dtype=np.int32)
        run_count = np.empty(n, dtype=np.int32)
        cls_bounds, runs, pos = [], {}, 0
        for ci in range(len(cls_list)):
            c0 = pos
            for t in range(T):
                r0 = pos
                for m in range(M):
                    s_ = sel[(ci, t, m)]
                    k = len(s_)
                    perm[pos:pos + k] = mm_off[(ci, t, m)] + np.arange(k)
                    scores[pos:pos + k] = dets_per_model[m][t].scores[s_].astype(np.float64) * float(self.ensemble_weights[m])
                    pos += k
                run_first[r0:pos] = r0
                run_count[r0:pos] = pos - r0
                runs[(ci, t)] = (r0, pos)
            cls_bounds.append((c0, pos))
        if np.array_equal(perm, np.arange(n)):
            packed, bbox0 = packed_mm, bbox_mm
        else:
            pd = ops.upload(perm)
            bbox0 = bbox_mm.index_select(0, pd)
            packed = ops.gather_regions(packed_mm, pd, bbox0)
        areas, boxes = [], []
        for (cls, _), (c0, c1) in zip(cls_list, cls_bounds):
            if c1 > c0:
                a_, b_, _ = ops.program_(packed[c0:c1], ["fill", "erode"] if cls in small_classes else ["fill", "erode", "dilate"], bbox0[c0:c1])
                areas.append(a_)
                boxes.append(b_)
        area = torch.cat(areas) if len(areas) > 1 else areas[0]
        bbox = (torch.cat(boxes) if len(boxes) > 1 else boxes[0]).contiguous()
        # (the masks' areas are not on the host yet: the point pool is sized from the mask count alone)
        cset = ops.trace(packed, max_contours=256, bbox=bbox, max_points=int(min(4096 * n + (1 << 16), 1 << 26)))
        ld = int(run_count.max())
        I = ops.pair_matrix(packed, bbox, run_first, run_count, None, ld)
        return dict(packed=packed, area=area, bbox=bbox, cset=cset, I=I, ld=ld, n=n, T=T, scores=scores, run_first=run_first,
                    runs=runs, hw=dets_per_model[0][0].hw)

    def _ensemble_passes_finish(self, h: dict, class_thresholds, small_classes):
        n, T, ld = h["n"], h["T"], h["ld"]
        extra = [h["area"].to(torch.int32), h["bbox"], h["I"]]
        try:
            area_h, bbox_h, I_h = h["cset"].fetch(extra=extra)      # THE wait of the class passes
        except _L.HipKernelError as e:
            if "overflow" not in str(e):
                raise
            # a few large ragged masks (boundaries of thousands of points) overflowed the pool sized from the mask COUNT at
            # launch time: the areas are on the device by now -- trace again with the area-sized pool MaskOps.trace defaults to
            self.d2h_waits += 1
            h["cset"] = self.ops.trace(h["packed"], max_contours=256, bbox=h["bbox"], total_area=int(h["area"].sum().item()))
            area_h, bbox_h, I_h = h["cset"].fetch(extra=extra)
        self.d2h_waits += 1
        per0 = h["cset"].first_contour_perimeter()
        area = np.ascontiguousarray(area_h, dtype=np.int64)
        bbox = np.ascontiguousarray(bbox_h.reshape(n, 4), dtype=np.int64)
        I = np.ascontiguousarray(I_h.reshape(n, ld), dtype=np.int32)
        area_img = h["hw"][0] * h["hw"][1]
        scores, run_first = h["scores"], h["run_first"]
        compact_bad = (per0 > 0) & ((4 * np.pi * area) / np.where(per0 > 0, per0, 1.0) ** 2 < 0.15)
        passes = []
        for ci, (cls, (_, iou_thr)) in enumerate(class_thresholds.items()):
            min_size = max(3, int(area_img * 0.000005)) if cls in small_classes else max(25, int(area_img * 0.0001))
            ok = (area >= min_size) & (bbox[:, 0] >= 0) & ~compact_bad
            k0_all = [np.nonzero(ok[r0:r1])[0] + r0 for r0, r1 in (h["runs"][(ci, t)] for t in range(T))]
            items = np.ascontiguousarray(np.concatenate(k0_all), dtype=np.int32) if k0_all else np.zeros((0,), dtype=np.int32)
            if len(items) == 0:
                continue
            tile_off = np.concatenate(([0], np.cumsum([len(k) for k in k0_all]))).astype(np.int32)
            keep_out = np.zeros(len(items), dtype=np.int32)
            keep_cnt = np.zeros(T, dtype=np.int32)
            sc_items = np.ascontiguousarray(scores[items])
            cl_items = np.full(len(items), cls, dtype=np.int32)
            _L.check(self.ops.lib.demia_host_dedup_smart(I.ctypes.data, ld, run_first.ctypes.data, area.ctypes.data, bbox.ctypes.data,
                                                         items.ctypes.data, sc_items.ctypes.data, cl_items.ctypes.data, tile_off.ctypes.data, T,
                                                         float(iou_thr), keep_out.ctypes.data, keep_cnt.ctypes.data), "demia_host_dedup_smart")
            res = []
            for t in range(T):
                keep = keep_out[tile_off[t]:tile_off[t] + keep_cnt[t]]
                gl = k0_all[t][keep]
                res.append((gl.tolist(), scores[gl].tolist()))
            passes.append((cls, h["packed"], res, _PassTables(area, bbox)))
        return passes

    @staticmethod
    def _score_type(v):
        return np.float32(v)          # single-model scores are the predictor's float32 values

    # ---- single-model class pass over many tiles, in two halves: everything that can be enqueued without knowing a
    # ---- device result (launch), then the host decisions over the fetched tables (finish)
    def _single_class_pass_launch(self, dets: Sequence[_Detections], target_class: int, small_classes, conf):
        T, ops = len(dets), self.ops
        sels = []
        for det in dets:
            sel = np.nonzero(det.classes == target_class)[0]
            sel = sel[det.scores[sel] >= conf]
            if len(sel) and bool(det.scores[sel].all()) < 0.5:
                sel = sel[:0]                                     # `ori_score.all() < score_threshold` (mask_utils.py:59): nothing kept
            sels.append(sel)
        lens = np.asarray([len(x) for x in sels], dtype=np.int32)
        n = int(lens.sum())
        if n == 0:
            return None
        packed, bbox = self._gather_selected(dets, sels, pool_name=f"class{target_class}")
        is_small = target_class in small_classes
        min_size = self.class_specific_settings.get(f"class_{target_class}", {}).get("min_size", 5 if is_small else 25)
        starts = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
        seg_np = np.repeat(np.arange(T, dtype=np.int32), lens)
        first_np = starts[seg_np]
        tab = ops.upload(np.stack([seg_np, np.arange(n, dtype=np.int32) - first_np, lens[seg_np]]))
        seg, rank, nt = tab[0], tab[1], tab[2]
        # the column-count truncation quirk (mask_utils.py:62-68) ON THE DEVICE: a call keeps its first min(n, columns) masks.
        # The truncated tail stays in the batch as dead weight -- every later stage is per mask or only looks at EARLIER masks
        # of the call (overlap removal), so the live masks come out exactly as if the tail had been cut -- and the host cuts
        # it from its index lists once it has the counts.
        ncols = (ops.column_counts(packed, seg, T, bbox=bbox) > min_size).sum(dim=1).to(torch.int32)
        nc_m = ncols[seg.long()]
        newlen_m = torch.where(nc_m >= nt, nt, nc_m)
        _, bbox, _ = ops.program_(packed, ["fill", "dilate", "erode"], bbox)
        ops.overlap_prefix_(packed, seg, bbox)
        if self.parallel_mask_processing:
            active = ((rank < newlen_m) & (newlen_m > 2)).to(torch.uint8)         # process_masks_parallel: calls with > 2 masks
            area, bbox, _ = ops.program_(packed, ["drop_multi", "gate", "fill", "erode", "dilate"], bbox, active)
        else:
            area, bbox, _ = ops.program_(packed, ["drop_multi"], bbox)
        ld = int(lens.max())
        I = ops.pair_matrix(packed, bbox, first_np, lens[seg_np], None, ld)
        return dict(packed=packed, area=area, bbox=bbox, ncols=ncols, I=I, sels=sels, lens=lens, starts=starts, first=first_np,
                    n=n, ld=ld, T=T, dets=dets)

    def _single_class_pass_finish(self, h: dict, tabs: dict, is_small: bool, iou_threshold: float):
        T, lens, starts = h["T"], h["lens"], h["starts"]
        ncols = tabs["ncols"]
        new_lens = np.asarray([(int(lens[t]) if ncols[t] >= lens[t] else int(ncols[t])) if lens[t] else 0 for t in range(T)], dtype=np.int32)
        thr = 0.5 if is_small else iou_threshold
        keep = np.zeros(h["n"], dtype=np.uint8)
        area = np.ascontiguousarray(tabs["area"], dtype=np.int64)
        seg_first = np.ascontiguousarray(starts[:-1], dtype=np.int32)
        _L.check(self.ops.lib.demia_host_greedy_keep(tabs["I"].ctypes.data, h["ld"], h["first"].ctypes.data, area.ctypes.data,
                                                     seg_first.ctypes.data, new_lens.ctypes.data, T, float(thr), keep.ctypes.data),
                 "demia_host_greedy_keep")
        res = []
        for t in range(T):
            s0 = int(starts[t])
            kept = (s0 + np.nonzero(keep[s0:s0 + int(new_lens[t])])[0]).tolist()
            sc = h["dets"][t].scores[h["sels"][t]] if lens[t] else []
            res.append((kept, [sc[i - s0] for i in kept]))
        return h["packed"], res, _PassTables(area, np.ascontiguousarray(tabs["bbox"], dtype=np.int64))

    def process_tile_batch_hostloops(self, key: str, tiles: torch.Tensor, small_classes, class_thresholds: Dict[int, Tuple[float, float]],
                                     spatial_cfg: Optional[dict] = None, um_pix: float = 1.0, model_ids: Sequence[int] = (0,),
                                     dets: Optional[List[_Detections]] = None):
        """The per-tile unit of work of the headline metric: one batched forward for B independent tiles, then per
        tile the class loop (a6, a9, a11, a12), the cross-class dedup (a14, 0.7), the spatial constraints (a15) and the
        contour measurements (a17, a18).  Every kernel is launched ONCE for all tiles (segment-aware where the
        reference's loop carries state), so 256 CUs see hundreds of masks per launch instead of a few dozen.
        Same results as :meth:`process_tile_batch_unbatched`.  Returns per tile ``(packed, scores, classes, records)``."""
        ensemble = len(model_ids) > 1
        if ensemble:
            dets_per_model = [self._predict_batch(m, key, tiles) for m in model_ids]
            dets = dets_per_model[0]
        elif dets is None:
            dets = self._predict_batch(model_ids[0], key, tiles)
        self.ops.set_frame_width(int(tiles.shape[2]))
        T, dev = len(dets), self.dev
        # ---- class loop: one batched pass per class; what survives is gathered ONCE per class into `allp` (class-major
        # storage; every tile keeps its own index list in the reference's order: class by class, score order inside)
        passes = []
        for cls, (conf, iou_thr) in class_thresholds.items():
            if ensemble:
                big, res, calg = self._ensemble_class_pass_batched(dets_per_model, cls, small_classes, conf, iou_thr)
            else:
                big, res, calg = self._single_class_pass_batched_hostloops(dets, cls, small_classes, conf, iou_thr)
            if big is not None and any(len(k) for k, _ in res):
                passes.append((cls, big, res, calg))
        out = [(None, [], [], []) for _ in range(T)]
        total = sum(len(k) for _, _, res, _ in passes for k, _ in res)
        if total == 0:
            return out
        allp = torch.empty((total,) + tuple(passes[0][1].shape[1:]), dtype=passes[0][1].dtype, device=dev)
        tile_items: List[List[int]] = [[] for _ in range(T)]
        scores_all, classes_all, area_parts, bbox_parts = [], [], [], []
        off = 0
        for cls, big, res, calg in passes:
            src = [i for kept, _ in res for i in kept]
            pos = off
            for t, (kept, sc) in enumerate(res):
                tile_items[t].extend(range(pos, pos + len(kept)))
                pos += len(kept)
                scores_all.extend(sc)
            classes_all.extend([cls] * len(src))
            self.ops.gather_regions(big, src, calg.bbox[src], out=allp[off:off + len(src)])   # tight boxes of the class pass
            area_parts.append(calg.area[src])           # already reduced by the class pass
            bbox_parts.append(calg.bbox[src])
            off += len(src)
        # ---- cross-class dedup (a14) for all tiles: one contour launch, one pair-count launch -------------------
        area_all = np.concatenate(area_parts)
        bbox_all = np.concatenate(bbox_parts)
        alg = DeviceMaskAlgebra(self.ops, allp, area=area_all, bbox=bbox_all, blocks=tile_items)
        # traced ONCE: reused for the final measurements
        cset = self.ops.trace(allp, max_contours=256, bbox=alg._bbox_dev, total_area=int(area_all.sum()))
        per0 = cset.first_contour_perimeter()
        keep0_all, groups = [], []
        for t in range(T):
            k0 = []
            for idx in tile_items[t]:
                if alg.bbox[idx, 0] < 0:
                    continue
                per = per0[idx]
                if per > 0 and (4 * np.pi * int(alg.area[idx])) / (per ** 2) < 0.15:
                    continue
                k0.append(idx)
            keep0_all.append(k0)
            cl = [classes_all[i] for i in k0]
            for c in set(cl):
                g = [k0[i] for i in range(len(k0)) if cl[i] == c]
                if len(g) > 1:
                    groups.append(g)
        alg.prefetch_overlapping_pairs(groups)
        final_idx: List[List[int]] = []
        for t in range(T):
            k0 = keep0_all[t]
            scores = [scores_all[i] for i in k0]
            classes = [classes_all[i] for i in k0]
            bb = [(int(alg.bbox[i, 0]), int(alg.bbox[i, 2]), int(alg.bbox[i, 1]), int(alg.bbox[i, 3])) for i in k0]
            keep = self._dedup_smart_order(alg, k0, scores, classes, bb, 0.7) if k0 else []
            gl = [k0[i] for i in keep]
            sc, cl = [scores[i] for i in keep], [classes[i] for i in keep]
            if gl and spatial_cfg is not None and spatial_cfg.get("enabled", False):
                kk = apply_spatial_constraints_indices(alg.view(gl), sc, cl, spatial_cfg)
                gl, sc, cl = [gl[i] for i in kk], [sc[i] for i in kk], [cl[i] for i in kk]
            final_idx.append(gl)
            out[t] = (None, sc, cl, [])
        flat = [i for gl in final_idx for i in gl]
        if not flat:
            return out
        finalp = self.ops.gather_regions(allp, flat, alg.bbox[flat])
        recs = cset.records(um_pix=um_pix, measure=True, select=flat)
        pos = 0
        # pixel counts / tight boxes of the final masks, per tile (what an instance table needs; already reduced)
        self.last_batch_stats = [(alg.area[final_idx[t]], alg.bbox[final_idx[t]]) for t in range(T)]
        for t in range(T):
            n = len(final_idx[t])
            if n:
                out[t] = (finalp[pos:pos + n], out[t][1], out[t][2], recs[pos:pos + n])
            pos += n
        return out

    # ------------------------------------------------------------------ a8
    def calculate_average_mask_sizes(self, sample_images: Sequence[Tuple[str, torch.Tensor]]) -> Dict[int, float]:
        """``inference.py:1626-1706``: first predictor, first <= 5 images, detections with score >= 0.7."""
        sizes: Dict[int, List[int]] = {}
        for key, img in sample_images[:5]:
            det = self._predict_batch(0, key + "|full", img[None])[0]
            conf = det.scores >= 0.7
            if not conf.any():
                continue
            area, _ = self.ops.area_bbox(det.packed.contiguous())
            area = area.cpu().numpy()
            for a, c in zip(area[conf], det.classes[conf]):
                sizes.setdefault(int(c), []).append(int(a))
        return {c: float(np.mean(v)) for c, v in sizes.items() if v}


def run_inference(dataset_name, output_dir, visualize=True, threshold=0.65, draw_id=False, dataset_format="json",
                  draw_scalebar=False):
    """Drop-in for ``src/functions/inference.py:499`` (same signature, outputs and skip-image semantics)."""
    global_config = get_config()
    dataset_config = get_config(dataset_name=dataset_name)
    inf = dataset_config.get("inference_overrides", {}) or dataset_config.get("inference_settings", {})
    confidence_mode = inf.get("confidence_mode", "auto")
    class_specific_settings = inf.get("class_specific_settings", {})
    tile_cfg = inf.get("tile_settings", {})
    tile_size = tile_cfg.get("tile_size", 512)
    overlap_ratio = tile_cfg.get("overlap_ratio", 0.1)
    upscale_factor = tile_cfg.get("upscale_factor", 2.0)
    edge_filter_enabled = tile_cfg.get("edge_filter_enabled", True)
    ens = inf.get("ensemble_settings", {})
    gens = global_config.get("inference_settings", {}).get("ensemble_settings", {})
    ensemble_enabled = ens.get("enabled", gens.get("enabled", True))
    ensemble_small_only = ens.get("small_classes_only", gens.get("small_classes_only", True))
    classes_to_infer = inf.get("inference_settings", {}).get("classes_to_infer", None)
    split_dir = str(Path(global_config["paths"]["split_dir"]).expanduser().resolve())
    category_json = str(Path(global_config["paths"]["category_json"]).expanduser().resolve())

    dataset_info = read_dataset_info(category_json)
    register_datasets(dataset_info, dataset_name, dataset_format=dataset_format)
    DatasetCatalog.get(f"{dataset_name}_train")
    metadata = MetadataCatalog.get(f"{dataset_name}_train")
    num_classes = len(metadata.thing_classes)

    predictors, models = [], []
    for r in (50, 101):
        paths = get_trained_model_paths(split_dir, r)
        if dataset_name in paths:
            try:
                p, _ = choose_and_use_model(paths, dataset_name, threshold, metadata, r)
                if p is not None:
                    p.model.eval()
                    predictors.append(p)
                    models.append(r)
            except Exception as e:
                system_logger.warning(f"Failed to load R{r} model: {e}")
    if not predictors:
        raise FileNotFoundError(f"No trained models found for dataset '{dataset_name}'")
    system_logger.info(f"Loaded models: {', '.join(f'R{r}' for r in models)}")

    inpath = get_image_folder_path()
    os.makedirs(output_dir, exist_ok=True)
    images_name = [f for f in os.listdir(inpath) if is_image_file(f)]
    pipe = InferencePipeline(predictors, dataset_name, inf, global_config)
    dev = pipe.dev
    spatial_cfg = load_spatial_constraints(dataset_name)
    # Multi-GPU (one process per GPU): what is sharded over the ranks.  A FOLDER of images is sharded by image (image j ->
    # rank j % world, SURVEY 8(e) "batch": no cross-rank dependency until rank 0 collects the rows it writes); a single large
    # image -- or fewer images than would keep every rank busy -- by tile, with the per-image all-gather of instance tables
    # before the global duplicate / containment filters (`gather_and_merge`).  DEEPEMIA_SHARD=images|tiles overrides.
    job_rank, job_world = pipe.rank, pipe.world
    shard_images = False
    if job_world > 1:
        import torch.distributed as dist
        box = [images_name, os.environ.get("DEEPEMIA_SHARD", "auto")] if job_rank == 0 else [None, None]
        dist.broadcast_object_list(box, src=0)             # every rank walks rank 0's list in rank 0's order
        images_name, mode = box
        shard_images = mode == "images" or (mode != "tiles" and len(images_name) >= 2 * job_world)
        if shard_images:
            pipe.rank, pipe.world = 0, 1                   # every image is whole on its rank: full-image pass + all tiles, no exchange
        system_logger.info(f"Rank {job_rank}/{job_world}: sharding by {'image' if shard_images else 'tile'}")
    my_images = images_name[job_rank::job_world] if shard_images else images_name

    # decode on a helper thread, one image ahead of the GPU work (the reference preloads with a thread pool too,
    # inference.py:150-166); the first <= 5 decoded images are kept for the small-class statistics and reused
    from concurrent.futures import ThreadPoolExecutor

    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 2
    # (PNG / TIFF decode of a 2048^2 frame is 30-40 ms: two threads could not feed a 35-ms image loop)
    decoder = ThreadPoolExecutor(max_workers=max(2, min(8, ncpu // 2)))
    decoded: Dict[str, object] = {}

    def prefetch(name):
        if name not in decoded:
            decoded[name] = decoder.submit(imread_bgr, os.path.join(inpath, name))

    def load(name):
        # (the upload stays on this thread: copying from the helper threads on a stream of their own -- round 5 -- made the loop
        # 13 ms per image SLOWER, 43.5 against 30.5 ms, same box, and one of eight runs died; the 3 ms per image it would save
        # are not worth a second thread talking to the runtime)
        prefetch(name)
        img = decoded.pop(name).result()
        return None if img is None else torch.from_numpy(img).to(dev)

    sample = []
    if not shard_images or job_rank == 0:
        for name in images_name[:6]:
            prefetch(name)
        for name in images_name[:5]:
            t = load(name)
            if t is not None:
                sample.append((name, t))
    sample_dev = dict(sample)          # the first images are already on the device: no second decode

    rle_by_image: Dict[str, List[str]] = {}       # per image the EncodedPixels texts of its final instances (a16)
    rows_by_image: Dict[str, List[list]] = {}     # per image its measurement rows (a17-a19), measured while the image is still resident
    keep_masks = os.environ.get("DEEPEMIA_KEEP_MASKS", "0") == "1"
    dedup_results: Dict[str, dict] = {}
    processed = set()
    # The forwards of a GROUP of images are batched across the images and enqueued one group AHEAD of the post-processing
    # (`prefetch_images`): while the host walks the class loops, dedups and writers of group g, the network of group g + 1
    # runs on a stream of its own.  The reference goes image by image (inference.py:713-942); the per-image semantics (an
    # image that fails is logged and skipped) are kept -- a group only shares its forwards.
    tiles_per_image = 1
    if sample:
        tiles_per_image = max(1, len(pipe._tile_offsets(int(sample[0][1].shape[0]), int(sample[0][1].shape[1]), tile_size, overlap_ratio)))
    if shard_images:
        box = [tiles_per_image]
        dist.broadcast_object_list(box, src=0)
        tiles_per_image = box[0]
    group_size = max(1, min(int(os.environ.get("DEEPEMIA_IMAGE_GROUP", pipe.forward_batch // tiles_per_image)), 16))
    groups = [my_images[i:i + group_size] for i in range(0, len(my_images), group_size)]
    seen_full: Dict[tuple, bool] = {}
    if len(groups) >= 3:
        # a long folder: capture the forwards' graphs at the FIRST group (before the image loop's clock) instead of the second
        pipe.graph_after = 1
    for gn in groups[:3]:                  # decode up to three groups ahead of the image loop on the helper threads
        for nm in gn:
            if nm not in sample_dev:
                prefetch(nm)
    # (stream priorities were tried in round 5: the network stream at low priority 28.2 vs 26.8 ms per image, the loop's own stream at
    # high priority 26.3 vs 26.8 -- nothing to gain)
    net_stream = torch.cuda.Stream(device=dev)
    if os.environ.get("DEEPEMIA_NET_CU_MASK"):
        # (experiment switch) the network's stream on a CU mask, e.g. mod:32:28 = 28 of every 32 compute units
        net_stream, n_cu = _L.cu_masked_stream(dev, os.environ["DEEPEMIA_NET_CU_MASK"])
        system_logger.info(f"Network stream restricted to {n_cu} of 256 compute units ({os.environ['DEEPEMIA_NET_CU_MASK']})")

    def enqueue_forwards(ok, model_ids):
        up = torch.cuda.Event()
        up.record(torch.cuda.current_stream(dev))
        try:
            with torch.cuda.stream(net_stream):
                net_stream.wait_event(up)
                return pipe.prefetch_images(ok, model_ids, tile_size, overlap_ratio, upscale_factor)
        except Exception as e:      # e.g. out of memory on the batched forward: the per-image passes run their own forwards
            system_logger.warning(f"Batched forwards of images {[nm for nm, _ in ok]} failed ({e}); falling back to per-image forwards")
            return None

    def launch_group(gnames, model_ids):
        """Decode / upload the group's images (the decode of the group after it is already running on the helper threads) and
        enqueue its forwards on the network stream.  Returns ({name: device image or None}, prefetch plan)."""
        items = {}
        for nm in gnames:
            items[nm] = sample_dev.pop(nm) if nm in sample_dev else load(nm)
        ok = [(nm, t) for nm, t in items.items() if t is not None]
        if ok and 0 < len(ok) < group_size and seen_full.get(tuple(ok[-1][1].shape)):
            # a SHORT group (the folder's last images, or one with an unreadable file) goes through the network at the full group's
            # batch size, padded with repeats of its last image: the captured graphs of that shape are replayed instead of a new
            # shape's eager forwards, arena and -- with three shapes cached -- an eviction (0.5-0.9 s per rank in the round-5 logs
            # against ~50 ms of padded network).  Results do not depend on the batch (per-image scale groups): the pads are dropped.
            pad = [(f"\x00pad{k}\x00{ok[-1][0]}", ok[-1][1]) for k in range(group_size - len(ok))]
            plan = enqueue_forwards(ok + pad, model_ids)
            return items, plan
        if len(ok) == group_size:
            seen_full[tuple(ok[-1][1].shape)] = True
        return items, (enqueue_forwards(ok, model_ids) if ok else None)

    def finish_group(plan, g):
        try:
            pipe.finish_prefetch(plan)          # the wait for the group's forwards
        except Exception as e:
            system_logger.warning(f"Batched forwards of group {g} failed ({e}); falling back to per-image forwards")
        for ck in [ck for ck in pipe._cache if ck[1].startswith("\x00pad")]:      # (the pads of a short group)
            del pipe._cache[ck]

    # group 0 goes through the first model before the small-class statistics: they read its full-image passes (inference.py:1626-1706)
    ahead = launch_group(groups[0], [0]) if groups else None
    if ahead is not None:
        finish_group(ahead[1], 0)
    if job_world > 1:
        # every rank must walk the class loop with the SAME small classes: rank 0 (the one that runs the full-image passes the
        # statistics read) decides, the others take its answer -- a rank that could not read one of the first images would
        # otherwise compute its own
        box = [sorted(determine_small_classes(pipe.calculate_average_mask_sizes(sample), 50))] if job_rank == 0 else [None]
        dist.broadcast_object_list(box, src=0)
        small_classes = set(box[0])
    else:
        small_classes = determine_small_classes(pipe.calculate_average_mask_sizes(sample), 50)
    system_logger.info(f"Small classes: {sorted(small_classes)}")
    if shard_images:
        for nm in [nm for nm in sample_dev if nm not in my_images]:       # statistics images another rank owns
            sample_dev.pop(nm)
            pipe.drop_cached(nm)
    sample = []
    t_all = time.perf_counter()
    targets_all = list(range(num_classes) if classes_to_infer is None else [c for c in classes_to_infer if c < num_classes])
    any_ens = len(predictors) > 1 and any(ensemble_enabled and (not ensemble_small_only or c in small_classes) for c in targets_all)
    models_needed = list(range(len(predictors))) if any_ens else [0]
    if ahead is not None:
        ok0 = [(nm, t) for nm, t in ahead[0].items() if t is not None]
        ahead = (ahead[0], enqueue_forwards(ok0, models_needed) if (ok0 and len(models_needed) > 1) else None)

    def prefetch_group(gnames):
        for nm in gnames:
            if nm not in sample_dev:
                prefetch(nm)

    def process_image(gi, name, image_dev, t0):
        """One image through the class loop, the merges, the cross-class pass, the constraints, RLE and its measurement rows
        (reference inference.py:735-931 + 1030-1291 for this image); a failure is logged and the image skipped."""
        if image_dev is None:
            system_logger.warning(f"Could not load image: {name}")
            if pipe.world > 1:
                # the other ranks may have loaded it: take part in the image's exchange with status 1, so that every rank skips it
                try:
                    pipe.gather_and_merge({}, (0, 0), {}, status=1)
                except PeerImageFailure as e:
                    system_logger.error(f"Error processing image {name}: {e}")
            return
        try:
            image_host = None
            parts, all_scores, all_classes = [], [], []
            locals_by_class, ens_by_class, local_err = {}, {}, None
            targets = range(num_classes) if classes_to_infer is None else [c for c in classes_to_infer if c < num_classes]
            fast = {}
            if phases_ok:
                for target_class in targets:
                    is_small = target_class in small_classes
                    ccfg = class_specific_settings.get(f"class_{target_class}", {})
                    use_ens = ensemble_enabled and (not ensemble_small_only or is_small)
                    if (use_ens and len(predictors) > 1) or pipe.uses_multiscale(target_class):
                        fast = None
                        break
                    if confidence_mode == "manual":
                        conf = ccfg.get("confidence_threshold", 0.3 if is_small else 0.5)
                    else:
                        if image_host is None:
                            image_host = image_dev.cpu().numpy()
                        conf = get_confidence_threshold(image_host, target_class, small_classes, global_config)
                    fast[target_class] = (conf, ccfg.get("iou_threshold", 0.5 if is_small else 0.7))
            if fast:
                # every class goes through the one model's standard passes: all classes in phases, two waits (tile_pipeline_all_classes)
                by_class = pipe.tile_pipeline_all_classes(name, image_dev, fast, small_classes, tile_size, overlap_ratio, upscale_factor, edge_filter_enabled)
                for target_class in targets:
                    m, s, c = by_class[target_class]
                    if m is not None and m.shape[0]:
                        parts.append(m)
                        all_scores.extend(s)
                        all_classes.extend(c)
                targets = []
            for target_class in targets:
                is_small = target_class in small_classes
                ccfg = class_specific_settings.get(f"class_{target_class}", {})
                if confidence_mode == "manual":
                    conf = ccfg.get("confidence_threshold", 0.3 if is_small else 0.5)
                else:
                    if image_host is None:
                        image_host = image_dev.cpu().numpy()
                    conf = get_confidence_threshold(image_host, target_class, small_classes, global_config)
                iou_thresh = ccfg.get("iou_threshold", 0.5 if is_small else 0.7)
                use_ens = ensemble_enabled and (not ensemble_small_only or is_small)
                model_ids = list(range(len(predictors))) if (use_ens and len(predictors) > 1) else [0]
                if pipe.world > 1:
                    # local passes only; ONE all-gather per image after the class loop.  A failure of THIS rank's passes
                    # (a kernel error, out of memory on its tiles) must not make it skip that all-gather: it takes part with
                    # an empty table and status 1, and every rank skips the image together (PeerImageFailure below)
                    if local_err is None:
                        try:
                            locals_by_class[target_class] = pipe._tile_pipeline_local(model_ids, name, image_dev, target_class, small_classes, conf,
                                                                                      tile_size, overlap_ratio, upscale_factor, iou_thresh,
                                                                                      edge_filter_enabled)
                            ens_by_class[target_class] = len(model_ids) > 1
                        except Exception as e:
                            system_logger.error(f"Rank {pipe.rank}: local passes of image {name} failed: {e}", exc_info=True)
                            local_err = e
                    continue
                m, s, c = pipe.tile_based_inference_pipeline(model_ids, name, image_dev, target_class, small_classes, conf,
                                                             tile_size, overlap_ratio, upscale_factor, iou_thresh,
                                                             edge_filter_enabled)
                if m is not None and m.shape[0]:
                    parts.append(m)
                    all_scores.extend(s)
                    all_classes.extend(c)
            if pipe.world > 1:
                merged = pipe.gather_and_merge({} if local_err is not None else locals_by_class, (int(image_dev.shape[0]), int(image_dev.shape[1])),
                                               ens_by_class, status=0 if local_err is None else 1)
                for target_class in targets:
                    r = merged[target_class]
                    if isinstance(r, Exception):
                        raise r               # the reference raises inside the class loop and skips the image (N4)
                    m, s, c = r
                    if m is not None and m.shape[0]:
                        parts.append(m)
                        all_scores.extend(s)
                        all_classes.extend(c)
            packed = torch.cat(parts, dim=0) if parts else None
            pipe.ops.set_frame_width(int(image_dev.shape[1]))
            constrained = bool(spatial_cfg and spatial_cfg.get("enabled", False)) and os.environ.get("DEEPEMIA_ALL_PAIRS", "1") == "1"   # (A/B switch)
            packed, scores, classes, tabs = pipe.deduplicate_masks_smart(packed, all_scores, all_classes, iou_threshold=0.7, with_tables=True,
                                                                        all_pairs=constrained)
            if packed is not None and packed.shape[0]:
                alg = DeviceMaskAlgebra(pipe.ops, packed, area=tabs[0], bbox=tabs[1])       # (pixel counts / boxes: already on the host)
                if tabs[2] is not None:
                    alg.preload(tabs[2])        # ... and every pair's intersection: the constraints below launch and wait for nothing
                keep = apply_spatial_constraints_indices(alg, scores, classes, spatial_cfg)
                if len(keep) != int(packed.shape[0]):
                    packed = pipe.ops.gather_regions(packed, keep, tabs[1][keep])
                    tabs = (tabs[0][keep], tabs[1][keep])
                scores, classes = [scores[i] for i in keep], [classes[i] for i in keep]
            n_final = 0 if packed is None else int(packed.shape[0])
            result = {"masks": packed, "scores": scores, "classes": classes, "hw": (int(image_dev.shape[0]), int(image_dev.shape[1])),
                      "area": None if tabs is None else tabs[0], "bbox": None if tabs is None else tabs[1]}
            # a16: one crop launch + one native call for the image's EncodedPixels texts; where this rank also measures the image, the
            # cropped words come to the host in the SAME copy as the contour tables (one wait for RLE + measurements)
            texts = []
            if shard_images or job_rank == 0:
                # the measurement phase of this image (inference.py:1030-1291) while its masks and pixels are still resident: the
                # reference walks the folder a second time after the image loop, which gives the same rows; done here, a folder
                # of any length holds the masks of ONE image group at a time, and a rank measures the images it owns
                crop = rle_crop_launch(pipe.ops, packed, tabs[0], tabs[1]) if n_final else None
                extra = [crop[0]] if crop is not None else None
                rows_by_image[name] = measure_image(pipe.ops, name, result, inpath, output_dir, metadata, dataset_name, draw_scalebar,
                                                    visualize, image_dev=image_dev, extra=extra)
                if crop is not None:
                    texts = rle_text_from_payload(extra[0], crop[1], crop[2], int(packed.shape[1]))
            elif n_final:
                texts = rle_text_packed(pipe.ops, packed, area=tabs[0], bbox=tabs[1])
            if not keep_masks:
                result["masks"] = None
            dedup_results[name] = result
            rle_by_image[name] = texts
            processed.add(name)
            system_logger.info(f"Image {name}: {n_final} instances in {time.perf_counter() - t0:.2f}s")
        except Exception as e:  # reference semantics: log, skip the image, continue (inference.py:928-931)
            system_logger.error(f"Error processing image {name}: {e}", exc_info=True)
        finally:
            pipe.drop_cached(name)
            log_memory_usage(f"After image {gi + 1}/{len(my_images)}: {name}")
    # DEEPEMIA_IMAGE_THREADS=k (experiment switch, default 1): the images of a group are post-processed by k host threads at once
    # (every thread its own MaskOps; forwards stay on this thread) -- their device-to-host waits then overlap instead of queueing
    # one after the other behind the next group's convolution grids
    phases_ok = pipe.world == 1 and pipe.merge_mode == "smart" and os.environ.get("DEEPEMIA_IMAGE_PHASES", "1") == "1"     # (A/B switch)
    image_threads = max(1, int(os.environ.get("DEEPEMIA_IMAGE_THREADS", "1"))) if pipe.world == 1 else 1
    image_pool = ThreadPoolExecutor(max_workers=image_threads) if image_threads > 1 else None

    def process_image_on_thread(gi, name, image_dev, t0):
        torch.cuda.set_device(dev)
        process_image(gi, name, image_dev, t0)

    flat = [(g, nm) for g, gn in enumerate(groups) for nm in gn]
    cur_group, cur_items = -1, {}
    pending = []
    for gi, (g, name) in enumerate(flat):
        t0 = time.perf_counter()
        log_memory_usage(f"Before image {gi + 1}/{len(my_images)}: {name}")
        if g != cur_group:
            for f_ in pending:                      # (image threads: a group's images are done before the next group's forwards are waited for)
                f_.result()
            pending = []
            cur_group = g
            cur_items, plan = ahead
            tg0 = time.perf_counter()
            finish_group(plan, g)
            tg1 = time.perf_counter()
            if g + 3 < len(groups):
                prefetch_group(groups[g + 3])       # decode three groups ahead on the helper threads
            ahead = launch_group(groups[g + 1], models_needed) if g + 1 < len(groups) else None
            system_logger.debug(f"Group {g}: waited {1e3 * (tg1 - tg0):.1f} ms for its forwards, next group loaded and enqueued in "
                                f"{1e3 * (time.perf_counter() - tg1):.1f} ms")
        image_dev = cur_items.pop(name, None)
        if image_pool is not None:
            pending.append(image_pool.submit(process_image_on_thread, gi, name, image_dev, t0))
        else:
            process_image(gi, name, image_dev, t0)
    for f_ in pending:
        f_.result()
    if image_pool is not None:
        image_pool.shutdown(wait=True)
    pipe.clear_cache()
    decoder.shutdown(wait=False)
    total = time.perf_counter() - t_all
    system_logger.info(f"Inference complete: {len(processed)}/{len(my_images)} images, avg "
                       f"{total / max(len(my_images), 1):.4f}s/image, {pipe.forward_calls} batched forwards")
    if shard_images:
        # the ONE exchange of the image-sharded job: every rank's rows and texts go to rank 0, which writes the files
        mine = (rle_by_image, rows_by_image, {k: {kk: vv for kk, vv in v.items() if kk != "masks"} for k, v in dedup_results.items()}, sorted(processed))
        gathered = [None] * job_world
        dist.all_gather_object(gathered, mine)         # (an all-gather: the object collective every backend, RCCL included, supports)
        if job_rank == 0:
            for r_rle, r_rows, r_res, r_done in gathered[1:]:
                rle_by_image.update(r_rle)
                rows_by_image.update(r_rows)
                for k, v in r_res.items():
                    dedup_results.setdefault(k, dict(v, masks=None))
                processed.update(r_done)
    unprocessed = set(images_name) - processed
    if unprocessed and (job_rank == 0 or not shard_images):
        system_logger.warning(f"Unprocessed images: {sorted(unprocessed)}")
    if job_rank != 0:
        return dedup_results     # rank 0 alone writes the output files (tile sharding: every rank holds the same merged result)
    try:     # rank 0's host-side output section: no collective below this line
        with open(os.path.join(output_dir, "R50_flip_results.csv"), "w", newline="") as f:
            wri = csv.writer(f)
            wri.writerow(["ImageId", "EncodedPixels"])
            for name in images_name:               # (the order the reference's image loop appends in, inference.py:917-925)
                for text in rle_by_image.get(name, []):
                    wri.writerow([name.rsplit(".", 1)[0], text])

        write_measurements(pipe.ops, dedup_results, inpath, output_dir, metadata, dataset_name, draw_scalebar, visualize,
                           rows_by_image=rows_by_image)
        with open(os.path.join(output_dir, "class_color_legend.txt"), "w") as f:
            f.write("Class Color Legend (BGR)\n")
            for i, cname in enumerate(metadata.thing_classes):
                f.write(f"Class {i} ({cname}): {CLASS_COLORS[i % len(CLASS_COLORS)]}\n")
    except Exception as e:
        if job_world > 1:
            raise OutputWriteError(f"writing the outputs failed on rank 0: {e}") from e
        raise
    return dedup_results


def _polyline_pixels(pts: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Pixels of the closed polyline through ``pts`` [n, 2] (x, y): ``cv2.drawContours(..., thickness=1)`` on a
    CHAIN_APPROX_SIMPLE contour, whose segments all run in one of the eight chain directions (every pixel of such a
    segment is on the integer lattice, so any line rasteriser sets the same pixels)."""
    xs, ys = [], []
    n = len(pts)
    for k in range(n):
        (xa, ya), (xb, yb) = pts[k], pts[(k + 1) % n]
        steps = int(max(abs(int(xb) - int(xa)), abs(int(yb) - int(ya))))
        t = np.arange(steps + 1)
        xs.append(np.rint(xa + (int(xb) - int(xa)) * t / max(steps, 1)).astype(np.int64))
        ys.append(np.rint(ya + (int(yb) - int(ya)) * t / max(steps, 1)).astype(np.int64))
    if not xs:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    return np.concatenate(xs), np.concatenate(ys)


def write_predictions_png(path: str, image_bgr: np.ndarray, crops, classes: Sequence[int], contours, thing_classes) -> List[Tuple[int, int, int, int]]:
    """``<img>_predictions.png`` (``inference.py:1080-1145``), mask by mask IN ORDER as the reference draws it: a 50 % colour
    overlay (``cv2.addWeighted(vis, 1.0, coloured_mask, 0.5, 0)``: inside the mask ``saturate(round(v + 0.5 colour))``, ties to
    even), the external contours in the class colour (``drawContours`` thickness 1), then the instance number and the class
    name at the centroid ``(int(m10 / m00), int(m01 / m00))`` -- so a later mask blends over an earlier one's outline and
    text.  Overlay and outlines are pixel-exact restatements (checked against ``oracle/pipeline_ref.py::overlay_without_text``);
    the TEXT is drawn with Pillow's font, not OpenCV's Hershey strokes: the boxes it may touch are returned so that a
    checker can leave them out.  ``crops`` = ``mask_crops(...)``: masks arrive as bounding-box crops, never as dense frames."""
    from PIL import Image, ImageDraw

    vis = np.ascontiguousarray(image_bgr[:, :, :3]).copy()
    H, W = vis.shape[:2]
    text_boxes: List[Tuple[int, int, int, int]] = []
    probe = ImageDraw.Draw(Image.new("RGB", (8, 8)))
    for i, (c, cls, recs) in enumerate(zip(crops, classes, contours)):
        color = CLASS_COLORS[int(cls) % len(CLASS_COLORS)]                       # BGR, as the reference's class_colors
        if c is not None:
            y0, x0, sub = c
            win = vis[y0:y0 + sub.shape[0], x0:x0 + sub.shape[1]]
            win[sub] = np.clip(np.rint(win[sub].astype(np.float32) + 0.5 * np.asarray(color, dtype=np.float32)), 0, 255).astype(np.uint8)
        for rec in recs:
            px, py = _polyline_pixels(np.asarray(rec["points"]).reshape(-1, 2))
            ok = (px >= 0) & (px < W) & (py >= 0) & (py < H)
            vis[py[ok], px[ok]] = color
        if c is None:
            continue
        ys, xs = np.nonzero(c[2])
        if len(ys) == 0:
            continue
        cx, cy = int((xs.sum() + c[1] * len(xs)) / len(xs)), int((ys.sum() + c[0] * len(ys)) / len(ys))
        cname = thing_classes[int(cls)] if int(cls) < len(thing_classes) else f"class_{int(cls)}"
        for text, (tx, ty) in ((f"{i + 1}", (cx, cy - 18)), (cname, (cx, cy + 6))):
            l, t, r, b = probe.textbbox((tx, ty), text)
            l, t, r, b = max(l - 1, 0), max(t - 1, 0), min(r + 1, W), min(b + 1, H)
            if r <= l or b <= t:
                continue
            patch = Image.fromarray(np.ascontiguousarray(vis[t:b, l:r, ::-1]))
            ImageDraw.Draw(patch).text((tx - l, ty - t), text, fill=(255, 255, 255))
            vis[t:b, l:r] = np.asarray(patch)[:, :, ::-1]
            text_boxes.append((l, t, r, b))
    Image.fromarray(np.ascontiguousarray(vis[:, :, ::-1])).save(path)
    return text_boxes


def measurement_rows(image_name: str, classes: Sequence[int], recs, thing_classes, min_area: float, contrast=None, psum: str = "0") -> List[list]:
    """The 20-column rows of ``measurements_results.csv`` for one image / tile (``inference.py:1148-1230``): one row per
    external contour that passes the area gate, ``<image>_<instance>`` ids counted from 1, ``None`` -> empty field."""
    rows = []
    for instance_id, (cls, contours) in enumerate(zip(classes, recs), 1):
        cls = int(cls)
        d10, d50, d90 = contrast[instance_id - 1] if contrast is not None else (None, None, None)
        cname = thing_classes[cls] if cls < len(thing_classes) else f"class_{cls}"
        for c in contours:
            if c["area"] < min_area:
                continue
            v = c["values"]
            rows.append([f"{image_name}_{instance_id}", cls, cname, float(v[0]), float(v[1]), float(v[2]), float(v[3]),
                         float(v[4]), float(v[5]), float(v[6]), float(v[7]), float(v[8]), float(v[9]), float(v[10]),
                         float(v[11]), d10, d50, d90, psum, image_name])
    return rows


def measurement_csv_text(tiles, thing_classes, min_area: float, psum: str = "0") -> str:
    """The text ``csv.writer`` produces for the measurement rows of many images / tiles (``tiles``: (image name, classes,
    records) each), byte for byte -- 20 columns, ``\r\n`` row ends, floats as ``repr()`` -- with the 12 float columns of ALL
    rows formatted by one native call (``demia_host_repr_rows``: CPython's shortest-repr layout restated; 2700 rows are
    21 ms through csv.writer, 4 ms here).  Names that would need quoting fall back to csv.writer itself."""
    import ctypes as C
    import io
    heads, tails, vals = [], [], []
    for name, classes, recs in tiles:
        for instance_id, (cls, contours) in enumerate(zip(classes, recs), 1):
            cls = int(cls)
            cname = thing_classes[cls] if cls < len(thing_classes) else f"class_{cls}"
            for c in contours:
                if c["area"] < min_area:
                    continue
                heads.append(f"{name}_{instance_id},{cls},{cname},")
                tails.append(f",,,,{psum},{name}\r\n")
                vals.append(c["values"])
    if not heads:
        return ""
    if any(ch in h or ch in t[4:-2] for h, t in zip(heads, tails) for ch in ('"', "\r", "\n")) or any(h.count(",") != 3 or t.count(",") != 5 for h, t in zip(heads, tails)):
        buf = io.StringIO()                                   # a name with a delimiter / quote in it: csv.writer's own quoting
        w = csv.writer(buf)
        for name, classes, recs in tiles:
            for r in measurement_rows(name, classes, recs, thing_classes, min_area, None, psum):
                w.writerow(r)
        return buf.getvalue()
    v = np.ascontiguousarray(np.stack(vals), dtype=np.float64)
    out = C.create_string_buffer(v.size * 26 + v.shape[0] + 16)
    n = _L.load().demia_host_repr_rows(v.ctypes.data, v.shape[0], v.shape[1], out, len(out))
    if n < 0:
        raise _L.HipKernelError("demia_host_repr_rows: buffer too small")
    floats = out.raw[:n].decode("ascii").split("\n")
    return "".join(h + f + t for h, f, t in zip(heads, floats, tails))


def measure_image(ops: MaskOps, test_img: str, data: dict, test_img_path: str, output_dir: str, metadata, dataset_name: str,
                  draw_scalebar: bool = False, visualize: bool = False, image_dev: Optional[torch.Tensor] = None,
                  extra: Optional[list] = None) -> List[list]:
    """The measurement phase of ONE image (``inference.py:1030-1291``): scale bar, contours + the 12 measurements of every final
    mask, optional contrast percentiles, optional overlay / scale-bar debug images; returns the image's CSV rows.
    ``image_dev``: the decoded image when the caller still holds it on the device (the image loop does: the reference re-reads
    the file here, which gives the same bytes)."""
    measure_contrast = bool(get_config().get("measure_contrast_distribution", False))     # inference.py:58: GLOBAL config
    host = {}

    def im_host():
        if "im" not in host:
            host["im"] = image_dev.cpu().numpy() if image_dev is not None else imread_bgr(os.path.join(test_img_path, test_img))
        return host["im"]

    # N8: the values of the measurement phase's own call (inference.py:1046-1057) are the ones the CSV uses
    im = im_host() if (draw_scalebar or scale_bar_needs_image(dataset_name)) else None
    if draw_scalebar and im is not None:
        debug_image = im.copy()
        psum, um_pix = detect_scale_bar(debug_image, roi_config=None, dataset_name=dataset_name, draw_debug=True)
        debug_path = os.path.join(output_dir, f"{test_img}_scalebar_debug.png")
        if not os.path.exists(debug_path):
            from PIL import Image
            Image.fromarray(np.ascontiguousarray(debug_image[..., ::-1])).save(debug_path)
            system_logger.info(f"Saved scalebar debug visualization to {debug_path}")
        del debug_image
    else:
        psum, um_pix = detect_scale_bar(im, roi_config=None, dataset_name=dataset_name)
    packed, classes = data["masks"], data["classes"]
    if packed is None or packed.shape[0] == 0:
        return []
    h, wd = data["hw"]
    ops.set_frame_width(wd)
    min_area = max(5, h * wd * 0.000005 * 0.05)
    if data.get("bbox") is not None:       # pixel counts / tight boxes already on the host (the image loop): no reduction, no wait for it
        # (``extra``: a list of int32 device tensors the image loop wants on the host in the same copy -- the cropped words of the RLE
        # texts --; replaced in place by their host arrays)
        recs = ops.contours(packed, max_contours=256, um_pix=um_pix, bbox=ops.upload(np.ascontiguousarray(data["bbox"], dtype=np.int32)),
                            total_area=int(np.sum(data["area"])), extra=extra)
        if extra is not None:
            recs, extra_host = recs
            extra[:] = extra_host
    else:
        recs = ops.contours(packed, max_contours=256, um_pix=um_pix)
        if extra is not None:
            extra[:] = [e.cpu().numpy() for e in extra]
    contrast = [(None, None, None)] * int(packed.shape[0])
    if measure_contrast:
        # measurements.py:195-215: gray levels under the whole instance mask; the histogram is a device reduction
        dev_im = image_dev if image_dev is not None else (None if im_host() is None else torch.from_numpy(im_host()).to(ops.device))
        if dev_im is not None:
            hist = ops.gray_histogram(packed, dev_im)
            contrast = [contrast_percentiles(hh) for hh in hist]
    if visualize and im_host() is not None:
        write_predictions_png(os.path.join(output_dir, f"{test_img}_predictions.png"), im_host(), mask_crops(ops, packed),
                              classes, recs, metadata.thing_classes)
    return measurement_rows(test_img, classes, recs, metadata.thing_classes, min_area, contrast, psum)


def write_measurements(ops: MaskOps, dedup_results: Dict[str, dict], test_img_path: str, output_dir: str, metadata,
                       dataset_name: str, draw_scalebar: bool = False, visualize: bool = False,
                       rows_by_image: Optional[Dict[str, List[list]]] = None) -> str:
    """Measurement phase (``inference.py:983-1291``): one CSV row per external contour that passes
    the area gate, 20 columns, ``None`` -> empty field, floats through ``csv.writer``.  ``rows_by_image``: rows the image
    loop (or another rank, when images are sharded over ranks) has already measured; images without an entry are measured here."""
    csv_filename = os.path.join(output_dir, "measurements_results.csv")
    with open(csv_filename, "w", newline="") as csvfile:
        w = csv.writer(csvfile)
        w.writerow(CSV_HEADER)
        for test_img in [f for f in os.listdir(test_img_path) if is_image_file(f)]:
            data = dedup_results.get(test_img)
            if data is None:
                continue   # skipped image: the reference finds no masks for it (dedup_results.get(..., {}))
            if rows_by_image is not None and test_img in rows_by_image:
                rows = rows_by_image[test_img]
            else:
                rows = measure_image(ops, test_img, data, test_img_path, output_dir, metadata, dataset_name, draw_scalebar, visualize)
            for r in rows:
                w.writerow(r)
            csvfile.flush()
    system_logger.info(f"Measurements complete. Results: {csv_filename}")
    return csv_filename
