"""ORACLE (test infrastructure, never shipped): dense-numpy restatement of the reference's
per-image orchestration -- ``run_inference`` class loop (``src/functions/inference.py:789-868``),
``tile_based_inference_pipeline`` (``2299-2485``), ``run_class_specific_inference`` (``1353-1461``),
``run_ensemble_inference`` (``1464-1598``), the measurement phase (``1148-1230``) -- on top of
``oracle/maskrcnn_ref.py`` (predictor) and ``oracle/postproc_ref.py`` (everything after it).
Parity unpinned as a whole (the reference cannot be imported here: cv2 / detectron2 missing);
its parts are pinned as their own headers say.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import maskrcnn_ref as NN
from . import postproc_ref as P


def cv_resize_linear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), INTER_LINEAR) on uint8 HxWxC: OpenCV's fixed-point path
    (11-bit coefficients; vertical: (((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2)."""
    h, w = img.shape[:2]

    def tables(n_in, n_out, vertical):
        scale = 1.0 / (float(n_out) / float(n_in))
        d = np.arange(n_out, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if vertical:
            s0, s1 = np.clip(s, 0, n_in - 1), np.clip(s + 1, 0, n_in - 1)
        else:
            f[s < 0] = 0.0
            s[s < 0] = 0
            f[s >= n_in - 1] = 0.0
            s[s >= n_in - 1] = n_in - 1
            s0, s1 = s, np.minimum(s + 1, n_in - 1)
        c0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
        return s0, s1, c0, c1

    x0, x1, a0, a1 = tables(w, out_w, False)
    y0, y1, b0, b1 = tables(h, out_h, True)
    src = img.astype(np.int64)
    hor = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]
    r0, r1 = hor[y0], hor[y1]
    v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


class RefPipeline:
    def __init__(self, state_dicts: Dict[int, dict], num_classes: int, threshold: float, inf_settings: dict,
                 global_inf_settings: dict, parallel_mask_processing: bool = True):
        self.models = [(d, state_dicts[d]) for d in (50, 101) if d in state_dicts]
        self.K = num_classes
        self.threshold = threshold
        self.inf = inf_settings
        self.weights = list(global_inf_settings.get("ensemble_settings", {}).get("weights", {"R50": 0.6, "R101": 0.4}).values())
        self.class_settings = inf_settings.get("class_specific_settings", {})
        self.parallel = parallel_mask_processing
        self.global_config = {"inference_settings": global_inf_settings}     # what get_confidence_threshold reads (inference.py:302, 352)
        self._cache = {}
        self.forward_calls = 0
        # f4, the product's flagged NON-parity modes (checker twins; neither exists in the reference's live path)
        self.merge_mode = str(inf_settings.get("merge_mode", "smart"))
        snm = inf_settings.get("soft_nms", {}) or {}
        self.soft_nms_sigma, self.soft_nms_thr = float(snm.get("sigma", 0.5)), float(snm.get("score_threshold", 0.001))
        self.multiscale_enabled = bool((inf_settings.get("multiscale_settings", {}) or {}).get("enabled", False))
        self.scales_visited: Dict[tuple, list] = {}

    def predict(self, mi: int, key, image: np.ndarray):
        ck = (mi, key)
        if ck not in self._cache:
            d, sd = self.models[mi]
            o = NN.predict(image, sd, d, self.threshold)
            self.forward_calls += 1
            self._cache[ck] = (o["pred_masks"].numpy(), o["scores"].numpy(), o["pred_classes"].numpy())
        return self._cache[ck]

    # inference.py:1353-1461 / 1464-1598
    def class_pass(self, model_ids, key, image, target_class, small_classes, conf, iou_threshold):
        hw = image.shape[:2]
        if len(model_ids) == 1:
            m, s, c = self.predict(model_ids[0], key, image)
            return P.single_model_class_pass(m, s, c, hw, target_class, small_classes, conf, iou_threshold,
                                             self.class_settings, self.parallel)
        all_masks, all_scores = [], []
        is_small = target_class in small_classes
        for mi, weight in zip(model_ids, self.weights):
            m, s, c = self.predict(mi, key, image)
            if len(s) == 0:
                continue
            sel = (c == target_class) & (s >= conf)
            for mask, score in zip(m[sel], s[sel]):
                cleaned = P.postprocess_masks_universal(np.array([mask]), hw, is_small)
                if cleaned:
                    all_masks.append(cleaned[0])
                    all_scores.append(float(score) * weight)
        if len(all_masks) == 0:
            return "EMPTY_NDARRAY", [], []
        return P.deduplicate_masks_smart(all_masks, all_scores, [target_class] * len(all_masks), iou_threshold)

    # inference.py:1833-2064 as the product composes it (one class pass per scale instead of the iterative masking loop)
    def multiscale_pass(self, model_ids, key, image, target_class, small_classes, conf, iou_threshold):
        h, w = image.shape[:2]
        is_small = target_class in small_classes
        base_min = max(3, int(h * w * 0.000005)) if is_small else max(25, int(h * w * 0.0001))

        def one(scale):
            sh, sw = (h, w) if scale == 1.0 else (int(h * scale), int(w * scale))
            im = image if scale == 1.0 else cv_resize_linear_u8(image, sh, sw)
            m, s, _ = self.class_pass(model_ids, (key, "full") if scale == 1.0 else (key, "scale", scale), im, target_class,
                                      small_classes, conf, iou_threshold)
            if isinstance(m, str):
                return [], []
            keep = [i for i in range(len(m)) if int((np.asarray(m[i]) > 0).sum()) >= int(base_min * scale ** 2)]
            return ([P.resize_nearest((np.asarray(m[i]) > 0).astype(np.uint8), h, w).astype(bool) if scale != 1.0 else np.asarray(m[i]) > 0
                     for i in keep], [s[i] for i in keep])

        per, order = {}, [0.7, 1.0, 1.5]
        for sc in order:
            per[sc] = one(sc)
        base = len(per[1.0][1])
        for unlocked, extra in ((len(per[1.5][1]) > base * 0.1, (2.0, 2.5)), (len(per[0.7][1]) > base * 0.1, (0.5, 0.6))):
            if unlocked:
                for sc in extra:
                    r = one(sc)
                    if len(r[1]) < base * 0.05:
                        break
                    per[sc] = r
                    order.append(sc)
        self.scales_visited[(key, target_class)] = list(order)
        rm, rs = P.multiscale_merge(per, order)
        return rm, rs, [target_class] * len(rs)

    # inference.py:2299-2485
    def tile_pipeline(self, model_ids, key, image, target_class, small_classes, conf, tile_size, overlap, upscale,
                      iou_threshold, edge_filter=True):
        h, w = image.shape[:2]
        if self.multiscale_enabled and self.class_settings.get(f"class_{target_class}", {}).get("use_multiscale", False):
            fm, fs, fc = self.multiscale_pass(model_ids, key, image, target_class, small_classes, conf, iou_threshold)
        else:
            fm, fs, fc = self.class_pass(model_ids, (key, "full"), image, target_class, small_classes, conf, iou_threshold)
        tiles = P.generate_tiles_with_overlap(image, tile_size, overlap)
        tm_all, ts_all, tc_all = [], [], []
        for ti, (tile, x_off, y_off) in enumerate(tiles):
            th, tw = tile.shape[:2]
            uh, uw = int(th * upscale), int(tw * upscale)
            up = tile if (uh, uw) == (th, tw) else cv_resize_linear_u8(tile, uh, uw)
            tm, ts, tc = self.class_pass(model_ids, (key, "tile", ti, tile_size, overlap, upscale), up, target_class,
                                         small_classes, conf, iou_threshold)
            if isinstance(tm, str) or len(tm) == 0:
                continue
            for mask, score, cls in zip(tm, ts, tc):
                small = P.resize_nearest(np.asarray(mask).astype(np.uint8), th, tw).astype(bool)
                if edge_filter and P.is_edge_mask(small, tile_size, overlap):
                    continue
                g = np.zeros((h, w), dtype=bool)
                ye, xe = min(y_off + th, h), min(x_off + tw, w)
                g[y_off:ye, x_off:xe] = small[: ye - y_off, : xe - x_off]
                tm_all.append(g)
                ts_all.append(score)
                tc_all.append(cls)
        if isinstance(fm, str):
            if tm_all:
                raise ValueError("operands could not be broadcast together")  # N4
            return [], [], []
        masks, scores, classes = list(fm) + tm_all, list(fs) + ts_all, list(fc) + tc_all
        if self.merge_mode == "soft_nms":
            if not masks:
                return [], [], []
            order, rs = P.soft_nms_masks(masks, scores, classes, self.soft_nms_sigma, self.soft_nms_thr)
            return [masks[i] for i in order], rs, [classes[i] for i in order]
        return P.deduplicate_masks_smart(masks, scores, classes, 0.4)

    # inference.py:1626-1736
    def small_classes(self, sample: Sequence[Tuple[str, np.ndarray]]):
        sizes: Dict[int, List[int]] = {}
        for key, img in sample[:5]:
            m, s, c = self.predict(0, (key, "full"), img)
            for mask, cls in zip(m[s >= 0.7], c[s >= 0.7]):
                sizes.setdefault(int(cls), []).append(int(np.sum(mask)))
        avg = {c: float(np.mean(v)) for c, v in sizes.items() if v}
        if not avg:
            return set()
        thr = np.percentile(list(avg.values()), 50)
        return {c for c, s in avg.items() if s <= thr}

    # inference.py:789-868
    def run_image(self, key: str, image: np.ndarray, small_classes, confidence_mode: str, spatial_cfg: dict,
                  ensemble_enabled=True, ensemble_small_only=True):
        tile_cfg = self.inf.get("tile_settings", {})
        ts, ov, up = tile_cfg.get("tile_size", 512), tile_cfg.get("overlap_ratio", 0.1), tile_cfg.get("upscale_factor", 2.0)
        edge = tile_cfg.get("edge_filter_enabled", True)
        masks, scores, classes = [], [], []
        for tc in range(self.K):
            is_small = tc in small_classes
            ccfg = self.class_settings.get(f"class_{tc}", {})
            if confidence_mode == "manual":
                conf = ccfg.get("confidence_threshold", 0.3 if is_small else 0.5)
            else:      # inference.py:809: the adaptive path reads the GLOBAL config, not the dataset override
                conf = P.get_confidence_threshold(image, tc, small_classes, self.global_config)
            iou_t = ccfg.get("iou_threshold", 0.5 if is_small else 0.7)
            use_ens = ensemble_enabled and (not ensemble_small_only or is_small)
            mids = list(range(len(self.models))) if (use_ens and len(self.models) > 1) else [0]
            m, s, c = self.tile_pipeline(mids, key, image, tc, small_classes, conf, ts, ov, up, iou_t, edge)
            masks.extend(m)
            scores.extend(s)
            classes.extend(c)
        masks, scores, classes = P.deduplicate_masks_smart(masks, scores, classes, 0.7)
        masks, scores, classes = P.apply_spatial_constraints(masks, scores, classes, spatial_cfg)
        self._cache.clear()
        return masks, scores, classes


def measurement_rows(name: str, masks, classes, thing_classes, um_pix=1.0, psum="0", image=None,
                     measure_contrast_distribution=False):
    """inference.py:1148-1230 -> list of 20-column rows (+ a trailing flag: ellipse fit numerically unstable)."""
    rows = []
    for iid, (mask, cls) in enumerate(zip(masks, classes), 1):
        cls = int(cls)
        for r in P.measure_mask(np.asarray(mask) > 0, um_pix, image, measure_contrast_distribution):
            rows.append([f"{name}_{iid}", cls, thing_classes[cls], r["major_axis_length"], r["minor_axis_length"], r["eccentricity"],
                         r["Length"], r["Width"], r["CircularED"], r["Aspect_Ratio"], r["Circularity"], r["Chords"],
                         r["Feret_diam"], r["Roundness"], r["Sphericity"], r["contrast_d10"], r["contrast_d50"], r["contrast_d90"],
                         psum, name, r["_ellipse_unstable"]])
    return rows


def overlay_without_text(image_bgr: np.ndarray, masks, classes, class_colors) -> np.ndarray:
    """The ``--visualize`` overlay of inference.py:1080-1101 without the putText calls, mask by mask in order:
    ``cv2.addWeighted(vis, 1.0, coloured_mask, 0.5, 0)`` (u8 saturating, cvRound = ties to even; outside the mask
    ``v * 1 + 0 * 0.5`` is v) and ``cv2.drawContours(vis, findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE), -1, colour, 1)``
    (the eight-direction segments between consecutive contour points, every lattice pixel on them)."""
    vis = image_bgr[:, :, :3].copy()
    for m, cls in zip(masks, classes):
        m = np.asarray(m) > 0
        color = np.asarray(class_colors[int(cls) % len(class_colors)], dtype=np.float64)
        vis[m] = np.clip(np.rint(vis[m].astype(np.float64) + 0.5 * color), 0, 255).astype(np.uint8)
        for c in P.find_external_contours(m):
            pts = np.asarray(c).reshape(-1, 2)
            n = len(pts)
            for k in range(n):
                (xa, ya), (xb, yb) = (int(v) for v in pts[k]), (int(v) for v in pts[(k + 1) % n])
                steps = max(abs(xb - xa), abs(yb - ya))
                for t in range(steps + 1):
                    x = xa + (xb - xa) * t // max(steps, 1) if steps else xa
                    y = ya + (yb - ya) * t // max(steps, 1) if steps else ya
                    vis[y, x] = color.astype(np.uint8)
    return vis
