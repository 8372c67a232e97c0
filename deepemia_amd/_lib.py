"""ctypes binding of ``libdeepemia_hip.so`` (the C ABI declared in ``include/deepemia_hip.h``).

The product path has **no CPU fallback**: if the shared library is missing, or a kernel
entry returns an error, this module raises.  PyTorch is used only for device memory and
streams; every pointer handed to the library is a raw device pointer.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
# DEEPEMIA_DEV_LIB=1: the dev build (`make -C deepemia_amd/csrc DEV=1`) with the kernels of the non-default precisions
LIB_PATH = _HERE / "csrc" / ("libdeepemia_hip_dev.so" if os.environ.get("DEEPEMIA_DEV_LIB", "0") == "1" else "libdeepemia_hip.so")

F32, BF16, F32X3, BF16X2, F16X2, P32 = 0, 1, 2, 3, 4, 5
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
RES_NONE, RES_SAME, RES_UP2 = 0, 1, 2
# stage codes of demia_mask_program (DEMIA_MOP_*)
MOP = {"fill": 1, "dilate": 2, "erode": 3, "drop_multi": 4, "flag_multi": 5, "gate": 6}


class HipExtensionMissing(RuntimeError):
    pass


class HipKernelError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p), ("w", C.c_void_p), ("scale", C.c_void_p), ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("out", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32), ("CoutPad", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("dtype", C.c_int32), ("out_dtype", C.c_int32), ("act", C.c_int32), ("res_mode", C.c_int32),
        ("out_ld", C.c_int32), ("tile_hint", C.c_int32),
        ("amax_in", C.c_void_p), ("amax_out", C.c_void_p),
    ]


class ConvP32Desc(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p), ("in_meta", C.c_void_p), ("w", C.c_void_p), ("scale", C.c_void_p), ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("res_meta", C.c_void_p), ("out", C.c_void_p), ("out_meta", C.c_void_p),
        ("wbound", C.c_float), ("bbound", C.c_float),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32), ("CoutPad", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("act", C.c_int32), ("res_mode", C.c_int32), ("out_f32", C.c_int32), ("out_ld", C.c_int32), ("tile_hint", C.c_int32),
        ("head_w", C.c_void_p), ("head_b", C.c_void_p), ("head_out", C.c_void_p),
        ("head_n", C.c_int32), ("head_ld", C.c_int32), ("head_act", C.c_int32),
        ("groups", C.c_int32), ("group_rows", C.c_int32), ("row0", C.c_int32), ("single", C.c_int32),
    ]


class RpnDesc(C.Structure):
    _fields_ = [
        ("head", C.c_void_p * 5), ("H", C.c_int32 * 5), ("W", C.c_int32 * 5), ("stride", C.c_int32 * 5),
        ("cell_anchors", C.c_void_p), ("head_ld", C.c_int32), ("N", C.c_int32),
        ("img_h", C.c_int32), ("img_w", C.c_int32), ("pre_topk", C.c_int32), ("post_topk", C.c_int32),
        ("nms_thresh", C.c_float), ("out_boxes", C.c_void_p), ("out_scores", C.c_void_p),
        ("out_count", C.c_void_p), ("workspace", C.c_void_p),
    ]


class RoiAlignDesc(C.Structure):
    _fields_ = [
        ("feat", C.c_void_p * 4), ("H", C.c_int32 * 4), ("W", C.c_int32 * 4),
        ("N", C.c_int32), ("R", C.c_int32), ("C", C.c_int32), ("P", C.c_int32), ("dtype", C.c_int32),
        ("boxes", C.c_void_p), ("count", C.c_void_p), ("out", C.c_void_p),
        ("meta", C.c_void_p * 4), ("out_meta", C.c_void_p), ("groups", C.c_int32), ("single", C.c_int32),
        ("order", C.c_void_p),
    ]


class DetsDesc(C.Structure):
    _fields_ = [
        ("logits", C.c_void_p), ("ld", C.c_int32), ("props", C.c_void_p), ("prop_count", C.c_void_p),
        ("N", C.c_int32), ("R", C.c_int32), ("K", C.c_int32), ("img_h", C.c_int32), ("img_w", C.c_int32),
        ("score_thresh", C.c_float), ("nms_thresh", C.c_float), ("topk", C.c_int32),
        ("det_boxes", C.c_void_p), ("det_scores", C.c_void_p), ("det_classes", C.c_void_p),
        ("det_count", C.c_void_p),
    ]


class PasteDesc(C.Structure):
    _fields_ = [
        ("mask_prob", C.c_void_p), ("ld", C.c_int32), ("det_boxes", C.c_void_p), ("det_classes", C.c_void_p),
        ("det_count", C.c_void_p), ("N", C.c_int32), ("D", C.c_int32), ("img_h", C.c_int32), ("img_w", C.c_int32),
        ("out_h", C.c_int32), ("out_w", C.c_int32), ("out_boxes", C.c_void_p), ("valid", C.c_void_p),
        ("packed", C.c_void_p), ("out_bbox", C.c_void_p), ("prev_bbox", C.c_void_p),
    ]


# every symbol include/deepemia_hip.h declares (tests check the library exports all of them)
EXPORTS = {
    "demia_abi_version": (C.c_int, []),
    "demia_conv_f16x2_kstep": (C.c_int, []),
    "demia_last_error": (C.c_char_p, []),
    "demia_build_arch": (C.c_char_p, []),
    "demia_build_flavor": (C.c_char_p, []),
    "demia_stream_create_cu_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "demia_stream_destroy": (C.c_int, [C.c_void_p]),
    "demia_conv2d_nhwc": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "demia_conv2d_p32": (C.c_int, [C.POINTER(ConvP32Desc), C.c_void_p]),
    "demia_conv2d_p32_single": (C.c_int, [C.POINTER(ConvP32Desc), C.c_void_p]),
    "demia_resize_h_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "demia_resize_v_norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int,
                                       C.c_void_p]),
    "demia_resize_linear_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "demia_stem_conv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p]),
    "demia_stem_conv_mfma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                        C.c_void_p]),
    "demia_stem_pool_mfma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "demia_maxpool3x3s2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "demia_maxpool3x3s2_p32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p]),
    "demia_subsample2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "demia_rpn_workspace_bytes": (C.c_int64, [C.c_int]),
    "demia_rpn_proposals": (C.c_int, [C.POINTER(RpnDesc), C.c_void_p]),
    "demia_roi_align": (C.c_int, [C.POINTER(RoiAlignDesc), C.c_void_p]),
    "demia_roi_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "demia_box_detections": (C.c_int, [C.POINTER(DetsDesc), C.c_void_p]),
    "demia_paste_masks": (C.c_int, [C.POINTER(PasteDesc), C.c_void_p]),
    "demia_unpack_masks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "demia_mask_area_bbox": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "demia_mask_program": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int64, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "demia_mask_program_wl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int64, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "demia_mask_overlap_prefix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "demia_mask_column_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "demia_mask_pair_intersections": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "demia_mask_pair_matrix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                          C.c_int, C.c_int, C.c_void_p]),
    "demia_host_greedy_keep": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                          C.c_void_p]),
    "demia_host_dedup_smart": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]),
    "demia_host_repr_rows": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64]),
    "demia_host_rle_text": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "demia_mask_place_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "demia_mask_crop_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "demia_mask_gather_regions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "demia_mask_gather_regions_pooled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                    C.c_int, C.c_void_p]),
    "demia_mask_crop_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "demia_mask_gray_histogram": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "demia_contour_work_ints": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "demia_contour_work_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "demia_contour_work_doubles": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "demia_mask_contours": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "demia_mask_contours_wl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "demia_contour_measure": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library or raise :class:`HipExtensionMissing` -- never fall back."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise HipExtensionMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C {LIB_PATH.parent}`); there is no CPU fallback for the hot path."
        )
    try:
        lib = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
    except OSError as e:  # missing libamdhip64 etc.
        raise HipExtensionMissing(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in EXPORTS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipExtensionMissing(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def is_dev_build() -> bool:
    return load().demia_build_flavor() == b"dev"


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().demia_last_error().decode(errors="replace")
        raise HipKernelError(f"{what} failed with status {status}: {msg}")


def ptr(t) -> int:
    """Raw device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else int(t.data_ptr())


def cu_masked_stream(device, spec: str):
    """A torch stream on ``device`` restricted by a CU mask (``demia_stream_create_cu_mask``).  ``spec`` = ``"mod:M:K"``: of every M
    consecutive compute-unit indices the ones >= K are left OUT (``mod:32:28``: 28 of every 32), over 256 CUs.  Returns
    (torch.cuda.ExternalStream, enabled CU count); the stream lives as long as the process."""
    import numpy as np
    import torch

    kind, m, k = spec.split(":")
    assert kind == "mod"
    m, k = int(m), int(k)
    bits = np.zeros(256, dtype=np.uint8)
    bits[(np.arange(256) % m) < k] = 1
    words = np.packbits(bits.reshape(8, 32), axis=1, bitorder="little").view(np.uint32).reshape(8).copy()
    handle = C.c_void_p()
    with torch.cuda.device(device):
        check(load().demia_stream_create_cu_mask(words.ctypes.data, 8, C.byref(handle)), "demia_stream_create_cu_mask")
    return torch.cuda.ExternalStream(handle.value, device=device), int(bits.sum())
