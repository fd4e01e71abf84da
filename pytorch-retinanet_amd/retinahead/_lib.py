"""ctypes binding of libretinahead.so (the C ABI declared in include/retinahead.h).

The library is the only compute path of this package: when it cannot be loaded
every operator raises, there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libretinahead.so")

RH_OK = 0
RH_NEGATIVE, RH_IGNORE, RH_POSITIVE = 0, 1, 2
RH_STATUS_LABEL_RANGE = 1
RH_STATUS_NONFINITE_BOX = 2
RH_STATUS_REDUCTION_GAVE_UP = 4
STAGE_ASSIGN, STAGE_STREAM, STAGE_FINALIZE, STAGE_ALL = 1, 2, 4, 7
POST_DECODE, POST_SORT, POST_NMS, POST_ALL = 1, 2, 4, 7

MAX_LEVELS = MAX_RATIOS = MAX_SCALES = 8
NMS_STATIC_MAX_BOXES = 4096      # RH_NMS_STATIC_MAX_BOXES


class AnchorCfg(C.Structure):
    _fields_ = [
        ("num_levels", C.c_int32),
        ("levels", C.c_int32 * MAX_LEVELS),
        ("strides", C.c_double * MAX_LEVELS),
        ("sizes", C.c_double * MAX_LEVELS),
        ("num_ratios", C.c_int32),
        ("ratios", C.c_double * MAX_RATIOS),
        ("num_scales", C.c_int32),
        ("scales", C.c_double * MAX_SCALES),
    ]


class FocalParams(C.Structure):
    _fields_ = [
        ("alpha", C.c_float),
        ("neg_thresh", C.c_float),
        ("pos_thresh", C.c_float),
        ("grad_cls", C.c_float),
        ("grad_reg", C.c_float),
        ("batch_divisor", C.c_int32),
        ("from_logits", C.c_int32),
        ("normalizer", C.c_int32),
        ("num_pos_total", C.c_void_p),
    ]


class AssignAhead(C.Structure):
    """rh_assign_ahead: the next step's assignment, carried by this step's stream launch."""
    _fields_ = [
        ("annotations", C.c_void_p),
        ("m", C.c_int32),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
        ("status", C.c_void_p),
        ("num_pos", C.c_void_p),
    ]


NORM_PER_IMAGE, NORM_GLOBAL = 0, 1
NORMALIZERS = {"per_image": NORM_PER_IMAGE, "global": NORM_GLOBAL}


class PostParams(C.Structure):
    _fields_ = [
        ("height", C.c_float),
        ("width", C.c_float),
        ("score_thresh", C.c_float),
        ("iou_thresh", C.c_float),
        ("per_class", C.c_int32),
        ("pre_nms_topk", C.c_int32),
        ("max_det", C.c_int32),
        ("cand_cap", C.c_int32),
        ("from_logits", C.c_int32),
    ]


_P = C.c_void_p
_I32, _I64, _F, _SZ = C.c_int32, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/retinahead.h one to one
SIGNATURES = {
    "rh_version": (C.c_int, []),
    "rh_last_error": (_SZ, [C.c_char_p, _SZ]),
    "rh_anchor_cfg_default": (None, [C.POINTER(AnchorCfg)]),
    "rh_anchor_count": (_I64, [C.POINTER(AnchorCfg), _I32, _I32]),
    "rh_anchors_f32": (C.c_int, [C.POINTER(AnchorCfg), _I32, _I32, _P, _I64, _P]),
    "rh_calc_iou_f32": (C.c_int, [_P, _I64, _P, _I64, _P, _P]),
    "rh_assign_workspace_bytes": (_SZ, [_I32, _I64, _I32]),
    "rh_assign": (C.c_int, [_P, _P, _I32, _I64, _I32, _I32, _F, _F, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "rh_focal_params_default": (None, [C.POINTER(FocalParams)]),
    "rh_focal_loss_workspace_bytes": (_SZ, [_I32, _I64, _I32, _I32]),
    "rh_focal_loss_levels_workspace_bytes": (_SZ, [_I32, _I32, _P, _P, _I32, _I32]),
    "rh_focal_loss_levels_fwd_bwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _P, _P, _I32, _I32, _I32,
                                               C.POINTER(FocalParams), _P, _P, _P, _P, _P, _P, _P, _SZ,
                                               _I32, _P]),
    "rh_focal_loss_fwd_bwd": (C.c_int, [_P, _P, _P, _P, _I32, _I64, _I32, _I32, C.POINTER(FocalParams),
                                        _P, _P, _P, _P, _P, _P, _P, _SZ, _I32, _P]),
    "rh_focal_loss_fwd_bwd_ahead": (C.c_int, [_P, _P, _P, _P, _I32, _I64, _I32, _I32, C.POINTER(FocalParams),
                                              _P, _P, _P, _P, _P, _P, _P, _SZ, _I32,
                                              C.POINTER(AssignAhead), _P]),
    "rh_apply_upstream_f32": (C.c_int, [_P, _I64, _P, _P, _I64, _P, _F, _P]),
    "rh_generic_focal_workspace_bytes": (_SZ, [_I64]),
    "rh_generic_focal_fwd_bwd": (C.c_int, [_P, _P, _I64, _F, _F, _I32, _I32, _P, _P, _P, _P, _SZ, _P]),
    "rh_bbox_transform_f32": (C.c_int, [_P, _P, _I32, _I64, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P]),
    "rh_clip_boxes_f32": (C.c_int, [_P, _I64, _F, _F, _P]),
    "rh_sigmoid_f32": (C.c_int, [_P, _P, _I64, _P]),
    "rh_collate_annotations_f32": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P]),
    "rh_sanitize_collate_annotations_f32": (C.c_int, [_P, _P, _P, _I32, _I32, C.c_float, _I32, _P, _P, _P]),
    "rh_match_detections": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, C.c_double, _P, _P]),
    "rh_nms_workspace_bytes": (_SZ, [_I64]),
    "rh_nms": (C.c_int, [_P, _P, _P, _I64, _F, _P, _P, _P, _SZ, _P]),
    "rh_nms_static": (C.c_int, [_P, _P, _P, _I64, _F, _P, _P, _P]),
    "rh_postprocess_workspace_bytes": (_SZ, [_I32, _I64, _I32, C.POINTER(PostParams)]),
    "rh_postprocess": (C.c_int, [_P, _P, _P, _I32, _I64, _I32, C.POINTER(PostParams), _P, _P, _P, _P,
                                 _P, _P, _P, _SZ, _I32, _P]),
    "rh_postprocess_levels_workspace_bytes": (_SZ, [_I32, _I32, _P, _P, _I32, _I32,
                                                    C.POINTER(PostParams)]),
    "rh_postprocess_levels": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P, _I32, _I32,
                                        C.POINTER(PostParams), _P, _P, _P, _P, _P, _P, _P, _SZ, _I32,
                                        _P]),
}

_lib = None


class RetinaHeadError(RuntimeError):
    pass


def load():
    """Returns the loaded library; raises if the shared object is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RetinaHeadError(
            f"{LIB_PATH} not found: build it with `make -C pytorch-retinanet_amd/csrc` "
            "(hipcc, gfx950).  This package has no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    buf = C.create_string_buffer(512)
    load().rh_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(code, what):
    if code != RH_OK:
        raise RetinaHeadError(f"{what} failed ({code}): {last_error()}")
