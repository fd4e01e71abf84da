"""retinahead: RetinaNet's per-anchor dense-head path as gfx950 (MI355X) HIP kernels.

Python keeps the reference's call signatures (mesnico/pytorch-retinanet); all
computation happens in libretinahead.so through the C ABI of include/retinahead.h.
"""
from ._lib import LIB_PATH, RetinaHeadError, load as load_library
from .modules import (Anchors, BBoxTransform, ClipBoxes, FocalLoss, GenericFocalLoss, batched_nms,
                      calc_iou, nms, postprocess)
from .functional import collate_annotations, sigmoid
from .evaluate import DetectionEvaluator
from .pipeline import PostprocessPipeline
from . import functional
from . import dist

__all__ = ["Anchors", "BBoxTransform", "ClipBoxes", "FocalLoss", "GenericFocalLoss", "batched_nms",
           "calc_iou", "nms", "postprocess", "sigmoid", "collate_annotations", "DetectionEvaluator", "PostprocessPipeline", "functional", "dist", "load_library", "LIB_PATH",
           "RetinaHeadError"]
