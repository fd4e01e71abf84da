"""Detection evaluation in the manner of csv_eval.evaluate (csv_eval.py:149-237).

The per-image part (IoU of detections with annotations and the greedy matching per class,
csv_eval.py:188-208) runs on the device; the dataset-level part (sorting all detections of a
class by score, cumulative sums, precision envelope, csv_eval.py:215-230 and :38-62) is a few
numpy lines on the host, as it works on one short list per class.
"""
from __future__ import annotations

import numpy as np
import torch

from . import functional as F


def _envelope_ap(recall, precision):
    """csv_eval.py:38-62."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


class DetectionEvaluator:
    """Accumulates matched detections over batches and reports AP per class.

    ``update`` takes what ``retinahead.postprocess(..., output='dicts')`` returns for a batch
    (already thresholded and in descending score order) and the batch's padded annotations.
    """

    def __init__(self, num_classes, iou_threshold=0.5, max_detections=100):
        self.num_classes = int(num_classes)
        self.iou_threshold = float(iou_threshold)
        self.max_detections = int(max_detections)
        self._scores, self._labels, self._tp = [], [], []
        self._num_annotations = np.zeros(self.num_classes, np.float64)

    def update(self, results, annotations, scales=None):
        """results: list of {'boxes' [K,4], 'labels' [K], 'scores' [K]} per image (device);
        annotations [B,M,5] device (rows with label -1 are padding), in original-image
        coordinates; scales: per-image resize factors the boxes are divided by
        (csv_eval.py:95), default 1."""
        bsz = len(results)
        dev = annotations.device
        d = max(1, min(self.max_detections, max(int(r["scores"].numel()) for r in results)))
        boxes = torch.zeros((bsz, d, 4), dtype=torch.float32, device=dev)
        labels = torch.zeros((bsz, d), dtype=torch.int64, device=dev)
        scores = torch.zeros((bsz, d), dtype=torch.float32, device=dev)
        counts = []
        for b, r in enumerate(results):
            k = min(int(r["scores"].numel()), d)
            counts.append(k)
            if k:
                bx = r["boxes"][:k].float()
                if scales is not None:
                    bx = bx / float(scales[b])
                boxes[b, :k], labels[b, :k], scores[b, :k] = bx, r["labels"][:k], r["scores"][:k]
        count_t = torch.tensor(counts, dtype=torch.int32).to(dev)
        tp = F.match_detections(boxes, labels, count_t, annotations, self.iou_threshold)
        tp_h, lab_h, sc_h = tp.cpu().numpy(), labels.cpu().numpy(), scores.cpu().numpy()
        ann_labels = annotations[:, :, 4].cpu().numpy()
        for b, k in enumerate(counts):
            self._scores.append(sc_h[b, :k].astype(np.float64))
            self._labels.append(lab_h[b, :k])
            self._tp.append(tp_h[b, :k].astype(np.float64))
        for c in range(self.num_classes):
            self._num_annotations[c] += float((ann_labels == c).sum())

    def average_precisions(self):
        """{class: (AP, number of annotations)} as csv_eval.evaluate returns it; a class without
        annotations gets AP 0 (csv_eval.py:211-213)."""
        scores = np.concatenate(self._scores + [np.zeros(0)])
        labels = np.concatenate(self._labels + [np.zeros(0, np.int64)])
        tp = np.concatenate(self._tp + [np.zeros(0)])
        out = {}
        for c in range(self.num_classes):
            num = self._num_annotations[c]
            if num == 0:
                out[c] = (0.0, 0.0)
                continue
            sel = labels == c
            order = np.argsort(-scores[sel], kind="stable")
            t = tp[sel][order]
            tps, fps = np.cumsum(t), np.cumsum(1.0 - t)
            recall = tps / num
            precision = tps / np.maximum(tps + fps, np.finfo(np.float64).eps)
            out[c] = (_envelope_ap(recall, precision), num)
        return out
