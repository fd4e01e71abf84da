"""Multi-threaded CPU restatement of the reference's loss and post-processing in
PyTorch eager ops + autograd.  TEST / BASELINE INFRASTRUCTURE ONLY.

Purpose: the ``cpu_baseline`` leg of ``bench.py`` times THIS (kind "port") on the
GPU box's host cores, because the reference itself cannot travel there and its
loss hard-codes ``.cuda()``.  It follows the reference's algorithm (same tensor
ops, same materialised N x M and N x C intermediates, autograd backward) for
models/retinanet/losses.py:29-141, utils.py:96-142 and model.py:320-340.  It is
somewhat lighter than the reference (one N x C target tensor instead of three, no
device moves): timed beside the reference's own loss in the build container
(tools/time_reference_cpu.py, 608 px, C=80, 8 threads) it runs between 1.0 and 2.3
times as fast, never slower, so the baseline it gives is an upper bound of the
reference's CPU rate.  Checked against the golden vectors in
tests/test_oracle_golden.py.  Nothing in the shipped package imports it.
"""
from __future__ import annotations

import torch


def pairwise_iou(anchor_boxes, gt_boxes):
    """[N,M] IoU, losses.py:6-23."""
    gt_area = (gt_boxes[:, 2] - gt_boxes[:, 0]) * (gt_boxes[:, 3] - gt_boxes[:, 1])
    ax1, ay1, ax2, ay2 = (anchor_boxes[:, k].unsqueeze(1) for k in range(4))
    w = (torch.min(ax2, gt_boxes[:, 2]) - torch.max(ax1, gt_boxes[:, 0])).clamp(min=0)
    h = (torch.min(ay2, gt_boxes[:, 3]) - torch.max(ay1, gt_boxes[:, 1])).clamp(min=0)
    union = ((ax2 - ax1) * (ay2 - ay1) + gt_area - w * h).clamp(min=1e-8)
    return (w * h) / union


def dense_head_loss(classifications, regressions, anchors, annotations, alpha=0.25):
    """{'classification_loss': [1], 'regression_loss': [1]} with autograd graph (gamma = 2)."""
    anchor = anchors[0]
    aw = anchor[:, 2] - anchor[:, 0]
    ah = anchor[:, 3] - anchor[:, 1]
    acx = anchor[:, 0] + 0.5 * aw
    acy = anchor[:, 1] + 0.5 * ah
    cls_terms, reg_terms = [], []
    zero = torch.zeros((), dtype=torch.float32)
    for img in range(classifications.shape[0]):
        gt = annotations[img]
        gt = gt[gt[:, 4] != -1]
        if gt.shape[0] == 0:
            cls_terms.append(zero)
            reg_terms.append(zero)
            continue
        p = classifications[img].clamp(1e-4, 1.0 - 1e-4)
        best, which = pairwise_iou(anchor, gt[:, :4]).max(dim=1)
        target = torch.full(p.shape, -1.0)
        target[best < 0.4] = 0
        pos = best >= 0.5
        npos = pos.sum()
        matched = gt[which]
        target[pos] = 0
        target[pos, matched[pos, 4].long()] = 1
        is_one = target == 1.0
        weight = torch.where(is_one, torch.full_like(p, alpha), torch.full_like(p, 1.0 - alpha))
        weight = weight * torch.where(is_one, 1.0 - p, p).pow(2.0)
        bce = -(target * p.log() + (1.0 - target) * (1.0 - p).log())
        loss = torch.where(target != -1.0, weight * bce, torch.zeros_like(p))
        cls_terms.append(loss.sum() / npos.float().clamp(min=1.0))
        if int(npos) > 0:
            m = matched[pos]
            gw = m[:, 2] - m[:, 0]
            gh = m[:, 3] - m[:, 1]
            gcx = m[:, 0] + 0.5 * gw
            gcy = m[:, 1] + 0.5 * gh
            gw = gw.clamp(min=1)
            gh = gh.clamp(min=1)
            t = torch.stack(((gcx - acx[pos]) / aw[pos], (gcy - acy[pos]) / ah[pos],
                             (gw / aw[pos]).log(), (gh / ah[pos]).log())).t()
            t = t / torch.tensor([[0.1, 0.1, 0.2, 0.2]])
            d = (t - regressions[img][pos]).abs()
            reg_terms.append(torch.where(d <= 1.0 / 9.0, 4.5 * d.pow(2), d - 0.5 / 9.0).mean())
        else:
            reg_terms.append(zero)
    return {"classification_loss": torch.stack(cls_terms).mean(dim=0, keepdim=True),
            "regression_loss": torch.stack(reg_terms).mean(dim=0, keepdim=True)}


def decode_clip(anchors, deltas, height, width):
    """utils.py:96-120 and :133-142."""
    w = anchors[:, :, 2] - anchors[:, :, 0]
    h = anchors[:, :, 3] - anchors[:, :, 1]
    cx = anchors[:, :, 0] + 0.5 * w
    cy = anchors[:, :, 1] + 0.5 * h
    pcx = cx + deltas[:, :, 0] * 0.1 * w
    pcy = cy + deltas[:, :, 1] * 0.1 * h
    pw = torch.exp(deltas[:, :, 2] * 0.2) * w
    ph = torch.exp(deltas[:, :, 3] * 0.2) * h
    x1 = (pcx - 0.5 * pw).clamp(min=0)
    y1 = (pcy - 0.5 * ph).clamp(min=0)
    x2 = (pcx + 0.5 * pw).clamp(max=width)
    y2 = (pcy + 0.5 * ph).clamp(max=height)
    return torch.stack([x1, y1, x2, y2], dim=2)


def greedy_nms(boxes, scores, thresh):
    """Vectorised-per-row greedy NMS (contract of torchvision.ops.nms)."""
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = torch.zeros(b.shape[0], dtype=torch.bool)
    keep = []
    for i in range(b.shape[0]):
        if dead[i]:
            continue
        keep.append(i)
        r = slice(i + 1, b.shape[0])
        w = (torch.min(b[i, 2], b[r, 2]) - torch.max(b[i, 0], b[r, 0])).clamp(min=0)
        h = (torch.min(b[i, 3], b[r, 3]) - torch.max(b[i, 1], b[r, 1])).clamp(min=0)
        inter = w * h
        dead[r] |= inter / (area[i] + area[r] - inter) > thresh
    return order[torch.tensor(keep, dtype=torch.long)]


def postprocess(classification, regression, anchors, height, width, score_thresh=0.05,
                iou_thresh=0.5):
    """model.py:320-340 applied to every image of the batch (class-agnostic NMS)."""
    boxes = decode_clip(anchors, regression, height, width)
    scores, labels = classification.max(dim=2)
    out = []
    for img in range(classification.shape[0]):
        sel = scores[img] > score_thresh
        if int(sel.sum()) == 0:
            out.append((torch.zeros(0), torch.zeros(0, dtype=torch.long), torch.zeros(0, 4)))
            continue
        keep = greedy_nms(boxes[img][sel], scores[img][sel], iou_thresh)
        out.append((scores[img][sel][keep], labels[img][sel][keep], boxes[img][sel][keep]))
    return out
