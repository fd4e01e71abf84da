"""CPU oracle for the RetinaNet dense-head path.  TEST INFRASTRUCTURE ONLY.

This file restates, in numpy float32/float64 arithmetic, what the reference
computes for anchors, IoU assignment, the focal / smooth-L1 losses with their
gradients, box decode / clip and the inference post-processing.  It is the
checker for the HIP kernels: only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing in the shipped
package calls it.

Pinning: every function except ``nms`` is checked against outputs of the
reference's own files (``tests/golden/*.npz`` written by
``tools/gen_golden.py``, which imports the reference by path in the build
container).  ``nms`` restates the documented contract of the third-party
``torchvision.ops.nms`` the reference calls at models/retinanet/model.py:336;
torchvision is neither vendored nor installed and its version is not pinned by
the reference, so NMS is **parity unpinned** and is checked only against a
brute-force greedy implementation and hand-computed cases.

All float32 arithmetic keeps the reference's operation order wherever a result
is compared bit-for-bit (anchors, IoU, assignment, decode, NMS keep lists).
Loss sums are accumulated in float64 and compared with a 1e-5 relative
tolerance.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

# Assignment states, shared with the HIP library (include/retinahead.h).
STATE_NEGATIVE = 0
STATE_IGNORE = 1
STATE_POSITIVE = 2


# --------------------------------------------------------------------------
# Anchors  (reference: models/retinanet/anchors.py:22-41, 48-79, 115-135)
# --------------------------------------------------------------------------
DEFAULT_LEVELS = (3, 4, 5, 6, 7)
DEFAULT_RATIOS = (0.5, 1.0, 2.0)
DEFAULT_SCALES = (2.0 ** 0, 2.0 ** (1.0 / 3.0), 2.0 ** (2.0 / 3.0))


def base_anchors(base_size, ratios=DEFAULT_RATIOS, scales=DEFAULT_SCALES):
    """The A = len(ratios)*len(scales) boxes centred on the origin, float64.

    anchors.py:60-79: ratio-major / scale-minor order; w = sqrt(area/ratio),
    h = w*ratio, box = (-w/2, -h/2, w - w/2, h - h/2) (the reference subtracts
    half the extent from a (0, 0, w, h) row, so x2 is ``w - 0.5*w``).
    """
    out = np.zeros((len(ratios) * len(scales), 4), dtype=np.float64)
    a = 0
    for r in ratios:
        for s in scales:
            side = float(base_size) * float(s)
            area = side * side
            w = np.sqrt(area / float(r))
            h = w * float(r)
            out[a] = (0.0 - w * 0.5, 0.0 - h * 0.5, w - w * 0.5, h - h * 0.5)
            a += 1
    return out


def level_shapes(height, width, levels=DEFAULT_LEVELS):
    """anchors.py:24-26: ceil(H / 2^l), ceil(W / 2^l) per pyramid level."""
    return [((height + 2 ** l - 1) // 2 ** l, (width + 2 ** l - 1) // 2 ** l) for l in levels]


def num_anchors(height, width, levels=DEFAULT_LEVELS, per_cell=9):
    return sum(h * w for h, w in level_shapes(height, width, levels)) * per_cell


def anchors_for_image(height, width, levels=DEFAULT_LEVELS, strides=None, sizes=None,
                      ratios=DEFAULT_RATIOS, scales=DEFAULT_SCALES):
    """[1, N, 4] float32 anchors, anchors.py:22-41.

    Order: level -> cell row (y) -> cell column (x) -> ratio -> scale.  The
    sum ``base + (i + 0.5) * stride`` is formed in float64 and narrowed once.
    """
    strides = [2 ** l for l in levels] if strides is None else list(strides)
    sizes = [2 ** (l + 2) for l in levels] if sizes is None else list(sizes)
    chunks = []
    for (fh, fw), stride, size in zip(level_shapes(height, width, levels), strides, sizes):
        base = base_anchors(size, ratios, scales)                       # [A, 4]
        sx = (np.arange(fw, dtype=np.float64) + 0.5) * stride
        sy = (np.arange(fh, dtype=np.float64) + 0.5) * stride
        shift = np.zeros((fh, fw, 1, 4), dtype=np.float64)
        shift[..., 0, 0] = sx[None, :]
        shift[..., 0, 2] = sx[None, :]
        shift[..., 0, 1] = sy[:, None]
        shift[..., 0, 3] = sy[:, None]
        chunks.append((base[None, None, :, :] + shift).reshape(-1, 4))
    allb = np.concatenate(chunks, axis=0) if chunks else np.zeros((0, 4))
    return allb.astype(F32)[None, :, :]


# --------------------------------------------------------------------------
# IoU + assignment  (reference: models/retinanet/losses.py:6-23, 48-79)
# --------------------------------------------------------------------------
def calc_iou(a, b):
    """[N, M] float32 IoU, losses.py:6-23 (no +1, union clamped at 1e-8)."""
    a = np.asarray(a, dtype=F32)
    b = np.asarray(b, dtype=F32)
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iw = np.minimum(a[:, 2:3], b[None, :, 2]) - np.maximum(a[:, 0:1], b[None, :, 0])
    ih = np.minimum(a[:, 3:4], b[None, :, 3]) - np.maximum(a[:, 1:2], b[None, :, 1])
    iw = np.maximum(iw, F32(0))
    ih = np.maximum(ih, F32(0))
    area_a = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None]
    ua = (area_a + area_b[None, :]) - iw * ih
    ua = np.maximum(ua, F32(1e-8))
    return ((iw * ih) / ua).astype(F32)


def valid_gt(annotation):
    """losses.py:48-49: rows whose label is not -1, original order kept."""
    annotation = np.asarray(annotation, dtype=F32)
    return annotation[annotation[:, 4] != F32(-1)]


def assign(anchors, annotation, neg_thresh=0.4, pos_thresh=0.5):
    """Per-anchor assignment for ONE image.

    Returns dict(iou_max f32[N], argmax i32[N] (index into the valid-GT list,
    first index on ties), state u8[N], label i32[N] (-1 unless positive),
    num_pos int, num_gt int).  losses.py:59-61, 70-79.  An image with no valid
    GT is skipped by the reference (losses.py:51-55): every anchor is reported
    as IGNORE so that it contributes neither loss nor gradient.
    """
    anchors = np.asarray(anchors, dtype=F32).reshape(-1, 4)
    gt = valid_gt(annotation)
    n = anchors.shape[0]
    if gt.shape[0] == 0:
        return dict(iou_max=np.zeros(n, F32), argmax=np.zeros(n, np.int32),
                    state=np.full(n, STATE_IGNORE, np.uint8), label=np.full(n, -1, np.int32),
                    num_pos=0, num_gt=0)
    iou = calc_iou(anchors, gt[:, :4])
    argmax = np.argmax(iou, axis=1).astype(np.int32)          # first index on ties
    iou_max = iou[np.arange(n), argmax]
    state = np.full(n, STATE_IGNORE, np.uint8)
    state[iou_max < F32(neg_thresh)] = STATE_NEGATIVE
    pos = iou_max >= F32(pos_thresh)
    state[pos] = STATE_POSITIVE
    label = np.full(n, -1, np.int32)
    label[pos] = gt[argmax[pos], 4].astype(np.int64).astype(np.int32)   # .long(): truncation
    return dict(iou_max=iou_max.astype(F32), argmax=argmax, state=state, label=label,
                num_pos=int(pos.sum()), num_gt=int(gt.shape[0]))


# --------------------------------------------------------------------------
# Focal + smooth-L1 loss, forward and closed-form backward
# (reference: models/retinanet/losses.py:29-141; backward = what autograd
#  derives from those lines)
# --------------------------------------------------------------------------
P_LO = F32(1e-4)
P_HI = F32(1.0 - 1e-4)


def _focal_image(p, state, label, alpha):
    """Element losses and d(loss)/dp for one image, before normalisation."""
    n, c = p.shape
    pc = np.clip(p, P_LO, P_HI).astype(F32)                    # losses.py:57
    one = F32(1)
    is_one = np.zeros((n, c), dtype=bool)
    pos_rows = np.nonzero(state == STATE_POSITIVE)[0]
    is_one[pos_rows, label[pos_rows]] = True                   # losses.py:78-79
    a_pos = F32(alpha)
    a_neg = one - F32(alpha)
    # losses.py:81-92
    q = (one - pc).astype(F32)
    log_p = np.log(pc).astype(F32)
    log_q = np.log(q).astype(F32)
    loss_one = (a_pos * (q * q)) * (-log_p)
    loss_zero = (a_neg * (pc * pc)) * (-log_q)
    loss = np.where(is_one, loss_one, loss_zero).astype(F32)
    g_one = a_pos * (F32(2) * q * log_p - (q * q) / pc)
    g_zero = a_neg * (F32(-2) * pc * log_q + (pc * pc) / q)
    grad = np.where(is_one, g_one, g_zero).astype(F32)
    ignore = (state == STATE_IGNORE)[:, None]
    loss = np.where(ignore, F32(0), loss)
    inside = (p >= P_LO) & (p <= P_HI)                         # clamp passes gradient on [lo, hi]
    grad = np.where(ignore | ~inside, F32(0), grad)
    return loss, grad


def _box_targets(anchors, gt_boxes):
    """losses.py:36-41, 106-123 for already-gathered positive rows."""
    aw = anchors[:, 2] - anchors[:, 0]
    ah = anchors[:, 3] - anchors[:, 1]
    acx = anchors[:, 0] + F32(0.5) * aw
    acy = anchors[:, 1] + F32(0.5) * ah
    gw = gt_boxes[:, 2] - gt_boxes[:, 0]
    gh = gt_boxes[:, 3] - gt_boxes[:, 1]
    gcx = gt_boxes[:, 0] + F32(0.5) * gw
    gcy = gt_boxes[:, 1] + F32(0.5) * gh
    gw = np.maximum(gw, F32(1))
    gh = np.maximum(gh, F32(1))
    t = np.stack([(gcx - acx) / aw, (gcy - acy) / ah,
                  np.log(gw / aw).astype(F32), np.log(gh / ah).astype(F32)], axis=1).astype(F32)
    return (t / np.array([0.1, 0.1, 0.2, 0.2], dtype=F32)[None, :]).astype(F32)


def sigmoid_f32(x):
    """float32 sigmoid as torch.sigmoid evaluates it, 1 / (1 + exp(-x)): the head's output
    activation, models/retinanet/model.py:177, 193."""
    x = np.asarray(x, dtype=F32)
    with np.errstate(over="ignore"):
        return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def focal_loss_fwd_bwd(classifications, regressions, anchors, annotations,
                       alpha=0.25, grad_cls=1.0, grad_reg=1.0, batch_divisor=None,
                       from_logits=False, normalizer="per_image", num_pos_total=None):
    """Losses and gradients for a batch (gamma is 2 as in losses.py:31).

    Returns dict with
      classification_loss, regression_loss : float32 [1]   (losses.py:139-141)
      cls_per_image, reg_per_image         : float32 [B]
      num_pos                              : int32 [B]
      d_classifications [B,N,C], d_regressions [B,N,4] : gradients of
        grad_cls*classification_loss + grad_reg*regression_loss.
    ``batch_divisor`` (default B) is the number of images the mean runs over;
    data-parallel shards pass the global batch size.
    ``from_logits``: classifications are the head's logits; the sigmoid of model.py:177,193
    is applied first and d_classifications is the gradient with respect to the logits
    (chain rule through the sigmoid, dp/dx = p (1 - p), as autograd applies it).
    ``normalizer='global'`` (no counterpart in the reference, whose normalisers are per image,
    losses.py:94, :134): one positive count P for the whole batch -- ``num_pos_total`` if given
    (a data-parallel shard passes the count over all ranks), else this batch's positives --
    replaces the per-image counts and the batch mean: classification_loss = sum of the images'
    focal sums / max(P, 1), regression_loss = sum of their smooth-L1 sums / (4 P); the
    per-image outputs are the images' shares; ``batch_divisor`` is not used.
    """
    if normalizer not in ("per_image", "global"):
        raise ValueError("normalizer must be 'per_image' or 'global'")
    batch_wide = normalizer == "global"
    logits_in = bool(from_logits)
    if logits_in:
        classifications = sigmoid_f32(classifications)
    cls = np.asarray(classifications, dtype=F32)
    reg = np.asarray(regressions, dtype=F32)
    anc = np.asarray(anchors, dtype=F32).reshape(-1, 4)
    ann = np.asarray(annotations, dtype=F32)
    bsz, n, c = cls.shape
    div = 1.0 if batch_wide else float(bsz if batch_divisor is None else batch_divisor)
    assigned = [assign(anc, ann[j]) for j in range(bsz)]
    if batch_wide:
        total = (sum(int(a["num_pos"]) for a in assigned) if num_pos_total is None
                 else int(num_pos_total))
    d_cls = np.zeros_like(cls)
    d_reg = np.zeros_like(reg)
    cls_img = np.zeros(bsz, F32)
    reg_img = np.zeros(bsz, F32)
    npos = np.zeros(bsz, np.int32)
    for j in range(bsz):
        asg = assigned[j]
        npos[j] = asg["num_pos"]
        if asg["num_gt"] == 0:
            continue                                            # losses.py:51-55
        loss, grad = _focal_image(cls[j], asg["state"], asg["label"], alpha)
        norm = F32(max(total if batch_wide else asg["num_pos"], 1))
        cls_img[j] = F32(loss.sum(dtype=np.float64)) / norm     # losses.py:94
        d_cls[j] = grad * (F32(grad_cls) / (norm * F32(div)))
        if logits_in:
            d_cls[j] = d_cls[j] * ((F32(1) - cls[j]) * cls[j])
        pos = asg["state"] == STATE_POSITIVE
        if asg["num_pos"] > 0:                                  # losses.py:98-134
            gt = valid_gt(ann[j])
            tgt = _box_targets(anc[pos], gt[asg["argmax"][pos], :4])
            delta = reg[j][pos] - tgt
            d = np.abs(delta).astype(F32)
            small = d <= F32(1.0 / 9.0)
            elem = np.where(small, F32(0.5 * 9.0) * (d * d), d - F32(0.5 / 9.0)).astype(F32)
            count = F32(4 * (total if batch_wide else asg["num_pos"]))
            reg_img[j] = F32(elem.sum(dtype=np.float64)) / count
            g = np.where(small, F32(9.0) * d, F32(1)) * np.sign(delta)
            d_reg[j][pos] = (g * (F32(grad_reg) / (count * F32(div)))).astype(F32)
    return dict(
        classification_loss=np.array([cls_img.sum(dtype=np.float64) / div], dtype=F32),
        regression_loss=np.array([reg_img.sum(dtype=np.float64) / div], dtype=F32),
        cls_per_image=cls_img, reg_per_image=reg_img, num_pos=npos,
        d_classifications=d_cls, d_regressions=d_reg)


# --------------------------------------------------------------------------
# Generic elementwise focal BCE  (reference: models/focal_loss.py:14-25)
# --------------------------------------------------------------------------
def generic_focal_fwd_bwd(inputs, targets, alpha=0.25, gamma=2.0, logits=True, reduce=True):
    """Returns (loss, d loss / d inputs).  float64 internally, float32 out.

    BCE = binary cross entropy (on logits or probabilities; torch clamps the
    log terms at -100 for probabilities), pt = exp(-BCE),
    F = alpha * (1-pt)^gamma * BCE, mean over all elements when ``reduce``.
    """
    x = np.asarray(inputs, dtype=np.float64)
    t = np.asarray(targets, dtype=np.float64)
    if logits:
        bce = np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x)))
        dbce = 1.0 / (1.0 + np.exp(-x)) - t
    else:
        with np.errstate(divide="ignore", invalid="ignore"):
            lp = np.log(x)
            lq = np.log(1.0 - x)
        lpc = np.maximum(lp, -100.0)
        lqc = np.maximum(lq, -100.0)
        bce = -(t * lpc + (1.0 - t) * lqc)
        # torch's binary_cross_entropy backward: (x - t) / max((1 - x) * x, 1e-12)
        dbce = (x - t) / np.maximum((1.0 - x) * x, 1e-12)
    pt = np.exp(-bce)
    om = 1.0 - pt
    f = alpha * om ** gamma * bce
    df_dbce = alpha * (gamma * om ** (gamma - 1.0) * pt * bce + om ** gamma)
    grad = df_dbce * dbce
    if reduce:
        return np.float32(f.mean()), (grad / x.size).astype(F32)
    return f.astype(F32), grad.astype(F32)


# --------------------------------------------------------------------------
# Decode / clip  (reference: models/retinanet/utils.py:96-120, 133-142)
# --------------------------------------------------------------------------
def bbox_transform(boxes, deltas, mean=(0.0, 0.0, 0.0, 0.0), std=(0.1, 0.1, 0.2, 0.2)):
    boxes = np.asarray(boxes, dtype=F32)
    deltas = np.asarray(deltas, dtype=F32)
    mean = np.asarray(mean, dtype=F32)
    std = np.asarray(std, dtype=F32)
    w = boxes[:, :, 2] - boxes[:, :, 0]
    h = boxes[:, :, 3] - boxes[:, :, 1]
    cx = boxes[:, :, 0] + F32(0.5) * w
    cy = boxes[:, :, 1] + F32(0.5) * h
    dx = deltas[:, :, 0] * std[0] + mean[0]
    dy = deltas[:, :, 1] * std[1] + mean[1]
    dw = deltas[:, :, 2] * std[2] + mean[2]
    dh = deltas[:, :, 3] * std[3] + mean[3]
    pcx = cx + dx * w
    pcy = cy + dy * h
    pw = np.exp(dw).astype(F32) * w
    ph = np.exp(dh).astype(F32) * h
    return np.stack([pcx - F32(0.5) * pw, pcy - F32(0.5) * ph,
                     pcx + F32(0.5) * pw, pcy + F32(0.5) * ph], axis=2).astype(F32)


def clip_boxes(boxes, height, width):
    out = np.array(boxes, dtype=F32, copy=True)
    out[:, :, 0] = np.maximum(out[:, :, 0], F32(0))
    out[:, :, 1] = np.maximum(out[:, :, 1], F32(0))
    out[:, :, 2] = np.minimum(out[:, :, 2], F32(width))
    out[:, :, 3] = np.minimum(out[:, :, 3], F32(height))
    return out


# --------------------------------------------------------------------------
# NMS  (contract of torchvision.ops.nms, called at model.py:336; PARITY UNPINNED)
# --------------------------------------------------------------------------
def box_iou_pair(a, b):
    """float32 IoU of two boxes with torchvision's arithmetic order."""
    xx1 = max(a[0], b[0])
    yy1 = max(a[1], b[1])
    xx2 = min(a[2], b[2])
    yy2 = min(a[3], b[3])
    w = max(F32(0), F32(xx2 - xx1))
    h = max(F32(0), F32(yy2 - yy1))
    inter = F32(w * h)
    area_a = F32(F32(a[2] - a[0]) * F32(a[3] - a[1]))
    area_b = F32(F32(b[2] - b[0]) * F32(b[3] - b[1]))
    with np.errstate(divide="ignore", invalid="ignore"):
        return F32(inter / F32(F32(area_a + area_b) - inter))


def sort_order(scores, tiebreak=None):
    """Indices by descending score; ties by ascending ``tiebreak`` (default: position)."""
    scores = np.asarray(scores, dtype=F32).reshape(-1)
    tb = np.arange(scores.shape[0]) if tiebreak is None else np.asarray(tiebreak)
    return np.lexsort((tb, -scores.astype(np.float64)))


def nms(boxes, scores, iou_threshold, labels=None):
    """Greedy NMS.  Returns int64 kept indices in descending-score order.

    A box is dropped when its IoU with an already kept box is strictly greater
    than ``iou_threshold``; with ``labels`` given only same-label boxes
    suppress each other (per-class NMS).  Score ties keep the lower index first.
    """
    boxes = np.asarray(boxes, dtype=F32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=F32).reshape(-1)
    k = boxes.shape[0]
    if k == 0:
        return np.zeros(0, np.int64)
    order = sort_order(scores)
    b = boxes[order]
    lab = None if labels is None else np.asarray(labels).reshape(-1)[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = np.zeros(k, dtype=bool)
    keep = []
    thr = F32(iou_threshold)
    for i in range(k):
        if dead[i]:
            continue
        keep.append(order[i])
        if i + 1 == k:
            break
        r = slice(i + 1, k)
        w = np.maximum(F32(0), np.minimum(b[i, 2], b[r, 2]) - np.maximum(b[i, 0], b[r, 0]))
        h = np.maximum(F32(0), np.minimum(b[i, 3], b[r, 3]) - np.maximum(b[i, 1], b[r, 1]))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / ((area[i] + area[r]) - inter)
        hit = iou > thr
        if lab is not None:
            hit &= lab[r] == lab[i]
        dead[r] |= hit
    return np.asarray(keep, dtype=np.int64)


def nms_bruteforce(boxes, scores, iou_threshold, labels=None):
    """O(K^2) scalar restatement used to cross-check ``nms`` in the tests."""
    boxes = np.asarray(boxes, dtype=F32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=F32).reshape(-1)
    order = sort_order(scores)
    keep = []
    for i in order:
        ok = True
        for kpt in keep:
            if labels is not None and labels[kpt] != labels[i]:
                continue
            if box_iou_pair(boxes[kpt], boxes[i]) > F32(iou_threshold):
                ok = False
                break
        if ok:
            keep.append(int(i))
    return np.asarray(keep, dtype=np.int64)


# --------------------------------------------------------------------------
# Inference post-processing  (reference: models/retinanet/model.py:319-340)
# --------------------------------------------------------------------------
def postprocess_image(classification, regression, anchors, height, width,
                      score_thresh=0.05, iou_thresh=0.5, mode="agnostic",
                      pre_nms_topk=None, max_det=None, boxes=None, from_logits=False):
    """One image: (scores f32[K'], labels i64[K'], boxes f32[K',4]).

    model.py:320-340: decode, clip, per-anchor max over classes, keep
    ``score > score_thresh``, greedy NMS at ``iou_thresh`` on (box, max score),
    outputs in descending score order with the arg-max class as label.
    ``mode='per_class'`` restricts suppression to equal labels;
    ``pre_nms_topk`` keeps the best-scoring candidates (ties: lower anchor
    index) before NMS; ``max_det`` truncates the kept list.  ``boxes`` [N,4], when
    given, replaces the decode + clip step (lets a test feed the device's own decoded
    boxes, whose exp() may differ from numpy's in the last place, so that keep lists
    can be compared exactly).  ``from_logits``: ``classification`` holds logits; the head's
    sigmoid (model.py:177,193) is applied first.
    """
    p = np.asarray(classification, dtype=F32)
    if from_logits:
        p = sigmoid_f32(p)
    if boxes is None:
        boxes = clip_boxes(bbox_transform(np.asarray(anchors, F32).reshape(1, -1, 4),
                                          np.asarray(regression, F32)[None]), height, width)[0]
    else:
        boxes = np.asarray(boxes, dtype=F32).reshape(-1, 4)
    score = p.max(axis=1)
    label = p.argmax(axis=1).astype(np.int64)                  # first index on ties
    cand = np.nonzero(score > F32(score_thresh))[0]
    if cand.size == 0:
        return np.zeros(0, F32), np.zeros(0, np.int64), np.zeros((0, 4), F32)
    if pre_nms_topk is not None and cand.size > pre_nms_topk:
        cand = cand[sort_order(score[cand], cand)[:pre_nms_topk]]
        cand.sort()
    keep = nms(boxes[cand], score[cand], iou_thresh,
               labels=label[cand] if mode == "per_class" else None)
    if max_det is not None:
        keep = keep[:max_det]
    sel = cand[keep]
    return score[sel], label[sel], boxes[sel]


# --------------------------------------------------------------------------
# Annotation collation  (reference: models/retinanet/model.py:27, 43-55)
# --------------------------------------------------------------------------
def sanitize_annotations(boxes, labels, min_size=1.0, xywh=False):
    """The annotation filter of the reference's dataset loaders, for float32 rows: drop a row
    whose width or height is below ``min_size`` (csv_dataset.py:129-136 tests x2 - x1 / y2 - y1,
    coco_dataset.py:97-99 the stored w / h; ``nan < 1`` is false, so a NaN extent stays), turn
    [x, y, w, h] into corners by x + w, y + h (coco_dataset.py:106-108), keep the order.
    Unpinned: the loaders import pycocotools / skimage / torchvision, absent here, so they cannot
    run; the rule is restated from the lines cited."""
    b = np.asarray(boxes, F32).reshape(-1, 4).copy()
    l = np.asarray(labels, F32).reshape(-1)
    if xywh:
        w, h = b[:, 2].copy(), b[:, 3].copy()
        b[:, 2] = b[:, 0] + w
        b[:, 3] = b[:, 1] + h
    else:
        w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    keep = ~((w < F32(min_size)) | (h < F32(min_size)))
    return b[keep], l[keep]


def collate_annotations(targets, min_size=None, xywh=False):
    """[B, M, 5] float32: image b's rows (x1, y1, x2, y2, label), padded with -1 to the largest
    count of the input lists (one row when every image is empty, model.py:55).  ``targets``:
    list of (boxes [m,4], labels [m]) pairs.  The reference function itself cannot run (it returns
    an undefined name, model.py:66), so this restatement is unpinned; it is a plain copy loop.
    ``min_size`` / ``xywh``: :func:`sanitize_annotations` applied to every image first."""
    m = max([len(b) for b, _ in targets] + [1])
    out = -np.ones((len(targets), m, 5), F32)
    for i, (b, l) in enumerate(targets):
        if min_size is not None or xywh:
            b, l = sanitize_annotations(b, l, -np.inf if min_size is None else min_size, xywh)
        k = len(b)
        if k:
            out[i, :k, :4] = np.asarray(b, F32).reshape(k, 4)
            out[i, :k, 4] = np.asarray(l, F32).reshape(k)
    return out


# --------------------------------------------------------------------------
# Detection evaluation  (reference: csv_eval.py:11-35, 38-62, 97-118, 176-230)
# --------------------------------------------------------------------------
def compute_overlap(a, b):
    """[N,K] float64 IoU, csv_eval.py:11-35 (union clamped at the float64 epsilon)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iw = np.minimum(a[:, 2:3], b[:, 2]) - np.maximum(a[:, 0:1], b[:, 0])
    ih = np.minimum(a[:, 3:4], b[:, 3]) - np.maximum(a[:, 1:2], b[:, 1])
    iw = np.maximum(iw, 0)
    ih = np.maximum(ih, 0)
    ua = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None] + area - iw * ih
    ua = np.maximum(ua, np.finfo(float).eps)
    return iw * ih / ua


def match_detections(det_boxes, det_labels, annotations, iou_threshold=0.5):
    """One image.  ``det_boxes`` [D,4] / ``det_labels`` [D] in descending score order,
    ``annotations`` [M,5] (label -1 rows are padding).  Returns uint8 [D]: 1 where the detection
    is a true positive, i.e. its best-overlapping annotation of the same class (first on ties)
    reaches the threshold and has not been claimed by an earlier detection (csv_eval.py:188-208).
    All arithmetic in float64 like the reference's numpy code."""
    det_boxes = np.asarray(det_boxes, np.float64).reshape(-1, 4)
    ann = np.asarray(annotations, np.float64).reshape(-1, 5)
    ann = ann[ann[:, 4] != -1]
    tp = np.zeros(det_boxes.shape[0], np.uint8)
    taken = np.zeros(ann.shape[0], bool)
    for d in range(det_boxes.shape[0]):
        same = np.nonzero(ann[:, 4] == det_labels[d])[0]
        if same.size == 0:
            continue
        ov = compute_overlap(det_boxes[d:d + 1], ann[same, :4])[0]
        k = int(np.argmax(ov))
        if ov[k] >= iou_threshold and not taken[same[k]]:
            tp[d] = 1
            taken[same[k]] = True
    return tp


def average_precision(recall, precision):
    """csv_eval.py:38-62: area under the precision envelope."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = max(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def average_precisions(scores, labels, tp, annotations, num_classes):
    """Dataset-level AP per class from per-image lists (csv_eval.py:176-230).
    ``scores`` / ``labels`` / ``tp``: lists with one array per image (detections in the order
    they were matched); ``annotations``: list of [M,5].  Returns (ap [C], num_annotations [C])."""
    ap = np.zeros(num_classes, np.float64)
    num = np.zeros(num_classes, np.float64)
    for c in range(num_classes):
        s = np.concatenate([np.asarray(sc, np.float64)[np.asarray(lb) == c] for sc, lb in zip(scores, labels)] + [np.zeros(0)])
        t = np.concatenate([np.asarray(x, np.float64)[np.asarray(lb) == c] for x, lb in zip(tp, labels)] + [np.zeros(0)])
        num[c] = sum(int((np.asarray(a)[:, 4] == c).sum()) for a in annotations)
        if num[c] == 0:
            continue                                  # csv_eval.py:211-213: AP 0
        order = np.argsort(-s, kind="stable")
        tps = np.cumsum(t[order])
        fps = np.cumsum(1.0 - t[order])
        recall = tps / num[c]
        precision = tps / np.maximum(tps + fps, np.finfo(np.float64).eps)
        ap[c] = average_precision(recall, precision)
    return ap, num


def evaluate_detections(det_boxes, det_scores, det_labels, det_counts, annotations, scales,
                        num_classes, iou_threshold=0.5, score_threshold=0.05, max_detections=100):
    """The whole of csv_eval.evaluate on padded arrays: boxes /= scale in float32, score filter,
    descending sort, truncation (csv_eval.py:95-110), matching, AP."""
    sc_l, lb_l, tp_l, an_l = [], [], [], []
    for i in range(len(det_counts)):
        d = int(det_counts[i])
        boxes = (np.asarray(det_boxes[i][:d], np.float32) / np.float32(scales[i])).astype(np.float32)
        scores = np.asarray(det_scores[i][:d], np.float32)
        labels = np.asarray(det_labels[i][:d])
        idx = np.where(scores > score_threshold)[0]
        order = idx[np.argsort(-scores[idx], kind="stable")][:max_detections]
        ann = np.asarray(annotations[i], np.float64)
        ann = ann[ann[:, 4] != -1]
        sc_l.append(scores[order])
        lb_l.append(labels[order])
        tp_l.append(match_detections(boxes[order], labels[order], ann, iou_threshold))
        an_l.append(ann)
    return average_precisions(sc_l, lb_l, tp_l, an_l, num_classes)
