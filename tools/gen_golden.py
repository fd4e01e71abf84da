#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the reference's own leaf files.

Runs ONLY in the build container (needs /root/reference).  The reference files
are loaded by path, unmodified; two process-local shims make
models/retinanet/losses.py runnable on a CPU-only, current PyTorch:
  * ``Tensor.cuda`` -> identity (the loss hard-codes ``.cuda()``,
    losses.py:52-53,68,81,92,123,136);
  * ``1 - bool_tensor`` -> logical not (dead line losses.py:125 raises on
    torch >= 1.2; its result is never used).
The fixtures hold inputs and the reference's outputs (data only).
"""
import hashlib
import importlib.util
import os
import sys

import numpy as np
import torch

REF = os.environ.get("RETINANET_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install_shims():
    torch.Tensor.cuda = lambda self, *a, **k: self
    orig_rsub = torch.Tensor.__rsub__

    def rsub(self, other):
        if self.dtype == torch.bool:
            return ~self
        return orig_rsub(self, other)

    torch.Tensor.__rsub__ = rsub


def rand_boxes(rng, m, size, lo=8.0):
    cx = rng.uniform(0, size, m)
    cy = rng.uniform(0, size, m)
    w = np.exp(rng.uniform(np.log(lo), np.log(size / 2), m))
    h = np.exp(rng.uniform(np.log(lo), np.log(size / 2), m))
    x1 = np.clip(cx - w / 2, 0, size - 1)
    y1 = np.clip(cy - h / 2, 0, size - 1)
    x2 = np.clip(cx + w / 2, x1 + 1, size)
    y2 = np.clip(cy + h / 2, y1 + 1, size)
    return np.stack([x1, y1, x2, y2], 1).astype(np.float32)


def make_annotations(rng, bsz, m, size, c, valid_counts):
    ann = -np.ones((bsz, m, 5), np.float32)
    for j, v in enumerate(valid_counts):
        if v:
            ann[j, :v, :4] = rand_boxes(rng, v, size)
            ann[j, :v, 4] = rng.integers(0, c, v).astype(np.float32)
    return ann


def run_loss(losses, cls, reg, anchors, ann, g_cls=1.0, g_reg=1.0, from_logits=False):
    """from_logits: ``cls`` holds logits; the reference loss is fed torch.sigmoid(cls), as the
    classification head does (model.py:177,193), and d_cls is the gradient wrt the logits."""
    tc = torch.from_numpy(cls).clone().requires_grad_(True)
    tr = torch.from_numpy(reg).clone().requires_grad_(True)
    probs = torch.sigmoid(tc) if from_logits else tc
    out = losses.FocalLoss()(probs, tr, torch.from_numpy(anchors), torch.from_numpy(ann))
    total = g_cls * out["classification_loss"].sum() + g_reg * out["regression_loss"].sum()
    if total.requires_grad:          # a batch of empty images yields constants (losses.py:52-53)
        total.backward()
    # per-image assignment, straight from the reference's calc_iou + torch.max
    bsz, n = cls.shape[0], cls.shape[1]
    iou_max = np.zeros((bsz, n), np.float32)
    argmax = np.zeros((bsz, n), np.int32)
    num_pos = np.zeros(bsz, np.int32)
    for j in range(bsz):
        a = torch.from_numpy(ann[j])
        a = a[a[:, 4] != -1]
        if a.shape[0] == 0:
            continue
        iou = losses.calc_iou(torch.from_numpy(anchors[0]), a[:, :4])
        mx, am = torch.max(iou, dim=1)
        iou_max[j] = mx.numpy()
        argmax[j] = am.numpy().astype(np.int32)
        num_pos[j] = int(torch.ge(mx, 0.5).sum())
    return dict(cls=cls, reg=reg, anchors=anchors, ann=ann,
                g=np.array([g_cls, g_reg], np.float32),
                cls_loss=out["classification_loss"].detach().numpy(),
                reg_loss=out["regression_loss"].detach().numpy(),
                d_cls=tc.grad.numpy() if tc.grad is not None else np.zeros_like(cls),
                d_reg=tr.grad.numpy() if tr.grad is not None else np.zeros_like(reg),
                iou_max=iou_max, argmax=argmax, num_pos=num_pos)


def train_probs(rng, shape):
    z = rng.normal(-4.595, 2.0, shape)
    return (1.0 / (1.0 + np.exp(-z))).astype(np.float32)


def main():
    os.makedirs(OUT, exist_ok=True)
    install_shims()
    losses = load("ref_losses", "models/retinanet/losses.py")
    anchors_mod = load("ref_anchors", "models/retinanet/anchors.py")
    utils = load("ref_utils", "models/retinanet/utils.py")
    gfocal = load("ref_focal", "models/focal_loss.py")

    # ---------------- anchors ----------------
    anc = {}
    for (h, w) in [(64, 64), (72, 40), (100, 333), (1, 1)]:
        a = anchors_mod.Anchors()(torch.zeros(1, 3, h, w)).numpy()
        anc[f"a_{h}_{w}"] = a
    for (h, w) in [(512, 512), (608, 608), (800, 1216), (1024, 1024)]:
        a = anchors_mod.Anchors()(torch.zeros(1, 3, h, w)).numpy()
        anc[f"n_{h}_{w}"] = np.array([a.shape[1]], np.int64)
        anc[f"sha_{h}_{w}"] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), np.uint8)
        anc[f"head_{h}_{w}"] = a[0, :18]
        anc[f"tail_{h}_{w}"] = a[0, -18:]
    np.savez_compressed(os.path.join(OUT, "anchors.npz"), **anc)

    # ---------------- calc_iou ----------------
    rng = np.random.default_rng(11)
    a = rand_boxes(rng, 200, 128.0)
    b = rand_boxes(rng, 9, 128.0)
    b[3] = b[2]                                   # duplicate GT
    b[5] = [10, 10, 10, 10]                       # zero-area GT
    a[0] = [5, 5, 5, 5]                           # zero-area anchor
    a[1] = b[0]                                   # identical boxes -> IoU 1
    iou = losses.calc_iou(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    np.savez_compressed(os.path.join(OUT, "calc_iou.npz"), a=a, b=b, iou=iou)

    # ---------------- loss cases ----------------
    def std_anchors(size):
        return anchors_mod.Anchors()(torch.zeros(1, 3, size, size)).numpy()

    cases = {}
    rng = np.random.default_rng(101)
    # normal: 3 images, one with only padding rows, C=4
    A = std_anchors(64)
    n = A.shape[1]
    ann = make_annotations(rng, 3, 5, 64, 4, [3, 0, 5])
    cases["normal"] = run_loss(losses, train_probs(rng, (3, n, 4)),
                               rng.normal(0, 1, (3, n, 4)).astype(np.float32), A, ann, 0.7, 1.3)
    # valid rows after a padding row; label -1 in the middle
    ann = make_annotations(rng, 2, 6, 64, 4, [4, 6])
    ann[0, 1] = -1
    ann[1, 0] = -1
    cases["holes"] = run_loss(losses, train_probs(rng, (2, n, 4)),
                              rng.normal(0, 1, (2, n, 4)).astype(np.float32), A, ann)
    # no positives: GT far too small for any standard anchor
    ann = -np.ones((1, 3, 5), np.float32)
    ann[0, 0] = [10, 10, 12, 12, 1]
    ann[0, 1] = [40, 41, 43, 44, 0]
    cases["no_pos"] = run_loss(losses, train_probs(rng, (1, n, 3)),
                               rng.normal(0, 1, (1, n, 4)).astype(np.float32), A, ann)
    # duplicate GT boxes with different labels -> argmax picks the first
    ann = make_annotations(rng, 1, 4, 64, 5, [4])
    ann[0, 2, :4] = ann[0, 0, :4]
    ann[0, 2, 4] = (ann[0, 0, 4] + 1) % 5
    cases["ties"] = run_loss(losses, train_probs(rng, (1, n, 5)),
                             rng.normal(0, 1, (1, n, 4)).astype(np.float32), A, ann)
    # probabilities on and outside the clamp bounds
    p = train_probs(rng, (1, n, 4))
    special = np.array([0.0, 1.0, 1e-4, 1.0 - 1e-4, 1e-5, 1.0 - 1e-6, 0.5, 9.9e-5,
                        np.float32(1e-4), np.nextafter(np.float32(1e-4), np.float32(0)),
                        np.float32(1.0 - 1e-4), np.nextafter(np.float32(1.0 - 1e-4), np.float32(1))],
                       np.float32)
    p.reshape(-1)[: special.size * 40] = np.tile(special, 40)
    ann = make_annotations(rng, 1, 4, 64, 4, [4])
    cases["clamp"] = run_loss(losses, p, rng.normal(0, 1, (1, n, 4)).astype(np.float32), A, ann)
    # single class
    ann = make_annotations(rng, 2, 3, 64, 1, [3, 2])
    cases["c1"] = run_loss(losses, train_probs(rng, (2, n, 1)),
                           rng.normal(0, 1, (2, n, 4)).astype(np.float32), A, ann)
    # odd class count (rows not 16-byte aligned), 2 images
    ann = make_annotations(rng, 2, 7, 64, 13, [7, 2])
    cases["c13"] = run_loss(losses, train_probs(rng, (2, n, 13)),
                            rng.normal(0, 1, (2, n, 4)).astype(np.float32), A, ann)
    # COCO-like width
    ann = make_annotations(rng, 2, 8, 64, 80, [8, 3])
    cases["c80"] = run_loss(losses, train_probs(rng, (2, n, 80)),
                            rng.normal(0, 1, (2, n, 4)).astype(np.float32), A, ann)
    # dense GT: 256 boxes on a 128 px image
    A128 = std_anchors(128)
    n128 = A128.shape[1]
    ann = make_annotations(rng, 1, 256, 128, 6, [256])
    cases["m256"] = run_loss(losses, train_probs(rng, (1, n128, 6)),
                             rng.normal(0, 1, (1, n128, 4)).astype(np.float32), A128, ann)
    # hand-made anchors: IoU exactly 0.5 / 0.4, GT narrower than 1 px, large regressions
    H = np.zeros((1, 70, 4), np.float32)
    H[0, :, 0] = np.arange(70) * 20.0
    H[0, :, 2] = H[0, :, 0] + 10.0
    H[0, :, 3] = 10.0
    H[0, 60:, 2] = H[0, 60:, 0] + 0.75           # anchors 0.75 wide
    H[0, 60:, 3] = 0.75
    ann = -np.ones((1, 8, 5), np.float32)
    ann[0, 0] = [0, 0, 10, 5, 2]                  # IoU 0.5 with anchor 0 -> positive
    ann[0, 1] = [20, 0, 30, 4, 1]                 # IoU 0.4 with anchor 1 -> ignore
    ann[0, 2] = [40, 0, 50, 3.9, 0]               # IoU 0.39 -> negative
    ann[0, 3] = [60, 0, 70, 10, 3]                # IoU 1
    ann[0, 4] = [1200, 0, 1200.5, 0.5, 1]         # inside anchor 60: w,h < 1 (clamped to 1)
    ann[0, 5] = [1220, 0, 1220.75, 0.75, 2]       # identical to anchor 61
    ann[0, 6] = [80, 1, 90, 9, 0]                 # IoU 0.8
    reg = rng.normal(0, 3, (1, 70, 4)).astype(np.float32)
    cases["handmade"] = run_loss(losses, train_probs(rng, (1, 70, 4)), reg, H, ann)
    # every image empty
    ann = -np.ones((2, 2, 5), np.float32)
    cases["all_empty"] = run_loss(losses, train_probs(rng, (2, n, 4)),
                                  rng.normal(0, 1, (2, n, 4)).astype(np.float32), A, ann)
    flat = {}
    for name, d in cases.items():
        for k, v in d.items():
            flat[f"{name}.{k}"] = v
    np.savez_compressed(os.path.join(OUT, "focal_loss.npz"), **flat)

    # ---------------- loss on logits (sigmoid folded in) ----------------
    rng = np.random.default_rng(151)
    lcases = {}
    ann = make_annotations(rng, 3, 5, 64, 4, [3, 0, 5])
    lcases["normal"] = run_loss(losses, rng.normal(-4.595, 2.0, (3, n, 4)).astype(np.float32),
                                rng.normal(0, 1, (3, n, 4)).astype(np.float32), A, ann, 0.7, 1.3,
                                from_logits=True)
    ann = make_annotations(rng, 2, 8, 64, 80, [8, 3])
    lcases["c80"] = run_loss(losses, rng.normal(-4.595, 2.0, (2, n, 80)).astype(np.float32),
                             rng.normal(0, 1, (2, n, 4)).astype(np.float32), A, ann,
                             from_logits=True)
    # logits around the clamp bounds (sigmoid = 1e-4 at -9.21) and far outside any range
    x = rng.normal(-4.595, 2.0, (1, n, 4)).astype(np.float32)
    special = np.array([0.0, -9.21034, 9.21034, -9.3, 9.3, -9.1, 9.1, -20.0, 20.0, -100.0, 100.0,
                        -1e4, 1e4, 6.9, 8.5, -0.0], np.float32)
    x.reshape(-1)[: special.size * 40] = np.tile(special, 40)
    ann = make_annotations(rng, 1, 4, 64, 4, [4])
    lcases["extreme"] = run_loss(losses, x, rng.normal(0, 1, (1, n, 4)).astype(np.float32), A, ann,
                                 from_logits=True)
    flat = {}
    for name, d in lcases.items():
        for k, v in d.items():
            flat[f"{name}.{k}"] = v
    np.savez_compressed(os.path.join(OUT, "focal_loss_logits.npz"), **flat)

    # ---------------- decode + clip + score threshold ----------------
    rng = np.random.default_rng(202)
    A = anchors_mod.Anchors()(torch.zeros(1, 3, 72, 40)).numpy()
    n = A.shape[1]
    deltas = rng.normal(0, 0.5, (3, n, 4)).astype(np.float32)
    deltas[0, :5] *= 20                                   # large exp()
    dec = utils.BBoxTransform()(torch.from_numpy(A), torch.from_numpy(deltas))
    clipped = utils.ClipBoxes()(dec.clone(), torch.zeros(3, 3, 72, 40))
    z = rng.normal(-4.0, 1.5, (3, n, 6))
    probs = (1.0 / (1.0 + np.exp(-z))).astype(np.float32)
    probs[1, 7, 2] = probs[1, 7, 4] = 0.9                 # tie inside a row -> first index
    tp = torch.from_numpy(probs)
    mx, am = torch.max(tp, dim=2)
    np.savez_compressed(os.path.join(OUT, "decode.npz"), anchors=A, deltas=deltas,
                        decoded=dec.numpy(), clipped=clipped.numpy(), hw=np.array([72, 40]),
                        probs=probs, score=mx.numpy(), label=am.numpy(),
                        over=(mx > 0.05).numpy())

    # ---------------- generic focal ----------------
    rng = np.random.default_rng(303)
    g = {}
    x = rng.normal(0, 3, (37, 11)).astype(np.float32)
    t = (rng.random((37, 11)) < 0.2).astype(np.float32)
    for logits in (True, False):
        for reduce in (True, False):
            inp = x if logits else (1 / (1 + np.exp(-x))).astype(np.float32)
            ti = torch.from_numpy(inp).clone().requires_grad_(True)
            out = gfocal.FocalLoss(logits=logits, reduce=reduce)(ti, torch.from_numpy(t))
            out.sum().backward()
            key = f"{'logits' if logits else 'probs'}_{'mean' if reduce else 'none'}"
            g[key + ".loss"] = out.detach().numpy()
            g[key + ".grad"] = ti.grad.numpy()
            g[key + ".inp"] = inp
    g["targets"] = t
    np.savez_compressed(os.path.join(OUT, "generic_focal.npz"), **g)

    # ---------------- detection evaluation (csv_eval.py:11-35, 149-237) ----------------
    # evaluate() wants a dataset and a network; both are stand-ins that hand it fixed
    # annotations and fixed detections, so what is recorded is the reference's own matching
    # (compute_overlap, greedy assignment per class) and its AP arithmetic.
    csv_eval = load("ref_csv_eval", "csv_eval.py")
    rng = np.random.default_rng(303)
    n_img, n_cls, max_gt, max_det = 7, 5, 6, 40
    ann = -np.ones((n_img, max_gt, 5), np.float32)
    gt_counts = [4, 0, 6, 2, 5, 3, 1]
    for i, k in enumerate(gt_counts):
        x1, y1 = rng.integers(0, 300, k), rng.integers(0, 300, k)
        ann[i, :k, 0], ann[i, :k, 1] = x1, y1
        ann[i, :k, 2], ann[i, :k, 3] = x1 + rng.integers(20, 120, k), y1 + rng.integers(20, 120, k)
        ann[i, :k, 4] = rng.integers(0, n_cls - 1, k)          # the last class never occurs
    scales = np.array([1.0, 1.0, 0.5, 1.0, 2.0, 1.0, 1.0], np.float32)
    det_boxes = np.zeros((n_img, max_det, 4), np.float32)
    det_scores = np.zeros((n_img, max_det), np.float32)
    det_labels = np.zeros((n_img, max_det), np.int64)
    det_counts = np.zeros(n_img, np.int32)
    for i in range(n_img):
        k = gt_counts[i]
        rows = []
        for j in range(k):                                       # 0-3 jittered copies of each box
            for _ in range(int(rng.integers(0, 4))):
                jit = rng.normal(0, rng.choice([2.0, 15.0]), 4)
                lab = ann[i, j, 4] if rng.random() < 0.8 else rng.integers(0, n_cls)
                rows.append(np.concatenate([ann[i, j, :4] + jit, [lab]]))
        for _ in range(int(rng.integers(0, 8))):                 # clutter
            x1, y1 = rng.uniform(0, 300, 2)
            rows.append(np.array([x1, y1, x1 + rng.uniform(10, 100), y1 + rng.uniform(10, 100),
                                  rng.integers(0, n_cls)]))
        if i == 5:
            rows = []                                            # an image without detections
        rows = rows[:max_det]
        d = len(rows)
        det_counts[i] = d
        if d:
            r = np.array(rows)
            sc = np.sort(rng.uniform(0.06, 1.0, d).astype(np.float32))[::-1]   # distinct, descending
            det_boxes[i, :d] = (r[:, :4] * scales[i]).astype(np.float32)       # network-scale boxes
            det_scores[i, :d] = sc
            det_labels[i, :d] = r[:, 4].astype(np.int64)

    class Dataset:
        def __len__(self):
            return n_img

        def __getitem__(self, index):
            self.current = index
            return {"img": torch.zeros(8, 8, 3), "scale": float(scales[index])}

        def num_classes(self):
            return n_cls

        def load_annotations(self, index):
            return ann[index, :gt_counts[index]].astype(np.float64)

        def label_to_name(self, label):
            return str(label)

    data = Dataset()

    class Net:
        def eval(self):
            return self

        def __call__(self, img):
            i, d = data.current, int(det_counts[data.current])
            return (torch.from_numpy(det_scores[i, :d].copy()), torch.from_numpy(det_labels[i, :d].copy()),
                    torch.from_numpy(det_boxes[i, :d].copy()))

    ev = {}
    for tag, thr in (("t50", 0.5), ("t75", 0.75)):
        aps = csv_eval.evaluate(data, Net(), iou_threshold=thr, score_threshold=0.05, max_detections=25)
        ev[tag + ".ap"] = np.array([aps[c][0] for c in range(n_cls)], np.float64)
        ev[tag + ".num_annotations"] = np.array([aps[c][1] for c in range(n_cls)], np.float64)
    a = rng.uniform(0, 50, (9, 4))
    a[:, 2:] += a[:, :2]
    b = rng.uniform(0, 50, (5, 4))
    b[:, 2:] += b[:, :2]
    b[4] = [3, 3, 3, 3]
    ev["overlap.a"], ev["overlap.b"] = a, b
    ev["overlap.iou"] = csv_eval.compute_overlap(a, b)
    ev.update(annotations=ann, scales=scales, det_boxes=det_boxes, det_scores=det_scores,
              det_labels=det_labels, det_counts=det_counts)
    np.savez_compressed(os.path.join(OUT, "evaluate.npz"), **ev)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    sys.exit(main())
