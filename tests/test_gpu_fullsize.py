"""Full-size runs of the benchmark configurations on the GPU.

The oracle is too slow for whole batches at these sizes, so each configuration is checked
(a) against the oracle on a slice it finishes in seconds and (b) through size-independent
properties on the whole batch: linearity in the upstream gradients, agreement of a batched call
with per-image calls, consistency of num_pos with the assignment states, determinism, and for
NMS sortedness, idempotence and the "no two kept boxes overlap" invariant.
"""
import numpy as np
import pytest
import torch

from conftest import REG_ATOL_REL, RTOL, grad_close
from oracle import dense_head_oracle as O

pytestmark = pytest.mark.gpu


# (name, batch, image size, classes, padded GT rows, valid GT rows per image)
CONFIGS = {
    "dummy512": (2, 512, 600, 10, 10),         # configs[0]: datasets/dummy_dataset.py:10-32 (labels 1..599)
    "coco608": (16, 608, 80, 20, None),        # the bench.py workload
    "oid800": (8, 800, 601, 8, 8),             # configs[2]: 601-class head, rows of 2404 bytes
    "dense1024": (4, 1024, 80, 256, 256),      # configs[3]: 256 GT boxes per image
}


@pytest.fixture(scope="module")
def rh():
    import retinahead
    retinahead.load_library()
    assert torch.cuda.is_available()
    return retinahead


def make_inputs(rh, name, seed=0):
    bsz, size, c, m, valid = CONFIGS[name]
    dev = torch.device("cuda:0")
    anchors = rh.Anchors()(torch.zeros(1, 3, size, size, device=dev))
    n = anchors.shape[1]
    gen = torch.Generator(device=dev).manual_seed(seed)
    cls = torch.sigmoid(torch.randn((bsz, n, c), device=dev, generator=gen) * 2.0 - 4.595)
    reg = torch.randn((bsz, n, 4), device=dev, generator=gen)
    rng = np.random.default_rng(seed)
    ann = -np.ones((bsz, m, 5), np.float32)
    for j in range(bsz):
        v = valid if valid is not None else int(rng.integers(1, m + 1))
        cx, cy = rng.uniform(0, size, v), rng.uniform(0, size, v)
        w = np.exp(rng.uniform(np.log(16), np.log(size / 2), v))
        h = np.exp(rng.uniform(np.log(16), np.log(size / 2), v))
        x1, y1 = np.clip(cx - w / 2, 0, size - 1), np.clip(cy - h / 2, 0, size - 1)
        x2, y2 = np.clip(cx + w / 2, x1 + 1, size), np.clip(cy + h / 2, y1 + 1, size)
        ann[j, :v, :4] = np.stack([x1, y1, x2, y2], 1)
        # dummy_dataset.py:22 draws its labels from [1, 600)
        ann[j, :v, 4] = rng.integers(1 if name == "dummy512" else 0, c, v)
    return cls, reg, anchors, torch.from_numpy(ann).to(dev), ann


@pytest.mark.parametrize("name", list(CONFIGS))
def test_loss_full_size(rh, name):
    F = rh.functional
    cls, reg, anchors, ann, ann_np = make_inputs(rh, name)
    bsz, n, c = cls.shape
    out = F.focal_loss_fwd_bwd(cls, reg, anchors, ann)
    losses = out["losses"].cpu().numpy().copy()
    per_image = out["per_image"].cpu().numpy().copy()
    num_pos = out["num_pos"].cpu().numpy().copy()
    d_cls, d_reg = out["d_classifications"], out["d_regressions"]
    assert np.isfinite(losses).all() and int(out["status"].max()) == 0

    # (a) oracle on one image (batch_divisor keeps the batch's 1/B scaling)
    j = bsz - 1
    ref = O.focal_loss_fwd_bwd(cls[j:j + 1].cpu().numpy(), reg[j:j + 1].cpu().numpy(),
                               anchors.cpu().numpy(), ann_np[j:j + 1], batch_divisor=bsz)
    assert num_pos[j] == ref["num_pos"][0]
    assert np.isclose(per_image[0, j], ref["cls_per_image"][0], rtol=RTOL)
    assert np.isclose(per_image[1, j], ref["reg_per_image"][0], rtol=RTOL)
    # the suite's one gradient bar: no floor for dL/dp, REG_ATOL_REL of the largest for dL/dr
    grad_close(d_cls[j].cpu().numpy(), ref["d_classifications"][0])
    grad_close(d_reg[j].cpu().numpy(), ref["d_regressions"][0], atol_rel=REG_ATOL_REL, kind="d_reg")

    # (b) batch mean of the per-image losses (losses.py:139-141)
    assert np.allclose(losses, per_image.mean(axis=1), rtol=1e-6)

    # (c) num_pos == number of POSITIVE states from the assignment entry point; gradients of
    #     regressions are non-zero exactly on positive rows
    asg = F.assign(anchors, ann, c)
    pos = asg["state"] == 2
    assert np.array_equal(pos.sum(dim=1).cpu().numpy(), num_pos)
    assert not bool((d_reg != 0).any(dim=2)[~pos].any())
    # ignored rows (state 1) and images without GT get exactly zero classification gradient
    ign = asg["state"] == 1
    assert not bool((d_cls != 0).any(dim=2)[ign].any())

    # (d) linearity in the upstream gradients, bit for bit for a power-of-two scale
    out2 = F.focal_loss_fwd_bwd(cls, reg, anchors, ann, grad_cls=2.0, grad_reg=0.5)
    assert torch.equal(out2["d_classifications"], d_cls * 2.0)
    assert torch.equal(out2["d_regressions"], d_reg * 0.5)
    assert np.array_equal(out2["losses"].cpu().numpy(), losses)          # (e) and deterministic

    # (f) a batched call agrees with per-image calls (tile boundaries differ, so sums may differ
    #     in the last bits; assignments and gradients of whole rows must be identical)
    for jj in (0, bsz // 2):
        one = F.focal_loss_fwd_bwd(cls[jj:jj + 1], reg[jj:jj + 1], anchors, ann[jj:jj + 1],
                                   batch_divisor=bsz)
        assert int(one["num_pos"][0]) == num_pos[jj]
        assert np.allclose(one["per_image"].cpu().numpy()[:, 0], per_image[:, jj], rtol=1e-6)
        assert torch.equal(one["d_classifications"][0], d_cls[jj])
        assert torch.equal(one["d_regressions"][0], d_reg[jj])


def test_dense_gt_assignment_full_size(rh):
    """configs[3]: 196 416 anchors x 256 GT boxes, every integer output against the oracle."""
    cls, reg, anchors, ann, ann_np = make_inputs(rh, "dense1024", seed=3)
    asg = rh.functional.assign(anchors, ann, cls.shape[2])
    ref = O.assign(anchors.cpu().numpy()[0], ann_np[1])
    assert np.array_equal(asg["state"][1].cpu().numpy(), ref["state"])
    assert np.array_equal(asg["argmax"][1].cpu().numpy(), ref["argmax"])
    assert np.array_equal(asg["label"][1].cpu().numpy(), ref["label"])
    assert np.array_equal(asg["iou_max"][1].cpu().numpy(), ref["iou_max"])
    assert int(asg["num_pos"][1]) == ref["num_pos"]


def infer_inputs(rh, bsz, size, c, seed=0, mu=-7.0):
    dev = torch.device("cuda:0")
    anchors = rh.Anchors()(torch.zeros(1, 3, size, size, device=dev))
    n = anchors.shape[1]
    gen = torch.Generator(device=dev).manual_seed(seed)
    cls = torch.sigmoid(torch.randn((bsz, n, c), device=dev, generator=gen) * 1.2 + mu)
    reg = torch.randn((bsz, n, 4), device=dev, generator=gen) * 0.5
    return cls, reg, anchors


@pytest.mark.parametrize("mode,topk,shape", [("agnostic", None, None), ("per_class", None, None),
                                             ("agnostic", 1000, None),
                                             ("agnostic", None, "dummy512"), ("per_class", None, "dummy512")])
def test_postprocess_full_size(rh, mode, topk, shape):
    size, c, bsz, mu = 608, 80, 16, -7.0
    if topk:                      # configs[3]: 1024 px, top-1000 before NMS at 0.05
        size, bsz = 1024, 4
    if shape == "dummy512":       # configs[0]: batch 2 at 512 px, 600 classes (a maximum over 600
        bsz, size, c = CONFIGS[shape][:3]      # elements: lower logits for a similar candidate rate)
        mu = -8.0
    cls, reg, anchors = infer_inputs(rh, bsz, size, c, mu=mu)
    res = rh.postprocess(cls, reg, anchors, (size, size), mode=mode, pre_nms_topk=topk)
    assert len(res) == bsz
    boxes_all = rh.ClipBoxes()(rh.BBoxTransform()(anchors, reg), torch.zeros(1, 3, size, size, device="cuda"))
    score_all, label_all = cls.max(dim=2)
    for b in range(bsz):
        r = res[b]
        s, l, bx = r["scores"], r["labels"], r["boxes"]
        k = s.numel()
        assert k > 0 and l.dtype == torch.int64 and bx.shape == (k, 4)
        assert bool((s[:-1] >= s[1:]).all())                     # descending scores
        assert bool((s > 0.05).all())
        # every output row is one of the image's anchors: same score, label and decoded box
        cand = torch.nonzero(score_all[b] > 0.05).flatten()
        cand_boxes = boxes_all[b][cand]
        match = (bx[:, None, :] == cand_boxes[None, :, :]).all(dim=2) & \
                (s[:, None] == score_all[b][cand][None, :])
        assert bool(match.any(dim=1).all())
        first = match.float().argmax(dim=1)
        assert torch.equal(l, label_all[b][cand][first])
        # no kept box overlaps an earlier kept box (of the same label, per class) by more than 0.5
        iou = rh.calc_iou(bx, bx)
        bad = torch.triu(iou > 0.5, diagonal=1)
        if mode == "per_class":
            bad &= l[:, None] == l[None, :]
        assert not bool(bad.any())
        # idempotence: NMS of the kept set keeps everything, in the same order
        again = rh.nms(bx, s, 0.5) if mode == "agnostic" else rh.batched_nms(bx, s, l, 0.5)
        assert torch.equal(again, torch.arange(k, device=again.device))
        if topk:
            assert k <= topk
    # one image against the oracle, fed with the device's own decoded boxes
    b = 1
    so, lo, bo = O.postprocess_image(cls[b].cpu().numpy(), reg[b].cpu().numpy(), anchors.cpu().numpy(),
                                     size, size, mode=mode, pre_nms_topk=topk,
                                     boxes=boxes_all[b].cpu().numpy())
    assert np.array_equal(res[b]["scores"].cpu().numpy(), so)
    assert np.array_equal(res[b]["labels"].cpu().numpy(), lo)
    assert np.array_equal(res[b]["boxes"].cpu().numpy(), bo)


@pytest.mark.parametrize("mode", ["agnostic", "per_class"])
def test_postprocess_one_image_608(rh, mode):
    """The call shape of the reference's inference scripts (visualize.py:95, evaluate.py:95,
    csv_eval.py:89, coco_eval.py:27): ONE 608 px image per call on one stream, here two different
    images through the same call twice each (buffers of the first call are reused by the later
    ones).  Scores, labels and boxes against the oracle bit for bit, fed with the device's own
    decoded boxes; and the same detections as the image gets inside a batch of 16."""
    size, c = 608, 80
    cls, reg, anchors = infer_inputs(rh, 16, size, c, seed=5)
    batch = rh.postprocess(cls, reg, anchors, (size, size), mode=mode)
    boxes_all = rh.ClipBoxes()(rh.BBoxTransform()(anchors, reg), torch.zeros(1, 3, size, size, device="cuda"))
    for rep in range(2):
        for b in (3, 12):
            one = rh.postprocess(cls[b:b + 1].contiguous(), reg[b:b + 1].contiguous(), anchors, (size, size),
                                 mode=mode)
            assert len(one) == 1 and one[0]["scores"].numel() > 100
            so, lo, bo = O.postprocess_image(cls[b].cpu().numpy(), reg[b].cpu().numpy(), anchors.cpu().numpy(),
                                             size, size, mode=mode, boxes=boxes_all[b].cpu().numpy())
            assert np.array_equal(one[0]["scores"].cpu().numpy(), so), (rep, b)
            assert np.array_equal(one[0]["labels"].cpu().numpy(), lo), (rep, b)
            assert np.array_equal(one[0]["boxes"].cpu().numpy(), bo), (rep, b)
            for k in ("scores", "labels", "boxes"):
                assert torch.equal(one[0][k], batch[b][k]), (rep, b, k)


def test_nms_properties_large(rh):
    rng = np.random.default_rng(0)
    k = 12000
    c = rng.uniform(0, 2000, (k, 2))
    wh = rng.uniform(10, 150, (k, 2))
    boxes = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)).cuda()
    scores = torch.from_numpy(rng.random(k).astype(np.float32)).cuda()
    keep = rh.nms(boxes, scores, 0.5)
    ks = scores[keep]
    assert bool((ks[:-1] >= ks[1:]).all())
    iou = rh.calc_iou(boxes[keep], boxes[keep])
    assert not bool(torch.triu(iou > 0.5, diagonal=1).any())
    # every dropped box overlaps a kept box with a score at least as high
    dropped = torch.ones(k, dtype=torch.bool, device="cuda")
    dropped[keep] = False
    d_idx = torch.nonzero(dropped).flatten()[:3000]
    over = rh.calc_iou(boxes[d_idx], boxes[keep]) > 0.5
    higher = scores[keep][None, :] >= scores[d_idx][:, None]
    assert bool((over & higher).any(dim=1).all())
    assert np.array_equal(keep.cpu().numpy(), O.nms(boxes.cpu().numpy(), scores.cpu().numpy(), 0.5))


def to_levels(flat, size, last, a=9):
    """[B, N, last] -> per-level [B, A*last, H, W] (inverse of the reference's permute / view / cat,
    model.py:152-154,196-202,311-313)."""
    out, at = [], 0
    for l in (3, 4, 5, 6, 7):
        h = w = (size + 2 ** l - 1) // 2 ** l
        k = h * w * a
        out.append(flat[:, at:at + k].reshape(flat.shape[0], h, w, a * last).permute(0, 3, 1, 2).contiguous())
        at += k
    assert at == flat.shape[1]
    return out


@pytest.mark.parametrize("name,from_logits", [("coco608", False), ("coco608", True), ("oid800", True)])
def test_head_layout_loss_full_size(rh, name, from_logits):
    """Loss on the per-level head tensors at full size: gradients bit-identical to the flat path."""
    F = rh.functional
    cls, reg, anchors, ann, _ = make_inputs(rh, name)
    size = CONFIGS[name][1]
    if from_logits:
        cls = torch.logit(cls.double()).float()
    flat = F.focal_loss_fwd_bwd(cls, reg, anchors, ann, from_logits=from_logits)
    c = cls.shape[2]
    lv = F.focal_loss_levels_fwd_bwd(to_levels(cls, size, c), to_levels(reg, size, 4), anchors, ann,
                                     from_logits=from_logits)
    assert torch.equal(lv["num_pos"], flat["num_pos"])
    assert np.allclose(lv["losses"].cpu().numpy(), flat["losses"].cpu().numpy(), rtol=1e-6)
    assert np.allclose(lv["per_image"].cpu().numpy(), flat["per_image"].cpu().numpy(), rtol=1e-6)
    for got, want in zip(lv["d_cls_levels"], to_levels(flat["d_classifications"], size, c)):
        assert torch.equal(got, want)
    for got, want in zip(lv["d_reg_levels"], to_levels(flat["d_regressions"], size, 4)):
        assert torch.equal(got, want)


def test_head_layout_postprocess_full_size(rh):
    size, c, bsz = 608, 80, 16
    cls, reg, anchors = infer_inputs(rh, bsz, size, c)
    flat = rh.postprocess(cls, reg, anchors, (size, size))
    lv = rh.postprocess(to_levels(cls, size, c), to_levels(reg, size, 4), anchors, (size, size))
    for a, b in zip(lv, flat):
        assert a["scores"].numel() > 0
        for k in ("scores", "labels", "boxes"):
            assert torch.equal(a[k], b[k])
