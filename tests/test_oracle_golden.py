"""Pin the CPU oracle against outputs of the reference's own files (tests/golden)."""
import hashlib

import numpy as np
import pytest

from conftest import golden_cases, load_golden, logits_grad_close
from oracle import dense_head_oracle as O

RTOL = 1e-5          # losses and gradients: 1e-5 relative (fp32)


def grad_close(got, want, rtol=RTOL):
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64)
    atol = 1e-7 * float(np.max(np.abs(want))) if want.size else 0.0
    bad = np.abs(got - want) > rtol * np.abs(want) + atol
    assert not bad.any(), f"{bad.sum()} elements off; worst {np.max(np.abs(got - want))}"
    # exact zeros of the reference (ignored rows, clamped probabilities, empty images) stay zero
    assert np.array_equal(got == 0, want == 0)


def test_anchors_bit_exact():
    g = load_golden("anchors.npz")
    for key in g.files:
        if key.startswith("a_"):
            _, h, w = key.split("_")
            mine = O.anchors_for_image(int(h), int(w))
            assert mine.dtype == np.float32 and mine.shape == g[key].shape
            assert np.array_equal(mine, g[key]), key
        if key.startswith("sha_"):
            _, h, w = key.split("_")
            mine = O.anchors_for_image(int(h), int(w))
            assert mine.shape[1] == int(g[f"n_{h}_{w}"][0]) == O.num_anchors(int(h), int(w))
            assert hashlib.sha256(mine.tobytes()).digest() == g[key].tobytes()
            assert np.array_equal(mine[0, :18], g[f"head_{h}_{w}"])
            assert np.array_equal(mine[0, -18:], g[f"tail_{h}_{w}"])


def test_survey_anchor_counts():
    assert [O.num_anchors(s, s) for s in (512, 608, 800, 1024)] == [49104, 69354, 120087, 196416]


def test_calc_iou_bit_exact():
    g = load_golden("calc_iou.npz")
    assert np.array_equal(O.calc_iou(g["a"], g["b"]), g["iou"])


@pytest.mark.parametrize("case", ["normal", "holes", "no_pos", "ties", "clamp", "c1", "c13", "c80",
                                  "m256", "handmade", "all_empty"])
def test_focal_loss_matches_reference(focal_golden, case):
    d = focal_golden[case]
    bsz = d["cls"].shape[0]
    # assignment: bit exact
    for j in range(bsz):
        asg = O.assign(d["anchors"][0], d["ann"][j])
        if asg["num_gt"] == 0:
            assert d["num_pos"][j] == 0
            continue
        assert np.array_equal(asg["iou_max"], d["iou_max"][j])
        assert np.array_equal(asg["argmax"], d["argmax"][j])
        assert asg["num_pos"] == int(d["num_pos"][j])
    out = O.focal_loss_fwd_bwd(d["cls"], d["reg"], d["anchors"], d["ann"],
                               grad_cls=float(d["g"][0]), grad_reg=float(d["g"][1]))
    assert np.allclose(out["classification_loss"], d["cls_loss"], rtol=RTOL, atol=0)
    assert np.allclose(out["regression_loss"], d["reg_loss"], rtol=RTOL, atol=0)
    assert out["classification_loss"].shape == (1,) and out["regression_loss"].shape == (1,)
    grad_close(out["d_classifications"], d["d_cls"])
    grad_close(out["d_regressions"], d["d_reg"])


@pytest.mark.parametrize("case", ["normal", "c80", "extreme"])
def test_focal_loss_on_logits_matches_reference(focal_logits_golden, case):
    """Fixtures: the reference loss fed torch.sigmoid(logits) (what the head does, model.py:177,193),
    gradients taken with respect to the logits."""
    d = focal_logits_golden[case]
    out = O.focal_loss_fwd_bwd(d["cls"], d["reg"], d["anchors"], d["ann"], grad_cls=float(d["g"][0]),
                               grad_reg=float(d["g"][1]), from_logits=True)
    assert np.array_equal(out["num_pos"], d["num_pos"])
    assert np.allclose(out["classification_loss"], d["cls_loss"], rtol=RTOL, atol=0)
    assert np.allclose(out["regression_loss"], d["reg_loss"], rtol=RTOL, atol=0)
    logits_grad_close(out["d_classifications"], d["d_cls"], d["cls"])
    grad_close(out["d_regressions"], d["d_reg"])


def test_handmade_states(focal_golden):
    d = focal_golden["handmade"]
    asg = O.assign(d["anchors"][0], d["ann"][0])
    s = asg["state"]
    assert s[0] == O.STATE_POSITIVE and asg["label"][0] == 2      # IoU exactly 0.5
    assert s[1] == O.STATE_IGNORE                                  # IoU exactly 0.4
    assert s[2] == O.STATE_NEGATIVE
    assert s[3] == O.STATE_POSITIVE and s[4] == O.STATE_POSITIVE
    assert s[61] == O.STATE_POSITIVE and asg["label"][61] == 2


def test_ties_pick_first(focal_golden):
    d = focal_golden["ties"]
    asg = O.assign(d["anchors"][0], d["ann"][0])
    assert not (asg["argmax"] == 2).any()       # GT 2 duplicates GT 0


def test_decode_clip_score():
    g = load_golden("decode.npz")
    dec = O.bbox_transform(g["anchors"], g["deltas"])
    # exp() may differ in the last place between libraries, and the corner is a difference
    # of centre and half extent: 1e-6 relative to the largest coordinate of the box,
    scale = np.max(np.abs(g["decoded"]), axis=2, keepdims=True)
    assert (np.abs(dec.astype(np.float64) - g["decoded"]) <= 1e-6 * scale).all()
    h, w = int(g["hw"][0]), int(g["hw"][1])
    # ... while the clip itself is exact given the same input
    assert np.array_equal(O.clip_boxes(g["decoded"], h, w), g["clipped"])
    assert np.array_equal(g["probs"].max(axis=2), g["score"])
    assert np.array_equal(g["probs"].argmax(axis=2), g["label"])
    assert np.array_equal(g["score"] > np.float32(0.05), g["over"])


def test_generic_focal():
    g = load_golden("generic_focal.npz")
    t = g["targets"]
    for logits in (True, False):
        for reduce in (True, False):
            key = f"{'logits' if logits else 'probs'}_{'mean' if reduce else 'none'}"
            loss, grad = O.generic_focal_fwd_bwd(g[key + ".inp"], t, logits=logits, reduce=reduce)
            assert np.allclose(loss, g[key + ".loss"], rtol=2e-5, atol=1e-7), key
            assert np.allclose(grad, g[key + ".grad"], rtol=2e-4, atol=1e-6 if not reduce else 1e-8), key


def test_detection_evaluation_matches_reference():
    """Fixtures: csv_eval.evaluate run on stand-in dataset / network objects that hand it fixed
    annotations and detections (tools/gen_golden.py), and csv_eval.compute_overlap."""
    g = load_golden("evaluate.npz")
    assert np.array_equal(O.compute_overlap(g["overlap.a"], g["overlap.b"]), g["overlap.iou"])
    for tag, thr in (("t50", 0.5), ("t75", 0.75)):
        ap, num = O.evaluate_detections(g["det_boxes"], g["det_scores"], g["det_labels"], g["det_counts"],
                                        g["annotations"], g["scales"], 5, iou_threshold=thr,
                                        score_threshold=0.05, max_detections=25)
        assert np.array_equal(num, g[tag + ".num_annotations"])
        assert np.allclose(ap, g[tag + ".ap"], rtol=1e-12, atol=0), (ap, g[tag + ".ap"])


def test_global_normaliser_is_a_reweighting_of_the_reference_mode(focal_golden):
    """normalizer='global' has no counterpart in the reference (parity unpinned); it is pinned to
    the reference-checked per-image mode instead: image j's loss and gradients are the per-image
    ones times max(num_pos_j, 1) * B / max(P, 1) (classification) and num_pos_j * B / P
    (regression), P = all positives of the batch."""
    g = focal_golden["normal"]
    cls, reg, anc, ann = (g[k] for k in ("cls", "reg", "anchors", "ann"))
    per = O.focal_loss_fwd_bwd(cls, reg, anc, ann)
    glo = O.focal_loss_fwd_bwd(cls, reg, anc, ann, normalizer="global")
    npos = per["num_pos"].astype(np.float64)
    assert np.array_equal(per["num_pos"], glo["num_pos"])
    bsz, total = len(npos), npos.sum()
    assert total > 0 and len(set(npos.tolist())) > 1        # the modes differ on this fixture
    wc = np.maximum(npos, 1) / max(total, 1)
    wr = npos / total
    assert np.allclose(glo["cls_per_image"], per["cls_per_image"] * wc, rtol=1e-6)
    assert np.allclose(glo["reg_per_image"], per["reg_per_image"] * wr, rtol=1e-6)
    assert np.isclose(glo["classification_loss"][0], (per["cls_per_image"] * wc).sum(), rtol=1e-6)
    assert np.isclose(glo["regression_loss"][0], (per["reg_per_image"] * wr).sum(), rtol=1e-6)
    assert np.allclose(glo["d_classifications"],
                       per["d_classifications"] * (wc * bsz)[:, None, None], rtol=1e-5, atol=1e-30)
    assert np.allclose(glo["d_regressions"],
                       per["d_regressions"] * (wr * bsz)[:, None, None], rtol=1e-5, atol=1e-30)
    # a count supplied from outside (all ranks of a data-parallel job) replaces the batch's own
    twice = O.focal_loss_fwd_bwd(cls, reg, anc, ann, normalizer="global", num_pos_total=2 * int(total))
    assert np.isclose(twice["classification_loss"][0] * 2, glo["classification_loss"][0], rtol=1e-6)
    assert np.allclose(twice["d_regressions"] * 2, glo["d_regressions"], rtol=1e-6, atol=1e-30)


def test_annotation_filter_rule():
    """The datasets' filter restated (unpinned: the loaders cannot be imported here): known answers
    for the lines cited in the oracle."""
    boxes = np.array([[10, 10, 10.5, 50], [10, 10, 11, 11], [0, 0, 100, 0.99], [5, 5, 60, 70]], np.float32)
    labels = np.array([1, 2, 3, 4], np.float32)
    b, l = O.sanitize_annotations(boxes, labels)                     # csv_dataset.py:135
    assert l.tolist() == [2.0, 4.0] and np.array_equal(b, boxes[[1, 3]])
    xywh = np.array([[10, 20, 0.5, 30], [10, 20, 1, 1], [3, 4, 50, 60]], np.float32)
    b, l = O.sanitize_annotations(xywh, labels[:3], xywh=True)       # coco_dataset.py:98, 107-108
    assert l.tolist() == [2.0, 3.0] and b.tolist() == [[10, 20, 11, 21], [3, 4, 53, 64]]
    out = O.collate_annotations([(boxes, labels), (xywh[:0], labels[:0])], min_size=1.0)
    assert out.shape == (2, 4, 5) and (out[0, 2:] == -1).all() and (out[1] == -1).all()
