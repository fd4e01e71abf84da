"""Parity of the HIP kernels (through the C ABI) with the golden vectors and the CPU oracle.

Bars: anchors, IoU, assignments and NMS keep lists bit exact; losses and gradients
within 1e-5 relative (float32).
"""
import hashlib

import numpy as np
import pytest
import torch

from conftest import FLOOR_STATS, REG_ATOL_REL, RTOL, grad_close, load_golden, logits_grad_close
from oracle import dense_head_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rh():
    import retinahead
    retinahead.load_library()
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return retinahead


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------ anchors
def test_anchors_golden(rh):
    g = load_golden("anchors.npz")
    mod = rh.Anchors()
    for key in g.files:
        if key.startswith("a_"):
            _, h, w = key.split("_")
            out = mod(torch.zeros(1, 3, int(h), int(w), device="cuda")).cpu().numpy()
            assert out.shape == g[key].shape and np.array_equal(out, g[key]), key
        if key.startswith("sha_"):
            _, h, w = key.split("_")
            out = mod(torch.zeros(2, 3, int(h), int(w), device="cuda")).cpu().numpy()
            assert out.shape == (1, int(g[f"n_{h}_{w}"][0]), 4)
            assert hashlib.sha256(out.tobytes()).digest() == g[key].tobytes(), key


def test_anchors_custom_config(rh):
    mod = rh.Anchors(pyramid_levels=[2, 4], ratios=[1.0, 3.0], scales=[1.0, 1.5, 2.0, 2.5])
    out = mod(torch.zeros(1, 3, 50, 70, device="cuda")).cpu().numpy()
    ref = O.anchors_for_image(50, 70, levels=(2, 4), ratios=(1.0, 3.0), scales=(1.0, 1.5, 2.0, 2.5))
    assert np.array_equal(out, ref)


# ------------------------------------------------------------------ IoU / assignment
def test_calc_iou_golden(rh):
    g = load_golden("calc_iou.npz")
    out = rh.calc_iou(dev(g["a"]), dev(g["b"])).cpu().numpy()
    assert np.array_equal(out, g["iou"])


@pytest.mark.parametrize("case", ["normal", "holes", "no_pos", "ties", "clamp", "c1", "c13", "c80",
                                  "m256", "handmade", "all_empty"])
def test_assign_and_loss_golden(rh, focal_golden, case):
    d = focal_golden[case]
    bsz, n, c = d["cls"].shape
    asg = rh.functional.assign(dev(d["anchors"]), dev(d["ann"]), c)
    for j in range(bsz):
        ref = O.assign(d["anchors"][0], d["ann"][j])
        assert np.array_equal(asg["state"][j].cpu().numpy(), ref["state"])
        assert np.array_equal(asg["label"][j].cpu().numpy(), ref["label"])
        assert int(asg["num_pos"][j]) == ref["num_pos"] == int(d["num_pos"][j])
        if ref["num_gt"]:
            assert np.array_equal(asg["iou_max"][j].cpu().numpy(), d["iou_max"][j])
            assert np.array_equal(asg["argmax"][j].cpu().numpy(), d["argmax"][j])
    # loss + gradients through the autograd wrapper, upstream gradients as in the fixture
    cls = dev(d["cls"]).requires_grad_(True)
    reg = dev(d["reg"]).requires_grad_(True)
    out = rh.FocalLoss()(cls, reg, dev(d["anchors"]), dev(d["ann"]))
    assert out["classification_loss"].shape == (1,) and out["regression_loss"].shape == (1,)
    total = float(d["g"][0]) * out["classification_loss"].sum() + float(d["g"][1]) * out["regression_loss"].sum()
    total.backward()
    assert np.allclose(out["classification_loss"].detach().cpu().numpy(), d["cls_loss"], rtol=RTOL, atol=0)
    assert np.allclose(out["regression_loss"].detach().cpu().numpy(), d["reg_loss"], rtol=RTOL, atol=0)
    grad_close(cls.grad.cpu().numpy(), d["d_cls"])
    grad_close(reg.grad.cpu().numpy(), d["d_reg"], atol_rel=REG_ATOL_REL, kind="d_reg")


@pytest.mark.parametrize("m", [1, 63, 64, 65, 130, 300, 1100])
def test_assign_ragged_annotation_rows(rh, m):
    """Valid rows scattered among padding rows, lists longer than one 64-row chunk, chunks
    without any valid row, boxes outside the image and degenerate boxes: states, arg-max (index
    into the valid list), labels, IoU maxima and counts bit for bit.  (The assignment kernel has
    a plain and a split form; rh_assign takes the split one above 128 rows, the loss, which wants
    the codes only, above 1024.)"""
    rng = np.random.default_rng(500 + m)
    size, c, bsz = 160, 7, 4
    anc = O.anchors_for_image(size, size)
    ann = -np.ones((bsz, m, 5), np.float32)
    for j in range(bsz):
        keep = rng.random(m) < (0.0, 0.15, 0.6, 1.0)[j]          # image 0: no valid row at all
        if j == 1 and m > 64:
            keep[:64] = False                                     # a whole chunk of padding first
            keep[64] = True
        v = int(keep.sum())
        x1, y1 = rng.uniform(-20, size, v), rng.uniform(-20, size, v)
        w, h = np.exp(rng.uniform(np.log(4), np.log(size), v)), np.exp(rng.uniform(np.log(4), np.log(size), v))
        rows = np.stack([x1, y1, x1 + w, y1 + h, rng.integers(0, c, v).astype(np.float64)], 1)
        if v > 2:
            rows[0, 2:4] = rows[0, 0:2]                           # zero-area box
            rows[1, :4] = rows[2, :4]                             # duplicate: the first one wins ties
        ann[j, keep] = rows.astype(np.float32)
    asg = rh.functional.assign(dev(anc), dev(ann), c)
    for j in range(bsz):
        ref = O.assign(anc[0], ann[j])
        assert int(asg["num_pos"][j]) == ref["num_pos"], j
        assert np.array_equal(asg["state"][j].cpu().numpy(), ref["state"]), j
        assert np.array_equal(asg["label"][j].cpu().numpy(), ref["label"]), j
        if ref["num_gt"]:
            assert np.array_equal(asg["argmax"][j].cpu().numpy(), ref["argmax"]), j
            assert np.array_equal(asg["iou_max"][j].cpu().numpy(), ref["iou_max"]), j
    assert int(asg["status"].max()) == 0
    # the fused loss sees the same lists (regression targets come from the published records)
    cls = rng.uniform(0.001, 0.6, (bsz, anc.shape[1], c)).astype(np.float32)
    reg = rng.normal(0, 1, (bsz, anc.shape[1], 4)).astype(np.float32)
    ref = O.focal_loss_fwd_bwd(cls, reg, anc, ann)
    out = rh.functional.focal_loss_fwd_bwd(dev(cls), dev(reg), dev(anc), dev(ann))
    assert np.array_equal(out["num_pos"].cpu().numpy(), ref["num_pos"])
    assert np.allclose(out["losses"].cpu().numpy(),
                       [ref["classification_loss"][0], ref["regression_loss"][0]], rtol=RTOL, atol=0)
    grad_close(out["d_regressions"].cpu().numpy(), ref["d_regressions"], atol_rel=REG_ATOL_REL, kind="d_reg")


def test_assign_ties_across_the_waves_of_the_split_form(rh):
    """Lists of more than 128 rows are scanned by four waves, every fourth 64-row chunk each, and
    merged: identical boxes placed in chunks of different waves (and in a later chunk of the same
    wave) must still resolve to the lowest GT index, the first arg-max of losses.py:61."""
    rng = np.random.default_rng(77)
    size, c, m = 192, 5, 300
    anc = O.anchors_for_image(size, size)
    ann = -np.ones((2, m, 5), np.float32)
    for j in range(2):
        x1, y1 = rng.uniform(0, size - 30, m), rng.uniform(0, size - 30, m)
        ann[j, :, :4] = np.stack([x1, y1, x1 + rng.uniform(10, 120, m), y1 + rng.uniform(10, 120, m)], 1)
        ann[j, :, 4] = rng.integers(0, c, m)
    twin = anc[0, 4000]                       # on top of an anchor: IoU 1 for it, ties for its copies
    for row in (260, 133, 69, 5):
        ann[0, row, :4] = twin
    ann[0, 5, 4], ann[0, 69, 4] = 1, 2        # different labels: the label must be the first row's
    ann[1, 10:200, 4] = -1                    # image 1: valid rows only in chunks 0, 3 and 4
    asg = rh.functional.assign(dev(anc), dev(ann), c)
    for j in range(2):
        ref = O.assign(anc[0], ann[j])
        assert int(asg["num_pos"][j]) == ref["num_pos"]
        assert np.array_equal(asg["argmax"][j].cpu().numpy(), ref["argmax"]), j
        assert np.array_equal(asg["iou_max"][j].cpu().numpy(), ref["iou_max"]), j
        assert np.array_equal(asg["state"][j].cpu().numpy(), ref["state"]), j
        assert np.array_equal(asg["label"][j].cpu().numpy(), ref["label"]), j
    assert int(asg["argmax"][0, 4000]) == 5 and int(asg["label"][0, 4000]) == 1


@pytest.mark.parametrize("m", [96, 400])
def test_loss_assignment_cull_on_the_thresholds(rh, m):
    """The loss wants packed codes only, so its assignment drops boxes that cannot reach the lower
    IoU threshold with any anchor of a wave (area ratio, share of a side inside the wave's
    bounding box).  Boxes built to sit ON the thresholds are where that is tightest: an anchor
    nested in the box (IoU = area ratio), the box nested in an anchor, a box that shares an
    anchor's full height and a threshold's share of its width, shifted copies -- each with IoU
    0.4 or 0.5 give or take a few ulp.  The loss's view of every anchor (positive: a regression
    gradient; ignored: no classification gradient; negative otherwise) must be the state that
    rh_assign, which culls nothing of the kind, and the oracle give."""
    rng = np.random.default_rng(900 + m)
    size, c, bsz = 256, 6, 2
    anc = O.anchors_for_image(size, size)
    n = anc.shape[1]
    ann = -np.ones((bsz, m, 5), np.float32)
    for j in range(bsz):
        rows = []
        for i in range(m):
            a = anc[0, rng.integers(0, n)].astype(np.float64)
            w, h = a[2] - a[0], a[3] - a[1]
            thr = (0.4, 0.5)[i % 2]
            f = thr * (1.0 + rng.choice([-2, -1, 0, 1, 2]) * 6e-8 * rng.integers(0, 3))
            kind = (i // 2) % 5
            if kind == 0:      # anchor nested in the box, same height: IoU = w / W
                b = [a[0], a[1], a[0] + w / f, a[3]]
            elif kind == 1:    # box nested in the anchor along x
                b = [a[0], a[1], a[0] + w * f, a[3]]
            elif kind == 2:    # box nested in the anchor along y, anchored below
                b = [a[0], a[3] - h * f, a[2], a[3]]
            elif kind == 3:    # the anchor shifted along x
                d = w * (1 - f) / (1 + f)
                b = [a[0] + d, a[1], a[2] + d, a[3]]
            else:              # anchor nested in a corner of the box
                g = np.sqrt(f)
                b = [a[2] - w / g, a[3] - h / g, a[2], a[3]]
            rows.append(b + [float(rng.integers(0, c))])
        ann[j] = np.asarray(rows, np.float32)
        if j == 1:
            ann[j, rng.random(m) < 0.3] = -1          # padding rows among them
    cls = rng.uniform(0.01, 0.6, (bsz, n, c)).astype(np.float32)
    reg = rng.normal(0, 1, (bsz, n, 4)).astype(np.float32)
    asg = rh.functional.assign(dev(anc), dev(ann), c)
    out = rh.functional.focal_loss_fwd_bwd(dev(cls), dev(reg), dev(anc), dev(ann))
    d_cls, d_reg = out["d_classifications"].cpu().numpy(), out["d_regressions"].cpu().numpy()
    on_threshold = 0
    for j in range(bsz):
        ref = O.assign(anc[0], ann[j])
        state = asg["state"][j].cpu().numpy()
        assert np.array_equal(state, ref["state"]), j
        assert int(out["num_pos"][j]) == ref["num_pos"] == int((state == 2).sum()), j
        assert np.array_equal((d_cls[j] == 0).all(axis=1), state == 1), j        # ignored rows
        assert np.array_equal((d_reg[j] != 0).any(axis=1), state == 2), j        # positive rows
        iou = ref["iou_max"]
        on_threshold += int((np.abs(iou - 0.4) < 1e-6).sum() + (np.abs(iou - 0.5) < 1e-6).sum())
    assert on_threshold > m // 2          # the construction does put anchors on the thresholds
    want = O.focal_loss_fwd_bwd(cls, reg, anc, ann)
    assert np.allclose(out["losses"].cpu().numpy(),
                       [want["classification_loss"][0], want["regression_loss"][0]], rtol=RTOL, atol=0)


def make_batch(seed, bsz, size, c, m, valid):
    rng = np.random.default_rng(seed)
    anc = O.anchors_for_image(size, size)
    n = anc.shape[1]
    z = rng.normal(-4.595, 2.0, (bsz, n, c))
    cls = (1.0 / (1.0 + np.exp(-z))).astype(np.float32)
    reg = rng.normal(0, 1, (bsz, n, 4)).astype(np.float32)
    ann = -np.ones((bsz, m, 5), np.float32)
    for j in range(bsz):
        v = valid[j % len(valid)]
        cx, cy = rng.uniform(0, size, v), rng.uniform(0, size, v)
        w = np.exp(rng.uniform(np.log(16), np.log(size / 2), v))
        h = np.exp(rng.uniform(np.log(16), np.log(size / 2), v))
        x1, y1 = np.clip(cx - w / 2, 0, size - 1), np.clip(cy - h / 2, 0, size - 1)
        x2, y2 = np.clip(cx + w / 2, x1 + 1, size), np.clip(cy + h / 2, y1 + 1, size)
        ann[j, :v, :4] = np.stack([x1, y1, x2, y2], 1)
        ann[j, :v, 4] = rng.integers(0, c, v)
    return cls, reg, anc, ann


@pytest.mark.parametrize("bsz,size,c,m,valid", [
    (3, 256, 80, 8, [8, 0, 3]),      # COCO-like, an empty image in the middle
    (2, 160, 601, 6, [6, 2]),        # rows of 2404 bytes: float4 accesses straddle rows
    (2, 96, 5, 4, [4, 1]),           # C not a multiple of 4
    (1, 64, 3, 4, [3]),              # C < 4: generic row kernel
    (2, 320, 80, 256, [256, 100]),   # dense GT, several LDS chunks
    (2, 160, 20, 1100, [1100, 300]), # a list long enough for the split form of the assignment
])
def test_loss_vs_oracle(rh, bsz, size, c, m, valid):
    cls, reg, anc, ann = make_batch(7 + c, bsz, size, c, m, valid)
    ref = O.focal_loss_fwd_bwd(cls, reg, anc, ann, grad_cls=1.0, grad_reg=1.0)
    out = rh.functional.focal_loss_fwd_bwd(dev(cls), dev(reg), dev(anc), dev(ann))
    torch.cuda.synchronize()
    assert np.array_equal(out["num_pos"].cpu().numpy(), ref["num_pos"])
    assert int(out["status"].max()) == 0
    assert np.allclose(out["losses"].cpu().numpy(),
                       [ref["classification_loss"][0], ref["regression_loss"][0]], rtol=RTOL, atol=0)
    pi = out["per_image"].cpu().numpy()
    assert np.allclose(pi[0], ref["cls_per_image"], rtol=RTOL, atol=0)
    assert np.allclose(pi[1], ref["reg_per_image"], rtol=RTOL, atol=0)
    grad_close(out["d_classifications"].cpu().numpy(), ref["d_classifications"])
    grad_close(out["d_regressions"].cpu().numpy(), ref["d_regressions"], atol_rel=REG_ATOL_REL, kind="d_reg")


@pytest.mark.parametrize("case", ["normal", "c80", "extreme"])
def test_loss_on_logits_golden(rh, focal_logits_golden, case):
    """from_logits=True against the reference fed torch.sigmoid(logits)."""
    d = focal_logits_golden[case]
    cls = dev(d["cls"]).requires_grad_(True)
    reg = dev(d["reg"]).requires_grad_(True)
    out = rh.FocalLoss(from_logits=True)(cls, reg, dev(d["anchors"]), dev(d["ann"]))
    total = float(d["g"][0]) * out["classification_loss"].sum() + float(d["g"][1]) * out["regression_loss"].sum()
    total.backward()
    assert np.allclose(out["classification_loss"].detach().cpu().numpy(), d["cls_loss"], rtol=RTOL, atol=0)
    assert np.allclose(out["regression_loss"].detach().cpu().numpy(), d["reg_loss"], rtol=RTOL, atol=0)
    logits_grad_close(cls.grad.cpu().numpy(), d["d_cls"], d["cls"])
    grad_close(reg.grad.cpu().numpy(), d["d_reg"], atol_rel=REG_ATOL_REL, kind="d_reg")


@pytest.mark.parametrize("bsz,size,c,m,valid", [
    (3, 256, 80, 8, [8, 0, 3]),
    (2, 96, 5, 4, [4, 1]),           # C not a multiple of 4
    (1, 64, 3, 4, [3]),              # generic row kernel
])
def test_loss_on_logits_vs_oracle_and_probability_mode(rh, bsz, size, c, m, valid):
    rng = np.random.default_rng(70 + c)
    _, reg, anc, ann = make_batch(7 + c, bsz, size, c, m, valid)
    x = rng.normal(-4.595, 2.0, (bsz, anc.shape[1], c)).astype(np.float32)
    x[0, :50, 0] = np.linspace(5.0, 9.5, 50, dtype=np.float32)      # p close to 1 and over the clamp
    ref = O.focal_loss_fwd_bwd(x, reg, anc, ann, from_logits=True)
    out = rh.functional.focal_loss_fwd_bwd(dev(x), dev(reg), dev(anc), dev(ann), from_logits=True)
    assert np.array_equal(out["num_pos"].cpu().numpy(), ref["num_pos"])
    assert np.allclose(out["losses"].cpu().numpy(),
                       [ref["classification_loss"][0], ref["regression_loss"][0]], rtol=RTOL, atol=0)
    logits_grad_close(out["d_classifications"].cpu().numpy(), ref["d_classifications"], x)
    grad_close(out["d_regressions"].cpu().numpy(), ref["d_regressions"], atol_rel=REG_ATOL_REL, kind="d_reg")
    # the same loss as the probability signature applied to torch.sigmoid of the logits
    xs = dev(x)
    prob = rh.functional.focal_loss_fwd_bwd(torch.sigmoid(xs), dev(reg), dev(anc), dev(ann))
    assert np.allclose(out["losses"].cpu().numpy(), prob["losses"].cpu().numpy(), rtol=RTOL, atol=0)
    assert torch.equal(out["d_regressions"], prob["d_regressions"])


def flat_from_levels(levels, a, last):
    """[B, A*last, H, W] per level -> [B, N, last] as the reference builds it:
    permute(0, 2, 3, 1), view(B, -1, last), cat over levels (model.py:152-154,196-202,311-313)."""
    parts = [t.transpose(0, 2, 3, 1).reshape(t.shape[0], -1, last) for t in levels]
    return np.concatenate(parts, axis=1)


def levels_from_flat(flat, shapes, a, last):
    out, at = [], 0
    for (h, w) in shapes:
        n = h * w * a
        out.append(flat[:, at:at + n].reshape(flat.shape[0], h, w, a * last).transpose(0, 3, 1, 2))
        at += n
    return out


@pytest.mark.parametrize("size,c,from_logits", [((96, 128), 8, False), ((64, 64), 5, True),
                                                ((256, 256), 80, True)])
def test_loss_on_head_layout(rh, size, c, from_logits):
    """forward_levels on [B, A*C, H, W] per level == the flat path on the permuted/concatenated
    tensors (gradients bit for bit, element by element), and == the oracle."""
    rng = np.random.default_rng(5 + c)
    h, w = size
    anc = O.anchors_for_image(h, w)
    shapes = [((h + 2 ** l - 1) // 2 ** l, (w + 2 ** l - 1) // 2 ** l) for l in (3, 4, 5, 6, 7)]
    a, bsz, m = 9, 2, 6
    assert anc.shape[1] == a * sum(x * y for x, y in shapes)
    z = [rng.normal(-4.595, 2.0, (bsz, a * c, y, x)).astype(np.float32) for (y, x) in shapes]
    cls_lv = z if from_logits else [(1.0 / (1.0 + np.exp(-t))).astype(np.float32) for t in z]
    reg_lv = [rng.normal(0, 1, (bsz, a * 4, y, x)).astype(np.float32) for (y, x) in shapes]
    ann = -np.ones((bsz, m, 5), np.float32)
    for j in range(bsz):
        v = 4 + j
        x1, y1 = rng.uniform(0, w - 40, v), rng.uniform(0, h - 40, v)
        ann[j, :v, :4] = np.stack([x1, y1, x1 + rng.uniform(16, 40, v), y1 + rng.uniform(16, 40, v)], 1)
        ann[j, :v, 4] = rng.integers(0, c, v)
    cls_flat, reg_flat = flat_from_levels(cls_lv, a, c), flat_from_levels(reg_lv, a, 4)

    tc = [dev(t).requires_grad_(True) for t in cls_lv]
    tr = [dev(t).requires_grad_(True) for t in reg_lv]
    mod = rh.FocalLoss(from_logits=from_logits)
    out = mod.forward_levels(tc, tr, dev(anc), dev(ann))
    (out["classification_loss"] + out["regression_loss"]).sum().backward()

    flat = rh.functional.focal_loss_fwd_bwd(dev(cls_flat), dev(reg_flat), dev(anc), dev(ann),
                                            from_logits=from_logits)
    assert np.array_equal(mod.last["num_pos"].cpu().numpy(), flat["num_pos"].cpu().numpy())
    assert int(flat["num_pos"].sum()) > 0
    assert np.allclose(out["classification_loss"].item(), flat["losses"][0].item(), rtol=1e-6)
    assert np.allclose(out["regression_loss"].item(), flat["losses"][1].item(), rtol=1e-6)
    want_c = levels_from_flat(flat["d_classifications"].cpu().numpy(), shapes, a, c)
    want_r = levels_from_flat(flat["d_regressions"].cpu().numpy(), shapes, a, 4)
    for l in range(len(shapes)):
        assert np.array_equal(tc[l].grad.cpu().numpy(), want_c[l]), l
        assert np.array_equal(tr[l].grad.cpu().numpy(), want_r[l]), l
    ref = O.focal_loss_fwd_bwd(cls_flat, reg_flat, anc, ann, from_logits=from_logits)
    assert np.allclose(out["classification_loss"].item(), ref["classification_loss"][0], rtol=RTOL)
    assert np.allclose(out["regression_loss"].item(), ref["regression_loss"][0], rtol=RTOL)


def test_loss_sum_uses_the_stored_total(rh, focal_golden):
    """train.py:185-190: sum(loss_dict.values()).backward().  The sum is the total the finalize
    kernel stored (no add launches), bit-equal to cls + reg, and back-propagates like it."""
    d = focal_golden["normal"]
    args = (dev(d["anchors"]), dev(d["ann"]))
    grads = []
    for how in ("sum", "plain_add", "scaled", "separate"):
        cls = dev(d["cls"]).requires_grad_(True)
        reg = dev(d["reg"]).requires_grad_(True)
        out = rh.FocalLoss()(cls, reg, *args)
        c, r = out["classification_loss"], out["regression_loss"]
        assert c.shape == (1,) and r.shape == (1,) and c.grad_fn is not None
        if how == "sum":
            loss = sum(l for l in out.values())
            assert isinstance(loss, torch.Tensor) and loss.shape == (1,)
            assert loss.item() == (c.detach().clone() + r.detach().clone()).item()
            assert loss.data_ptr() == c.data_ptr() + 16                  # the stored total, no new tensor
            again = sum(out.values())                                    # handed out once: a new tensor
            assert again.data_ptr() != loss.data_ptr() and again.item() == loss.item()
        elif how == "plain_add":
            loss = c + r
        elif how == "scaled":
            loss = (c * 1.0 + r).mean()                                  # ordinary autograd path
        else:
            loss = None
            torch.autograd.backward([c, r], [torch.ones_like(c), torch.ones_like(r)])
        if loss is not None:
            loss.backward()
        grads.append((cls.grad.clone(), reg.grad.clone()))
        assert isinstance(float(c), float) and isinstance(c.item(), float)
    for gc, gr in grads[1:]:
        assert torch.equal(gc, grads[0][0]) and torch.equal(gr, grads[0][1])
    grad_close(grads[0][0].cpu().numpy(), d["d_cls"] / d["g"][0])
    # using the total AND one part: the part's gradient counts twice
    cls = dev(d["cls"]).requires_grad_(True)
    reg = dev(d["reg"]).requires_grad_(True)
    out = rh.FocalLoss()(cls, reg, *args)
    (sum(out.values()) + out["classification_loss"]).sum().backward()
    assert torch.allclose(cls.grad, 2 * grads[0][0], rtol=1e-6, atol=0)
    assert torch.equal(reg.grad, grads[0][1])
    # gradient accumulation, train.py:185-190 word for word: the sum is divided IN PLACE
    for expected in (0.25, 1.0):
        cls = dev(d["cls"]).requires_grad_(True)
        reg = dev(d["reg"]).requires_grad_(True)
        loss_dict = rh.FocalLoss(expected_grad=expected)(cls, reg, *args)
        loss = sum(loss for loss in loss_dict.values())
        monitor_loss = loss.clone().detach()
        loss /= 4
        assert np.isclose(float(loss.item()) * 4, float(monitor_loss.item()), rtol=1e-6)
        loss.backward()
        assert torch.allclose(cls.grad, grads[0][0] / 4, rtol=1e-6, atol=0)
        assert torch.allclose(reg.grad, grads[0][1] / 4, rtol=1e-6, atol=0)
        # the dict still holds the undivided losses (train.py:193-197 logs them)
        assert np.isclose(sum(float(v) for v in loss_dict.values()), float(monitor_loss), rtol=1e-6)
        assert np.isclose(float(sum(loss_dict.values())), float(monitor_loss), rtol=1e-6)
    # in-place operations on the individual losses work as on any tensor
    cls = dev(d["cls"]).requires_grad_(True)
    reg = dev(d["reg"]).requires_grad_(True)
    out = rh.FocalLoss()(cls, reg, *args)
    c = out["classification_loss"]
    c *= 2
    (c + out["regression_loss"]).backward()
    assert torch.allclose(cls.grad, 2 * grads[0][0], rtol=1e-6, atol=0)
    assert torch.equal(reg.grad, grads[0][1])


@pytest.mark.parametrize("bsz,size,c,m,valid,norm", [
    (3, 256, 80, 8, [8, 0, 3], "per_image"),
    (2, 160, 601, 6, [6, 2], "per_image"),       # tiles cut into parts: several records per tile
    (70, 64, 8, 3, [3, 1, 0, 2], "per_image"),   # more images than a wave has lanes; N = 774, tiles straddle images
    (5, 96, 20, 4, [4, 1], "global"),
    (150, 64, 5, 2, [2, 1, 0], "per_image"),     # more images than finishers may ride in a launch: own launch
])
def test_final_reduction_inside_the_stream_launch(rh, bsz, size, c, m, valid, norm):
    """A call that runs the stream and the final reduction together reduces each image inside the
    stream launch (finishing workgroups at the end of the grid, per-image tickets); run apart, the
    stages use the finalize kernel.  Same code, same order: the same bits -- also on the second
    call on a workspace (tickets back at zero) and against the oracle."""
    F, L = rh.functional, rh._lib
    cls, reg, anc, ann = make_batch(90 + c, bsz, size, c, m, valid)
    d = [dev(x) for x in (cls, reg, anc, ann)]
    apart = F.focal_loss_fwd_bwd(*d, stages=L.STAGE_ASSIGN, normalizer=norm)
    F.focal_loss_fwd_bwd(*d, stages=L.STAGE_STREAM, out=apart, normalizer=norm)
    F.focal_loss_fwd_bwd(*d, stages=L.STAGE_FINALIZE, out=apart, normalizer=norm)
    for rep in range(3):
        fused = F.focal_loss_fwd_bwd(*d, normalizer=norm)
        torch.cuda.synchronize()
        for k in ("losses", "exchange", "total", "per_image", "num_pos", "d_classifications", "d_regressions"):
            assert torch.equal(fused[k], apart[k]), (k, rep)
    assert torch.isfinite(fused["losses"]).all()
    if norm == "per_image":
        ref = O.focal_loss_fwd_bwd(cls, reg, anc, ann, grad_cls=1.0, grad_reg=1.0)
        assert np.allclose(fused["losses"].cpu().numpy(),
                           [ref["classification_loss"][0], ref["regression_loss"][0]], rtol=RTOL, atol=0)
    # stream + finalize on a workspace whose assignment ran in an earlier call
    late = F.focal_loss_fwd_bwd(*d, stages=L.STAGE_ASSIGN, normalizer=norm)
    F.focal_loss_fwd_bwd(*d, stages=L.STAGE_STREAM | L.STAGE_FINALIZE, out=late, normalizer=norm)
    assert torch.equal(late["losses"], apart["losses"]) and torch.equal(late["per_image"], apart["per_image"])


@pytest.mark.parametrize("fill", [0xFF, 0x01, "random", "records"])
def test_loss_on_a_workspace_full_of_old_bytes(rh, fill):
    """The workspace is caller-owned memory and needs no initialisation: whatever bytes it holds
    before the first call (all ones would read as a valid odd tag; random bytes stand for another
    user's data; 'records' are the tagged records a call of ANOTHER shape left behind, which land
    on other offsets of this shape's layout), the assignment stage clears the tags the reduction
    inside the launch polls and the call gives the bits of a zero-filled workspace."""
    F = rh.functional
    cls, reg, anc, ann = make_batch(77, 3, 192, 20, 5, [5, 0, 2])
    d = [dev(x) for x in (cls, reg, anc, ann)]
    cuda = torch.device("cuda:0")
    zeroed = F.focal_workspace(3, anc.shape[1], 20, 5, cuda).zero_()
    clean = F.focal_loss_fwd_bwd(*d, ws=zeroed)
    ws = F.focal_workspace(3, anc.shape[1], 20, 5, cuda)
    if fill == "random":
        ws.copy_(torch.randint(0, 256, ws.shape, dtype=torch.uint8, generator=torch.Generator().manual_seed(5)).cuda())
    elif fill == "records":
        big = F.focal_workspace(4, anc.shape[1], 33, 9, cuda)
        assert big.numel() >= ws.numel()
        o_cls, o_reg, o_anc, o_ann = make_batch(78, 4, 192, 33, 9, [9, 1, 0, 4])
        # the stream stage alone: its records stay in place (nothing consumes them)
        bufs = F.focal_loss_fwd_bwd(dev(o_cls), dev(o_reg), dev(o_anc), dev(o_ann), ws=big)
        for _ in range(3):
            F.focal_loss_fwd_bwd(dev(o_cls), dev(o_reg), dev(o_anc), dev(o_ann), ws=big, out=bufs,
                                 stages=rh._lib.STAGE_ASSIGN | rh._lib.STAGE_STREAM)
        ws = big
    else:
        ws.fill_(fill)
    for rep in range(2):
        got = F.focal_loss_fwd_bwd(*d, ws=ws)
        torch.cuda.synchronize()
        for k in ("losses", "per_image", "num_pos", "d_classifications", "d_regressions"):
            assert torch.equal(got[k], clean[k]), (k, rep)


def test_loss_deterministic_and_unaligned(rh):
    cls, reg, anc, ann = make_batch(3, 2, 128, 80, 5, [5, 2])
    a = rh.functional.focal_loss_fwd_bwd(dev(cls), dev(reg), dev(anc), dev(ann))
    la, ga = a["losses"].cpu().numpy().copy(), a["d_classifications"].cpu().numpy().copy()
    b = rh.functional.focal_loss_fwd_bwd(dev(cls), dev(reg), dev(anc), dev(ann))
    assert np.array_equal(la, b["losses"].cpu().numpy())          # bitwise reproducible
    assert np.array_equal(ga, b["d_classifications"].cpu().numpy())
    # a misaligned view goes through the wrapper's copy and gives the same bits
    pad = torch.zeros(cls.size + 1, device="cuda")
    pad[1:] = dev(cls).reshape(-1)
    view = pad[1:].view(cls.shape)
    assert view.data_ptr() % 16 != 0
    c = rh.functional.focal_loss_fwd_bwd(view, dev(reg), dev(anc), dev(ann))
    assert np.array_equal(la, c["losses"].cpu().numpy())


def test_loss_batch_divisor_and_expected_grad(rh):
    cls, reg, anc, ann = make_batch(5, 4, 96, 7, 4, [4, 2, 0, 3])
    ref = O.focal_loss_fwd_bwd(cls, reg, anc, ann, grad_cls=0.25, grad_reg=0.25)
    # two "ranks" with two images each, batch_divisor = 4: losses add up, gradients concatenate
    parts = []
    for lo in (0, 2):
        t = dev(cls[lo:lo + 2]).requires_grad_(True)
        r = dev(reg[lo:lo + 2]).requires_grad_(True)
        mod = rh.FocalLoss(batch_divisor=4, expected_grad=0.25)
        o = mod(t, r, dev(anc), dev(ann[lo:lo + 2]))
        ((o["classification_loss"] + o["regression_loss"]) * 0.25).sum().backward()
        parts.append((o, t.grad.cpu().numpy(), r.grad.cpu().numpy()))
    cl = sum(float(p[0]["classification_loss"]) for p in parts)
    rl = sum(float(p[0]["regression_loss"]) for p in parts)
    assert np.isclose(cl, ref["classification_loss"][0], rtol=RTOL)
    assert np.isclose(rl, ref["regression_loss"][0], rtol=RTOL)
    grad_close(np.concatenate([p[1] for p in parts]), ref["d_classifications"])
    grad_close(np.concatenate([p[2] for p in parts]), ref["d_regressions"], atol_rel=REG_ATOL_REL, kind="d_reg")


def test_plain_backward_launches_nothing_more(rh, focal_golden, monkeypatch):
    """loss.backward() on the summed loss (train.py:185, 190) with the default expected gradient:
    the stored gradients are final, so the backward pass runs no scaling launch; every other way of
    reaching the node still scales (or zeroes) them."""
    d = focal_golden["normal"]
    args = (dev(d["anchors"]), dev(d["ann"]))
    calls = []
    real = rh.functional.apply_upstream
    monkeypatch.setattr(rh.functional, "apply_upstream", lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def run(how, **kw):
        cls = dev(d["cls"]).requires_grad_(True)
        reg = dev(d["reg"]).requires_grad_(True)
        out = rh.FocalLoss(**kw)(cls, reg, *args)
        del calls[:]
        how(out)
        return cls.grad, reg.grad, len(calls)

    gc, gr, n = run(lambda o: sum(o.values()).backward())
    assert n == 0
    grad_close(gc.cpu().numpy(), d["d_cls"] / d["g"][0])
    grad_close(gr.cpu().numpy(), d["d_reg"] / d["g"][1], atol_rel=REG_ATOL_REL, kind="d_reg")
    gc2, gr2, n = run(lambda o: (o["classification_loss"] + o["regression_loss"]).backward())
    assert n == 0 and torch.equal(gc, gc2) and torch.equal(gr, gr2)
    # an explicit gradient, a scaled loss, one loss alone, another expected gradient: all scale
    gc2, gr2, n = run(lambda o: sum(o.values()).backward(torch.full((1,), 2.0, device="cuda")))
    assert n == 1 and torch.allclose(gc2, 2 * gc, rtol=1e-6, atol=0)
    gc2, gr2, n = run(lambda o: (sum(o.values()) * 3).backward())
    assert n == 1 and torch.allclose(gr2, 3 * gr, rtol=1e-6, atol=0)
    gc2, gr2, n = run(lambda o: o["classification_loss"].backward())
    assert n == 1 and torch.equal(gc2, gc) and int((gr2 != 0).sum()) == 0
    gc2, gr2, n = run(lambda o: sum(o.values()).backward(), expected_grad=0.5)
    assert n == 1 and torch.allclose(gc2, gc, rtol=1e-6, atol=0)
    # retain_graph is passed through; the node itself still serves one backward per forward
    cls = dev(d["cls"]).requires_grad_(True)
    out = rh.FocalLoss()(cls, dev(d["reg"]), *args)
    loss = sum(out.values())
    loss.backward(retain_graph=True)
    with pytest.raises(rh.RetinaHeadError):
        loss.backward()


@pytest.mark.parametrize("bsz,size,c,m,valid", [
    (3, 256, 80, 8, [8, 0, 3]),      # tile kernel, an empty image
    (2, 96, 5, 4, [4, 1]),           # C not a multiple of 4
    (1, 64, 3, 4, [3]),              # generic row kernel
])
def test_global_normaliser_vs_oracle(rh, bsz, size, c, m, valid):
    """normalizer='global' (one positive count for the batch; not in the reference, so pinned to the
    oracle's restatement, which the CPU suite ties to the reference-checked per-image mode)."""
    cls, reg, anc, ann = make_batch(11 + c, bsz, size, c, m, valid)
    ref = O.focal_loss_fwd_bwd(cls, reg, anc, ann, normalizer="global")
    out = rh.functional.focal_loss_fwd_bwd(dev(cls), dev(reg), dev(anc), dev(ann), normalizer="global")
    assert np.array_equal(out["num_pos"].cpu().numpy(), ref["num_pos"])
    assert np.allclose(out["losses"].cpu().numpy(),
                       [ref["classification_loss"][0], ref["regression_loss"][0]], rtol=RTOL, atol=0)
    assert out["total"].item() == (out["losses"][0] + out["losses"][1]).item()
    # the vector data-parallel ranks sum: both losses, the call's positives, its images
    assert out["exchange"].tolist() == [out["losses"][0].item(), out["losses"][1].item(),
                                        float(ref["num_pos"].sum()), float(bsz)]
    pi = out["per_image"].cpu().numpy()
    assert np.allclose(pi[0], ref["cls_per_image"], rtol=RTOL, atol=0)
    assert np.allclose(pi[1], ref["reg_per_image"], rtol=RTOL, atol=0)
    grad_close(out["d_classifications"].cpu().numpy(), ref["d_classifications"])
    grad_close(out["d_regressions"].cpu().numpy(), ref["d_regressions"], atol_rel=REG_ATOL_REL, kind="d_reg")
    # a count formed elsewhere (all ranks of a data-parallel job): stages run apart, as the module does
    from retinahead import _lib
    total = torch.tensor([2.0 * float(ref["num_pos"].sum()) + 3.0], device="cuda")
    ref2 = O.focal_loss_fwd_bwd(cls, reg, anc, ann, normalizer="global", num_pos_total=int(total.item()))
    args = (dev(cls), dev(reg), dev(anc), dev(ann))
    st = rh.functional.focal_loss_fwd_bwd(*args, normalizer="global", stages=_lib.STAGE_ASSIGN)
    assert np.array_equal(st["num_pos"].cpu().numpy(), ref["num_pos"])      # counted in the assign stage
    st = rh.functional.focal_loss_fwd_bwd(*args, normalizer="global", num_pos_total=total, out=st,
                                          stages=_lib.STAGE_STREAM | _lib.STAGE_FINALIZE)
    assert np.allclose(st["losses"].cpu().numpy(),
                       [ref2["classification_loss"][0], ref2["regression_loss"][0]], rtol=RTOL, atol=0)
    grad_close(st["d_classifications"].cpu().numpy(), ref2["d_classifications"])
    grad_close(st["d_regressions"].cpu().numpy(), ref2["d_regressions"], atol_rel=REG_ATOL_REL, kind="d_reg")
    with pytest.raises(rh.RetinaHeadError):
        rh.functional.focal_loss_fwd_bwd(*args, num_pos_total=total)          # per-image mode takes none


def test_global_normaliser_module_and_head_layout(rh):
    """The module with the data-parallel hook (world of one: the hook only sums the shard's counts)
    equals the unhooked call, on the flat and on the head layout; equal positive counts in every
    image make the mode coincide with the reference's per-image normaliser."""
    from retinahead import dist as rdist
    rng = np.random.default_rng(77)
    h = w = 128
    c, a, bsz = 6, 9, 3
    anc = O.anchors_for_image(h, w)
    shapes = [((h + 2 ** l - 1) // 2 ** l, (w + 2 ** l - 1) // 2 ** l) for l in (3, 4, 5, 6, 7)]
    cls_lv = [rng.uniform(0.001, 0.4, (bsz, a * c, y, x)).astype(np.float32) for (y, x) in shapes]
    reg_lv = [rng.normal(0, 1, (bsz, a * 4, y, x)).astype(np.float32) for (y, x) in shapes]
    _, _, _, ann = make_batch(78, bsz, h, c, 5, [5, 1, 3])
    cls_flat, reg_flat = flat_from_levels(cls_lv, a, c), flat_from_levels(reg_lv, a, 4)
    ref = O.focal_loss_fwd_bwd(cls_flat, reg_flat, anc, ann, normalizer="global")
    assert len(set(ref["num_pos"].tolist())) > 1

    results = []
    for hooked in (False, True):
        mod = rh.FocalLoss(normalizer="global")
        if hooked:
            rdist.DataParallelFocalLoss(mod)          # installs reduce_total
            assert mod.cfg["reduce_total"] is not None
        tc, tr = dev(cls_flat).requires_grad_(True), dev(reg_flat).requires_grad_(True)
        out = mod(tc, tr, dev(anc), dev(ann))
        sum(out.values()).backward()
        lc = [dev(t).requires_grad_(True) for t in cls_lv]
        lr = [dev(t).requires_grad_(True) for t in reg_lv]
        out_lv = mod.forward_levels(lc, lr, dev(anc), dev(ann))
        sum(out_lv.values()).backward()
        results.append((out, tc.grad, tr.grad, out_lv, lc, lr))
        for o in (out, out_lv):
            assert np.isclose(o["classification_loss"].item(), ref["classification_loss"][0], rtol=RTOL)
            assert np.isclose(o["regression_loss"].item(), ref["regression_loss"][0], rtol=RTOL)
        grad_close(tc.grad.cpu().numpy(), ref["d_classifications"])
        grad_close(tr.grad.cpu().numpy(), ref["d_regressions"], atol_rel=REG_ATOL_REL, kind="d_reg")
        want_c = levels_from_flat(tc.grad.cpu().numpy(), shapes, a, c)
        want_r = levels_from_flat(tr.grad.cpu().numpy(), shapes, a, 4)
        for l in range(len(shapes)):
            assert np.array_equal(lc[l].grad.cpu().numpy(), want_c[l]), l
            assert np.array_equal(lr[l].grad.cpu().numpy(), want_r[l]), l
    assert torch.equal(results[0][1], results[1][1]) and torch.equal(results[0][2], results[1][2])
    assert results[0][0]["classification_loss"].item() == results[1][0]["classification_loss"].item()

    # every image with the same annotations -> same positive counts -> both normalisers agree
    same = np.repeat(ann[:1], bsz, axis=0)
    args = (dev(cls_flat), dev(reg_flat), dev(anc), dev(same))
    per = rh.functional.focal_loss_fwd_bwd(*args)
    glo = rh.functional.focal_loss_fwd_bwd(*args, normalizer="global")
    assert int(per["num_pos"][0]) > 0 and len(set(per["num_pos"].tolist())) == 1
    assert torch.allclose(per["losses"], glo["losses"], rtol=1e-6, atol=0)
    assert torch.allclose(per["d_classifications"], glo["d_classifications"], rtol=1e-6, atol=0)
    assert torch.allclose(per["d_regressions"], glo["d_regressions"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("size,c", [(64, 8200), (8, 6)])
def test_shapes_outside_the_tile_kernels(rh, size, c):
    """More classes than the tile kernels index (C >= 8192) and fewer anchors than one tile
    (N = 45 for an 8 x 8 image): the row kernels take over, same results."""
    rng = np.random.default_rng(size + c)
    anc = O.anchors_for_image(size, size)
    n = anc.shape[1]
    assert (n < 64) == (size == 8)
    bsz = 2
    cls = rng.uniform(0.0005, 0.3, (bsz, n, c)).astype(np.float32)
    reg = rng.normal(0, 1, (bsz, n, 4)).astype(np.float32)
    ann = -np.ones((bsz, 2, 5), np.float32)
    ann[0, 0] = [*anc[0, 3], c - 1]             # ground truth on top of an anchor: positives for sure
    ann[1, 1] = [*anc[0, n // 2], 0]
    ref = O.focal_loss_fwd_bwd(cls, reg, anc, ann)
    assert ref["num_pos"].sum() > 0
    out = rh.functional.focal_loss_fwd_bwd(dev(cls), dev(reg), dev(anc), dev(ann))
    assert np.array_equal(out["num_pos"].cpu().numpy(), ref["num_pos"])
    assert np.allclose(out["losses"].cpu().numpy(),
                       [ref["classification_loss"][0], ref["regression_loss"][0]], rtol=RTOL, atol=0)
    grad_close(out["d_classifications"].cpu().numpy(), ref["d_classifications"])
    grad_close(out["d_regressions"].cpu().numpy(), ref["d_regressions"], atol_rel=REG_ATOL_REL, kind="d_reg")
    # post-processing on the same shapes
    hot = rng.random((bsz, n)) < 0.3
    cls[hot, rng.integers(0, c, int(hot.sum()))] = rng.uniform(0.06, 0.99, int(hot.sum())).astype(np.float32)
    check_post(rh, cls, (0.3 * reg).astype(np.float32), anc, size)


def test_other_dtypes_and_layouts_go_through_float32(rh):
    """Half / double / non-contiguous inputs are cast to contiguous float32 by the Python layer;
    gradients come back in the inputs' dtypes and layouts."""
    cls, reg, anc, ann = make_batch(41, 2, 96, 6, 4, [4, 2])
    ref = O.focal_loss_fwd_bwd(cls.astype(np.float16).astype(np.float32), reg, anc, ann)
    c16 = dev(cls).half().requires_grad_(True)
    r64 = dev(reg).double().requires_grad_(True)
    out = rh.FocalLoss()(c16, r64, dev(anc), dev(ann).double())
    sum(out.values()).backward()
    assert c16.grad.dtype == torch.float16 and r64.grad.dtype == torch.float64
    assert np.isclose(out["classification_loss"].item(), ref["classification_loss"][0], rtol=RTOL)
    assert np.allclose(r64.grad.cpu().numpy(), ref["d_regressions"], rtol=1e-5,
                       atol=REG_ATOL_REL * np.abs(ref["d_regressions"]).max())
    assert np.allclose(c16.grad.float().cpu().numpy(), ref["d_classifications"], rtol=2e-3, atol=1e-7)
    # a transposed (non-contiguous) view of the same data
    base = dev(np.ascontiguousarray(cls.transpose(0, 2, 1))).requires_grad_(True)     # [B, C, N]
    view = base.transpose(1, 2)                                                       # [B, N, C], strided
    assert not view.is_contiguous()
    out2 = rh.FocalLoss()(view, dev(reg), dev(anc), dev(ann))
    sum(out2.values()).backward()
    ref32 = O.focal_loss_fwd_bwd(cls, reg, anc, ann)
    assert base.grad.shape == base.shape
    grad_close(base.grad.transpose(1, 2).cpu().numpy(), ref32["d_classifications"])


def test_label_out_of_range_flag(rh):
    cls, reg, anc, ann = make_batch(9, 1, 64, 4, 3, [3])
    ann[0, 1, 4] = 4.0                                   # == C: IndexError in the reference
    with pytest.raises(IndexError):
        rh.FocalLoss(check_labels=True)(dev(cls), dev(reg), dev(anc), dev(ann))


@pytest.mark.parametrize("value", [float("nan"), float("inf")])
def test_non_finite_box_flag(rh, value):
    """A valid annotation row with a NaN or infinite coordinate sets its own status bit (the
    reference lets the NaN IoU decide, losses.py:61; here the pair is skipped, so the image's
    result is flagged instead of trusted); padding rows may hold anything."""
    from retinahead import _lib
    cls, reg, anc, ann = make_batch(9, 2, 64, 4, 3, [3, 2])
    ann[1, 2, :4] = value                                # image 1, row 2 is padding (label -1)
    out = rh.functional.focal_loss_fwd_bwd(dev(cls), dev(reg), dev(anc), dev(ann))
    assert out["status"].tolist() == [0, 0]
    ann[0, 1, 2] = value
    out = rh.functional.focal_loss_fwd_bwd(dev(cls), dev(reg), dev(anc), dev(ann))
    assert out["status"].tolist() == [_lib.RH_STATUS_NONFINITE_BOX, 0]
    with pytest.raises(ValueError):
        rh.FocalLoss(check_labels=True)(dev(cls), dev(reg), dev(anc), dev(ann))


def test_cpu_tensor_rejected(rh):
    cls, reg, anc, ann = make_batch(9, 1, 64, 4, 3, [3])
    with pytest.raises(rh.RetinaHeadError):
        rh.FocalLoss()(torch.from_numpy(cls), torch.from_numpy(reg), torch.from_numpy(anc),
                       torch.from_numpy(ann))


# ------------------------------------------------------------------ generic focal
def test_generic_focal_golden(rh):
    g = load_golden("generic_focal.npz")
    t = dev(g["targets"])
    for logits in (True, False):
        for reduce in (True, False):
            key = f"{'logits' if logits else 'probs'}_{'mean' if reduce else 'none'}"
            x = dev(g[key + ".inp"]).requires_grad_(True)
            out = rh.GenericFocalLoss(logits=logits, reduce=reduce)(x, t)
            out.sum().backward()
            assert np.allclose(out.detach().cpu().numpy(), g[key + ".loss"], rtol=2e-5, atol=1e-7), key
            assert np.allclose(x.grad.cpu().numpy(), g[key + ".grad"], rtol=2e-4,
                               atol=1e-6 if not reduce else 1e-8), key


def test_collate_annotations(rh):
    rng = np.random.default_rng(4)
    counts = [3, 0, 7, 1]
    tg = [(rng.uniform(0, 100, (k, 4)).astype(np.float32), rng.integers(0, 9, k).astype(np.float32))
          for k in counts]
    got = rh.collate_annotations([{"boxes": dev(b), "labels": dev(l).reshape(-1, 1)} for b, l in tg])
    assert got.shape == (4, 7, 5) and np.array_equal(got.cpu().numpy(), O.collate_annotations(tg))
    empty = [(np.zeros((0, 4), np.float32), np.zeros(0, np.float32))] * 2
    got = rh.collate_annotations([{"boxes": dev(b), "labels": dev(l)} for b, l in empty])
    assert got.shape == (2, 1, 5) and np.array_equal(got.cpu().numpy(), O.collate_annotations(empty))
    # int64 labels, as datasets produce them, and the result feeds the loss as is
    tg64 = [{"boxes": dev(b), "labels": dev(l.astype(np.int64))} for b, l in tg]
    assert np.array_equal(rh.collate_annotations(tg64).cpu().numpy(), O.collate_annotations(tg))


@pytest.mark.parametrize("xywh", [False, True])
def test_collate_with_the_datasets_annotation_filter(rh, xywh):
    """min_size / box_format: rows under one pixel in width or height dropped, [x, y, w, h] turned
    into corners, order kept (csv_dataset.py:129-136, coco_dataset.py:97-108), for lists longer than
    one 64-row pass, an image that loses every row and an empty image."""
    rng = np.random.default_rng(8 + xywh)
    counts = [5, 0, 150, 3, 64]
    tg = []
    for k in counts:
        xy = rng.uniform(0, 500, (k, 2))
        wh = np.where(rng.random((k, 2)) < 0.3, rng.uniform(0, 1, (k, 2)), rng.uniform(1, 200, (k, 2)))
        if k == 3:
            wh[:] = 0.5                                  # nothing survives
        if k == 150:
            wh[7] = [1.0, 1.0]                           # exactly one pixel stays
            wh[8] = [np.nan, 5.0]                        # nan < 1 is false: stays
        b = np.concatenate([xy, wh if xywh else xy + wh], 1).astype(np.float32)
        tg.append((b, rng.integers(0, 9, k).astype(np.float32)))
    want = O.collate_annotations(tg, min_size=1.0, xywh=xywh)
    got, kept = rh.collate_annotations([{"boxes": dev(b), "labels": dev(l)} for b, l in tg],
                                       min_size=1.0, box_format="xywh" if xywh else "xyxy",
                                       return_counts=True)
    assert got.shape == want.shape
    assert np.array_equal(got.cpu().numpy(), want, equal_nan=True)
    want_kept = [int((want[i, :, 4] != -1).sum()) for i in range(len(tg))]
    assert kept.tolist() == want_kept and want_kept[1] == 0 and want_kept[3] == 0
    assert 0 < want_kept[2] < 150
    if xywh:            # conversion alone, nothing dropped
        got = rh.collate_annotations([{"boxes": dev(b), "labels": dev(l)} for b, l in tg],
                                     box_format="xywh")
        assert np.array_equal(got.cpu().numpy(), O.collate_annotations(tg, xywh=True), equal_nan=True)


def test_detection_matching_and_ap_golden(rh):
    """Device matching + host AP against csv_eval.evaluate's results (tests/golden/evaluate.npz)
    and, flag by flag, against the oracle."""
    g = load_golden("evaluate.npz")
    n_img = len(g["det_counts"])
    ann = dev(g["annotations"])
    for tag, thr in (("t50", 0.5), ("t75", 0.75)):
        ev = rh.DetectionEvaluator(5, iou_threshold=thr, max_detections=25)
        results = []
        for i in range(n_img):
            d = int(g["det_counts"][i])
            results.append({"boxes": dev(g["det_boxes"][i, :d]), "labels": dev(g["det_labels"][i, :d]),
                            "scores": dev(g["det_scores"][i, :d])})
        ev.update(results[:3], ann[:3], scales=g["scales"][:3])          # two batches
        ev.update(results[3:], ann[3:], scales=g["scales"][3:])
        aps = ev.average_precisions()
        assert np.allclose([aps[c][0] for c in range(5)], g[tag + ".ap"], rtol=1e-12, atol=0)
        assert np.array_equal([aps[c][1] for c in range(5)], g[tag + ".num_annotations"])
    # flags against the oracle on random, dense cases (many detections per annotation, ties)
    rng = np.random.default_rng(8)
    bsz, d, m = 5, 300, 70
    a = -np.ones((bsz, m, 5), np.float32)
    for j in range(bsz):
        k = int(rng.integers(0, m + 1)) if j else m
        xy = rng.integers(0, 200, (k, 2))
        a[j, :k, :2], a[j, :k, 2:4], a[j, :k, 4] = xy, xy + rng.integers(10, 60, (k, 2)), rng.integers(0, 3, k)
    src = a[np.arange(bsz)[:, None], rng.integers(0, m, (bsz, d))]        # copies of annotation rows
    boxes = (src[:, :, :4] + rng.integers(-6, 7, (bsz, d, 4))).astype(np.float32)
    boxes[:, ::7] = src[:, ::7, :4]                                        # exact duplicates: IoU ties
    labels = rng.integers(0, 3, (bsz, d))
    counts = np.array([d, 0, 17, d, 250], np.int32)
    tp = rh.functional.match_detections(dev(boxes), dev(labels), dev(counts), dev(a), 0.5).cpu().numpy()
    for j in range(bsz):
        k = int(counts[j])
        assert np.array_equal(tp[j, :k], O.match_detections(boxes[j, :k], labels[j, :k], a[j], 0.5)), j
        assert not tp[j, k:].any()


# ------------------------------------------------------------------ decode / clip
def test_decode_clip_golden(rh):
    g = load_golden("decode.npz")
    dec = rh.BBoxTransform()(dev(g["anchors"]), dev(g["deltas"]))
    scale = np.max(np.abs(g["decoded"]), axis=2, keepdims=True)
    assert (np.abs(dec.cpu().numpy().astype(np.float64) - g["decoded"]) <= 1e-6 * scale).all()
    h, w = int(g["hw"][0]), int(g["hw"][1])
    boxes = dev(g["decoded"])
    ret = rh.ClipBoxes()(boxes, torch.zeros(3, 3, h, w, device="cuda"))
    assert ret.data_ptr() == boxes.data_ptr()            # in place, like the reference
    assert np.array_equal(boxes.cpu().numpy(), g["clipped"])


# ------------------------------------------------------------------ NMS
def rand_dets(seed, k, size=200.0, cluster=True):
    rng = np.random.default_rng(seed)
    if cluster:
        centers = rng.uniform(0, size, (max(k // 8, 1), 2))
        c = centers[rng.integers(0, centers.shape[0], k)] + rng.normal(0, 4, (k, 2))
    else:
        c = rng.uniform(0, size, (k, 2))
    wh = rng.uniform(8, 40, (k, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    scores = rng.random(k).astype(np.float32)
    return boxes, scores


@pytest.mark.parametrize("k", [1, 2, 63, 64, 65, 500, 2000, 5000, 8192, 8193])
def test_nms_vs_oracle(rh, k):
    boxes, scores = rand_dets(k, k)
    scores[: k // 3] = scores[k // 3: 2 * (k // 3)]      # score ties: lower index first
    keep = rh.nms(dev(boxes), dev(scores), 0.5).cpu().numpy()
    assert keep.dtype == np.int64
    assert np.array_equal(keep, O.nms(boxes, scores, 0.5))
    keep2 = rh.nms(dev(boxes), dev(scores[:, None]), 0.3).cpu().numpy()   # [K,1] scores (model.py:336)
    assert np.array_equal(keep2, O.nms(boxes, scores, 0.3))


@pytest.mark.parametrize("thr", [0.5, 0.55, 0.75])
def test_nms_pairs_on_the_threshold_with_centres_on_the_edges(rh, thr):
    """From a threshold of one half on the pair search also requires each box's centre inside the
    other box.  The cases where that rule is tightest: a box nested in a corner or along an edge of
    another so that their IoU is the threshold give or take a few ulp -- the larger box's centre
    then lies on the smaller one's edge -- and shifted copies whose IoU is the threshold.  Pairs
    are placed apart from one another; which of them suppress is decided by the float32
    division, exactly as the oracle (torchvision's arithmetic) decides it."""
    rng = np.random.default_rng(int(thr * 100))
    boxes, cell = [], 250.0
    per_row = 40
    n_pairs = 2000
    for i in range(n_pairs):
        ox, oy = (i % per_row) * cell, (i // per_row) * cell
        w, h = rng.uniform(8, 100), rng.uniform(8, 100)
        x, y = ox + rng.uniform(0, 100), oy + rng.uniform(0, 100)
        f = thr * (1.0 + rng.choice([-3, -1, 0, 1, 3]) * 6e-8 * rng.integers(0, 3))
        kind = i % 6
        a = [x, y, x + w, y + h]
        if kind == 0:        # nested, sharing the left edge: IoU = f
            b = [x, y, x + w * f, y + h]
        elif kind == 1:      # nested, sharing the right edge
            b = [x + w * (1 - f), y, x + w, y + h]
        elif kind == 2:      # nested, sharing the top edge
            b = [x, y, x + w, y + h * f]
        elif kind == 3:      # nested in a corner: sqrt(f) of each side
            g = np.sqrt(f)
            b = [x + w * (1 - g), y + h * (1 - g), x + w, y + h]
        elif kind == 4:      # a copy shifted along x: IoU = (w - d) / (w + d)
            d = w * (1 - f) / (1 + f)
            b = [x + d, y, x + w + d, y + h]
        else:                # a copy shifted along both axes (same IoU from both)
            s = 1 - np.sqrt(2 * f / (1 + f))
            b = [x + w * s, y + h * s, x + w + w * s, y + h + h * s]
        boxes += [a, b]
    boxes = np.asarray(boxes, np.float32)
    scores = rng.permutation(len(boxes)).astype(np.float32) / len(boxes) * 0.9 + 0.05
    order = rng.permutation(len(boxes))              # partners far apart in the list as well
    boxes, scores = boxes[order], scores[order]
    want = O.nms(boxes, scores, thr)
    dropped = len(boxes) - len(want)
    assert n_pairs * 0.2 < dropped < n_pairs * 0.8   # the cases do straddle the threshold
    keep = rh.nms(dev(boxes), dev(scores), thr).cpu().numpy()
    assert np.array_equal(keep, want)


@pytest.mark.parametrize("kind", ["one_bin", "two_values", "below_the_bins", "logits", "wide"])
def test_nms_score_distributions_of_the_rank_sort(rh, kind):
    """The rank sort buckets the list by the scores' upper bits and compares a key only inside its
    bin; a bin too full for that (equal scores, scores below 1/32 or above 1) sends the block back
    to counting over the whole list.  Either way: ranks by (score, lower index first)."""
    k = 3000
    boxes, scores = rand_dets(11, k)
    rng = np.random.default_rng(12)
    if kind == "one_bin":
        scores[:] = 0.5
    elif kind == "two_values":
        scores[:] = np.where(rng.random(k) < 0.5, 0.25, 0.75)
    elif kind == "below_the_bins":
        scores[:] = rng.random(k).astype(np.float32) * 0.03
    elif kind == "logits":
        scores[:] = rng.normal(0, 4, k).astype(np.float32)
    else:
        scores[:] = np.exp(rng.uniform(np.log(0.04), 0, k)).astype(np.float32)
    keep = rh.nms(dev(boxes), dev(scores), 0.5).cpu().numpy()
    assert np.array_equal(keep, O.nms(boxes, scores, 0.5))


def test_nms_legacy_and_per_class(rh):
    boxes, scores = rand_dets(77, 700)
    dets = np.concatenate([boxes, scores[:, None]], 1)
    assert np.array_equal(rh.nms(dev(dets), 0.4).cpu().numpy(), O.nms(boxes, scores, 0.4))
    labels = np.random.default_rng(1).integers(0, 5, 700)
    got = rh.batched_nms(dev(boxes), dev(scores), dev(labels), 0.5).cpu().numpy()
    assert np.array_equal(got, O.nms(boxes, scores, 0.5, labels=labels))


def test_nms_large_global_sort(rh):
    boxes, scores = rand_dets(5, 20000, size=3000.0)     # beyond the LDS sort: global bitonic path
    keep = rh.nms(dev(boxes), dev(scores), 0.5).cpu().numpy()
    assert np.array_equal(keep, O.nms(boxes, scores, 0.5))


def test_nms_empty_and_degenerate(rh):
    assert rh.nms(torch.zeros(0, 4, device="cuda"), torch.zeros(0, device="cuda"), 0.5).numel() == 0
    boxes = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [5, 5, 5, 5], [5, 5, 5, 5], [0, 0, 10, 5.0]], np.float32)
    scores = np.array([0.9, 0.9, 0.5, 0.4, 0.3], np.float32)
    keep = rh.nms(dev(boxes), dev(scores), 0.5).cpu().numpy()
    assert np.array_equal(keep, O.nms(boxes, scores, 0.5))
    assert np.array_equal(keep, O.nms_bruteforce(boxes, scores, 0.5))


# ------------------------------------------------------------------ post-processing
def infer_batch(seed, bsz, size, c, mu=-5.0):
    rng = np.random.default_rng(seed)
    anc = O.anchors_for_image(size, size)
    n = anc.shape[1]
    z = rng.normal(mu, 1.2, (bsz, n, c))
    cls = (1.0 / (1.0 + np.exp(-z))).astype(np.float32)
    reg = rng.normal(0, 0.5, (bsz, n, 4)).astype(np.float32)
    return cls, reg, anc


def check_post(rh, cls, reg, anc, size, **kw):
    res = rh.postprocess(dev(cls), dev(reg), dev(anc), (size, size), **kw)
    okw = {k: v for k, v in kw.items() if k != "max_candidates"}
    # the device's own decode + clip (checked against the oracle's to 1e-6 of the box scale) feeds
    # the oracle's NMS, so keep lists compare exactly even where exp() differs in the last place
    dboxes = rh.ClipBoxes()(rh.BBoxTransform()(dev(anc), dev(reg)),
                            torch.zeros(1, 3, size, size, device="cuda")).cpu().numpy()
    oboxes = O.clip_boxes(O.bbox_transform(anc, reg), size, size)
    scale = np.maximum(np.max(np.abs(oboxes), axis=2, keepdims=True), 1.0)
    assert (np.abs(dboxes.astype(np.float64) - oboxes) <= 1e-6 * scale).all()
    for b in range(cls.shape[0]):
        s, l, bx = O.postprocess_image(cls[b], reg[b], anc, size, size, boxes=dboxes[b], **okw)
        got = res[b]
        assert np.array_equal(got["labels"].cpu().numpy(), l), f"image {b}"
        assert np.array_equal(got["scores"].cpu().numpy(), s)
        assert np.array_equal(got["boxes"].cpu().numpy(), bx)
    return res


def test_postprocess_agnostic(rh):
    cls, reg, anc = infer_batch(1, 3, 256, 80)
    cls[1] = 0.01                                        # image without candidates
    res = check_post(rh, cls, reg, anc, 256)
    assert res[1]["scores"].numel() == 0 and res[1]["boxes"].shape == (0, 4)
    tup = rh.postprocess(dev(cls), dev(reg), dev(anc), torch.zeros(3, 3, 256, 256), output="tuple")
    assert len(tup) == 3 and torch.equal(tup[0], res[0]["scores"])


def test_operators_in_the_reference_forward_pattern(rh):
    """The operators one by one, called the way the tail of the reference forward calls them
    (models/retinanet/model.py:315-340: anchors, loss in training; decode, clip, max over classes,
    threshold on image 0, nms, gather in evaluation, with stock torch ops in between): the result is
    the fused postprocess() bit for bit, and the loss is the fused module's."""
    cls, reg, anc_ref = infer_batch(21, 2, 192, 12)
    img = torch.zeros(2, 3, 192, 192, device="cuda")
    classification, regression = dev(cls), dev(reg)
    anchors_op, decode_op, clip_op = rh.Anchors(), rh.BBoxTransform(), rh.ClipBoxes()
    anchors = anchors_op(img)
    assert np.array_equal(anchors.cpu().numpy(), anc_ref)
    # evaluation branch
    boxes = clip_op(decode_op(anchors, regression), img)
    scores = torch.max(classification, dim=2, keepdim=True)[0]
    over = (scores > 0.05)[0, :, 0]
    assert int(over.sum()) > 100
    cand_cls, cand_boxes, cand_scores = classification[:, over, :], boxes[:, over, :], scores[:, over, :]
    keep = rh.nms(cand_boxes[0, :, :], cand_scores[0, :, :], 0.5)
    assert keep.dtype == torch.int64
    top_scores, top_class = cand_cls[0, keep, :].max(dim=1)
    top_boxes = cand_boxes[0, keep, :]
    fused = rh.postprocess(classification, regression, anchors, img, output="tuple")
    assert torch.equal(fused[0], top_scores) and torch.equal(fused[1], top_class)
    assert torch.equal(fused[2], top_boxes)
    s, l, bx = O.postprocess_image(cls[0], reg[0], anc_ref, 192, 192, boxes=boxes[0].cpu().numpy())
    assert np.array_equal(top_scores.cpu().numpy(), s) and np.array_equal(top_class.cpu().numpy(), l)
    # the legacy call of the removed extension (model.py:335): dets [K,5] and a threshold
    keep_legacy = rh.nms(torch.cat([cand_boxes, cand_scores], dim=2)[0, :, :], 0.5)
    assert torch.equal(keep_legacy, keep)
    # nothing over the threshold: three empty tensors (model.py:327-329)
    none = rh.postprocess(classification * 0.0, regression, anchors, img, output="tuple")
    assert [tuple(t.shape) for t in none] == [(0,), (0,), (0, 4)]
    # training branch: the dict train.py:181-190 consumes
    ann = -np.ones((2, 4, 5), np.float32)
    ann[0, :2] = [[20, 30, 90, 120, 3], [100, 40, 180, 100, 7]]
    ann[1, :1] = [[60, 60, 130, 150, 0]]
    c_in, r_in = classification.clone().requires_grad_(True), regression.clone().requires_grad_(True)
    loss_dict = rh.FocalLoss()(c_in, r_in, anchors, dev(ann))
    assert set(loss_dict) == {"classification_loss", "regression_loss"}
    loss = sum(loss for loss in loss_dict.values())
    loss.backward()
    ref = O.focal_loss_fwd_bwd(cls, reg, anc_ref, ann)
    assert np.isclose(loss_dict["classification_loss"].item(), ref["classification_loss"][0], rtol=RTOL)
    assert np.isclose(loss_dict["regression_loss"].item(), ref["regression_loss"][0], rtol=RTOL)
    grad_close(c_in.grad.cpu().numpy(), ref["d_classifications"])
    grad_close(r_in.grad.cpu().numpy(), ref["d_regressions"], atol_rel=REG_ATOL_REL, kind="d_reg")


def test_postprocess_dense_tiles_and_image_borders(rh):
    """Candidate lists with full 64-row tiles, candidates on both sides of the tile that two
    images share, equal scores, and a capacity that is met exactly."""
    cls, reg, anc = infer_batch(5, 3, 256, 20, mu=-9.0)
    n = anc.shape[1]
    assert n % 64 != 0                                   # images share tiles
    cls[0, 1000:1200, 7] = 0.9                           # 200 consecutive candidates, equal scores
    cls[0, -5:, 3] = 0.8                                 # last rows of image 0 ...
    cls[1, :5, 4] = 0.7                                  # ... and first rows of image 1: one tile
    cls[2, 64:128, :] = 0.6                              # one full tile, all classes tied -> label 0
    res = check_post(rh, cls, reg, anc, 256)
    assert (res[2]["labels"][res[2]["scores"] == 0.6] == 0).all()
    cnt = [int((cls[b].max(axis=1) > 0.05).sum()) for b in range(3)]
    check_post(rh, cls, reg, anc, 256, max_candidates=max(cnt))           # exactly full
    check_post(rh, cls, reg, anc, 256, max_candidates=max(cnt) - 1)       # one short: second pass
    check_post(rh, cls, reg, anc, 256, mode="per_class", pre_nms_topk=150)


@pytest.mark.parametrize("size,c,from_logits", [((96, 128), 8, False), ((64, 64), 5, True),
                                                ((256, 256), 80, True)])
def test_postprocess_on_head_layout(rh, size, c, from_logits):
    """postprocess on the per-level [B, A*C, H, W] / [B, A*4, H, W] tensors == postprocess on the
    flat tensors the reference builds from them, bit for bit."""
    rng = np.random.default_rng(50 + c)
    h, w = size
    anc = O.anchors_for_image(h, w)
    shapes = [((h + 2 ** l - 1) // 2 ** l, (w + 2 ** l - 1) // 2 ** l) for l in (3, 4, 5, 6, 7)]
    a, bsz = 9, 3
    mu = -7.5 if c == 80 else -5.0
    z = [rng.normal(mu, 1.5, (bsz, a * c, y, x)).astype(np.float32) for (y, x) in shapes]
    if from_logits:
        z[0][1, :c, :2, :] = 3.0             # a run of equal scores across classes and cells
        z[0][2] = -20.0                      # image 2: candidates on the coarser levels only
        cls_lv = z
    else:
        cls_lv = [(1.0 / (1.0 + np.exp(-t))).astype(np.float32) for t in z]
        cls_lv[0][1, :c, :2, :] = 0.95
        cls_lv[0][2] = 0.0
    reg_lv = [rng.normal(0, 0.4, (bsz, a * 4, y, x)).astype(np.float32) for (y, x) in shapes]
    cls_flat, reg_flat = flat_from_levels(cls_lv, a, c), flat_from_levels(reg_lv, a, 4)
    an = dev(anc)
    for kw in (dict(), dict(mode="per_class", pre_nms_topk=200), dict(max_candidates=64)):
        lv = rh.postprocess([dev(t) for t in cls_lv], [dev(t) for t in reg_lv], an, (h, w),
                            from_logits=from_logits, **kw)
        fl = rh.postprocess(dev(cls_flat), dev(reg_flat), an, (h, w), from_logits=from_logits, **kw)
        assert len(lv) == len(fl) == bsz
        for ra, rb in zip(lv, fl):
            for k in ("scores", "labels", "boxes"):
                assert torch.equal(ra[k], rb[k]), (kw, k)
        assert lv[0]["scores"].numel() > 0 and lv[1]["scores"].numel() > 0


def test_sigmoid_and_postprocess_on_logits(rh):
    """The logits-in post-processing equals the probability form applied to the library's own
    float32 sigmoid, bit for bit; that sigmoid is within a few ulp of numpy's / the oracle's."""
    rng = np.random.default_rng(31)
    x = np.concatenate([rng.normal(-4, 3, 5000), [0.0, -0.0, 9.21034, -9.21034, 20, -20, 88, -88,
                                                    100, -100, 1e4, -1e4, np.inf, -np.inf]]).astype(np.float32)
    got = rh.sigmoid(dev(x)).cpu().numpy()
    want = O.sigmoid_f32(x)
    assert got.dtype == np.float32 and np.isfinite(got).all()
    assert (got >= 0).all() and (got <= 1).all()
    # a few ulp from the reciprocal, exp2 and numpy's own exp, plus the rounding of the exp2
    # argument x * log2(e), which is worth up to |x| ulp; below x = -60 (p < 1e-26) the value
    # is only required to be tiny (exp2 is capped at 1e30 inside)
    mid = np.abs(x) <= 60
    ulp = np.spacing(np.abs(want[mid]))
    assert (np.abs(got[mid].astype(np.float64) - want[mid]) <= (4 + 2 * np.abs(x[mid])) * ulp).all()
    assert (got[x < -60] <= 1e-26).all() and (got[x > 60] == 1.0).all()

    size, c, bsz = 160, 20, 3
    anc = O.anchors_for_image(size, size)
    n = anc.shape[1]
    logits = rng.normal(-6.0, 2.2, (bsz, n, c)).astype(np.float32)
    logits[0, :40, 3] = np.linspace(-3.2, -2.7, 40, dtype=np.float32)     # around logit(0.05) = -2.944
    logits[1, 10:14, :] = 12.0                                              # saturated ties: first class wins
    reg = rng.normal(0, 0.4, (bsz, n, 4)).astype(np.float32)
    xl, rg, an = dev(logits), dev(reg), dev(anc)
    probs = rh.sigmoid(xl)
    for mode, topk in (("agnostic", None), ("per_class", 300)):
        a = rh.postprocess(xl, rg, an, (size, size), mode=mode, pre_nms_topk=topk, from_logits=True)
        b = rh.postprocess(probs, rg, an, (size, size), mode=mode, pre_nms_topk=topk)
        for ra, rb in zip(a, b):
            assert ra["scores"].numel() > 0
            for k in ("scores", "labels", "boxes"):
                assert torch.equal(ra[k], rb[k]), (mode, k)
    # and against the oracle working from the same probabilities
    boxes = rh.ClipBoxes()(rh.BBoxTransform()(an, rg), torch.zeros(1, 3, size, size, device="cuda"))
    so, lo, bo = O.postprocess_image(probs[1].cpu().numpy(), reg[1], anc, size, size,
                                     boxes=boxes[1].cpu().numpy())
    res = rh.postprocess(xl, rg, an, (size, size), from_logits=True)[1]
    assert np.array_equal(res["scores"].cpu().numpy(), so)
    assert np.array_equal(res["labels"].cpu().numpy(), lo)
    assert np.array_equal(res["boxes"].cpu().numpy(), bo)


def test_postprocess_many_images_with_long_lists(rh):
    """More images than a sixth of the compute units, each with a few thousand candidates: the
    rank sort gives an image only a handful of workgroups, which then take several clusters of 64
    list entries each, in more than one pass; lists longer than three quarters of the room make
    the pair kernel's workgroups take a second unit."""
    cls, reg, anc = infer_batch(31, 44, 192, 8, mu=-4.6)
    found = (cls.max(axis=2) > 0.05).sum(axis=1)
    assert found.min() > 1500 and found.max() < 4096, found
    check_post(rh, cls, reg, anc, 192)
    check_post(rh, cls[:3], reg[:3], anc, 192, max_candidates=int(found[:3].max()))   # room met exactly
    # nearly all scores equal: one score bin holds the list, and the workgroups count over the whole
    # list instead, pass after pass
    tied = np.where(cls > 0.05, 0.25, 0.0).astype(np.float32)
    tied[:, ::7, 0] = 0.75
    check_post(rh, tied, reg, anc, 192)


def test_postprocess_per_class_topk_maxdet(rh):
    cls, reg, anc = infer_batch(2, 2, 192, 13, mu=-4.0)
    check_post(rh, cls, reg, anc, 192, mode="per_class")
    check_post(rh, cls, reg, anc, 192, mode="per_class", pre_nms_topk=300)
    check_post(rh, cls, reg, anc, 192, pre_nms_topk=1000, max_det=50)


def boxes_as_head_outputs(boxes, scores, labels, c):
    """Inputs whose decoded boxes are `boxes` (zero deltas on anchors equal to them) with one
    score per row at its label's column."""
    n = len(boxes)
    cls = np.zeros((1, n, c), np.float32)
    cls[0, np.arange(n), labels] = scores
    return cls, np.zeros((1, n, 4), np.float32), np.asarray(boxes, np.float32)[None]


def check_post_hw(rh, cls, reg, anc, hw, **kw):
    h, w = hw
    res = rh.postprocess(dev(cls), dev(reg), dev(anc), hw, **kw)[0]
    dboxes = rh.ClipBoxes()(rh.BBoxTransform()(dev(anc), dev(reg)),
                            torch.empty(1, 3, h, w, device="cuda")).cpu().numpy()
    s, l, bx = O.postprocess_image(cls[0], reg[0], anc, h, w, boxes=dboxes[0], **kw)
    assert np.array_equal(res["labels"].cpu().numpy(), l)
    assert np.array_equal(res["scores"].cpu().numpy(), s)
    assert np.array_equal(res["boxes"].cpu().numpy(), bx)
    return len(s)


@pytest.mark.parametrize("hw", [(1000, 1300), (3000, 20000)])
@pytest.mark.parametrize("case", ["threshold", "one_spot", "shapes"])
def test_postprocess_pair_search_on_hard_box_sets(rh, case, hw):
    """Box sets that strain the pair search of the fused call, which for thresholds from one half
    on and images of moderate size bins the boxes by area and centre and otherwise (the wide
    image here) compares clusters of the ranked list: pairs whose IoU is the threshold give or take
    an ulp with a centre on the partner's edge; thousands of boxes around one spot (one bin holds
    them all); thin, tiny, empty and image-sized boxes, and boxes that the clip turns inside out."""
    h, w = hw
    rng = np.random.default_rng(len(case) + h)
    c = 4
    if case == "threshold":
        thr = 0.5
        boxes = []
        for i in range(1500):                # (3 800 boxes in all: the list fits the default room)
            bw, bh = rng.uniform(8, 100), rng.uniform(8, 100)
            x, y = rng.uniform(0, min(w, 1300) - 110), rng.uniform(0, h - 110)
            f = thr * (1.0 + rng.choice([-3, -1, 0, 1, 3]) * 6e-8 * rng.integers(0, 3))
            a = [x, y, x + bw, y + bh]
            kind = i % 4
            if kind == 0:        # nested, sharing an edge: the larger box's centre on the smaller's edge
                b = [x, y, x + bw * f, y + bh]
            elif kind == 1:      # nested in a corner
                g = np.sqrt(f)
                b = [x + bw * (1 - g), y + bh * (1 - g), x + bw, y + bh]
            elif kind == 2:      # a shifted copy: equal areas
                d = bw * (1 - f) / (1 + f)
                b = [x + d, y, x + bw + d, y + bh]
            else:                # twice the area, the smaller box inside: IoU = 1/2 from the area bound
                b = [x, y, x + bw, y + 2 * bh * (1.0 + rng.choice([-1, 0, 1]) * 1e-7)]
            boxes += [a, b]
        # areas a hair under a power of two (the edge of an area bin) with a partner half as large
        # again, so in the next bin
        for i in range(400):
            side = 2.0 ** rng.integers(3, 7)
            bw, bh = side * (1 - 3e-6) * rng.choice([1, 2]), side
            x, y = rng.uniform(0, min(w, 1300) - 300), rng.uniform(0, h - 300)
            grow = rng.uniform(1.3, 1.9)
            boxes += [[x, y, x + bw, y + bh], [x, y, x + bw, y + bh * grow]]
        boxes = np.asarray(boxes, np.float32)
    elif case == "one_spot":
        k = 3500
        ctr = np.array([w * 0.37, h * 0.61]) + rng.normal(0, 3, (k, 2))
        wh = rng.uniform(20, 28, (k, 2))
        boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
        boxes[::50] = boxes[1::50]                       # exact copies
    else:
        k = 3000
        ctr = rng.uniform(0, 1, (k, 2)) * [min(w, 1300), h]
        wh = np.empty((k, 2))
        kinds = rng.integers(0, 7, k)
        wh[kinds == 0] = rng.uniform(0.05, 0.9, ((kinds == 0).sum(), 2))            # under a pixel
        wh[kinds == 1] = np.stack([rng.uniform(1, 3, (kinds == 1).sum()),
                                   rng.uniform(300, 900, (kinds == 1).sum())], 1)   # thin, tall
        wh[kinds == 2] = np.stack([rng.uniform(300, 1200, (kinds == 2).sum()),
                                   rng.uniform(1, 3, (kinds == 2).sum())], 1)       # thin, wide
        wh[kinds == 3] = rng.uniform(700, 2600, ((kinds == 3).sum(), 2))            # clipped to the image
        wh[kinds == 4] = rng.uniform(10, 60, ((kinds == 4).sum(), 2))
        wh[kinds == 5] = 0.0                                                        # empty
        wh[kinds == 6] = rng.uniform(10, 60, ((kinds == 6).sum(), 2))               # outside the image: clipped
        ctr[kinds == 6] = -rng.uniform(80, 300, ((kinds == 6).sum(), 2))            # to a negative width and height
        ctr[kinds == 1, 0] = np.round(ctr[kinds == 1, 0] / 40) * 40                 # so that they do overlap
        ctr[kinds == 2, 1] = np.round(ctr[kinds == 2, 1] / 40) * 40
        boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    n = len(boxes)
    order = rng.permutation(n)
    boxes = boxes[order]
    scores = (rng.permutation(n).astype(np.float32) / n * 0.9 + 0.06).astype(np.float32)
    labels = rng.integers(0, c, n)
    cls, reg, anc = boxes_as_head_outputs(boxes, scores, labels, c)
    kept = check_post_hw(rh, cls, reg, anc, hw)
    assert 0 < kept < n                                   # some suppression did happen
    kept_pc = check_post_hw(rh, cls, reg, anc, hw, mode="per_class")
    assert kept <= kept_pc < n
    check_post_hw(rh, cls, reg, anc, hw, iou_thresh=0.75)
    check_post_hw(rh, cls, reg, anc, hw, iou_thresh=0.45)  # below one half: the cluster form
    check_post_hw(rh, cls, reg, anc, hw, pre_nms_topk=2000, mode="per_class")


def test_postprocess_topk_preselection(rh):
    """pre_nms_topk at most half the candidate room goes through the histogram pre-selection:
    heavy score ties (the bin that straddles the cut holds hundreds of equal scores, resolved by
    anchor index), fewer candidates than the top-k, and a cut deep inside the list."""
    cls, reg, anc = infer_batch(11, 2, 256, 20, mu=-3.5)
    tied = np.round(cls * 16) / 16                     # scores on a grid of 1/16: 0.0625, 0.125, ...
    tied[1] = np.where(tied[1] > 0.1, 0.25, tied[1])   # image 1: one value for nearly everything
    res = check_post(rh, tied.astype(np.float32), reg, anc, 256, pre_nms_topk=500)
    check_post(rh, tied.astype(np.float32), reg, anc, 256, pre_nms_topk=500, mode="per_class")
    check_post(rh, cls, reg, anc, 256, pre_nms_topk=2000)       # more than an image has: all pass
    check_post(rh, cls, reg, anc, 256, pre_nms_topk=64)
    check_post(rh, cls, reg, anc, 256, pre_nms_topk=1, max_det=1)
    assert all(r["scores"].numel() > 0 for r in res)
    # candidate counts: everything over the threshold is still reported
    raw = rh.functional.postprocess_raw(dev(cls), dev(reg), dev(anc), 256, 256, pre_nms_topk=64)
    over = (cls.max(axis=2) > 0.05).sum(axis=1)
    assert np.array_equal(raw["counts"][1].cpu().numpy(), over)
    # ... and although they are three times the room (4096 per image), nothing is dropped: only
    # the few hundred keys from the cut's histogram bin up are ever stored
    assert int(over.min()) > 3 * 4096 - 100 and int(raw["counts"][2].max()) == 0


def test_postprocess_workspace_is_bounded(rh, monkeypatch):
    """Every anchor over the threshold and no pre_nms_topk: the retry would size the NMS bitmap for
    all of them; past the limit the call refuses with advice instead of allocating."""
    cls, reg, anc = infer_batch(12, 2, 256, 4, mu=0.0)
    monkeypatch.setattr(rh.functional, "MAX_POST_WORKSPACE_BYTES", 8 << 20)
    with pytest.raises(rh.RetinaHeadError, match="pre_nms_topk"):
        rh.postprocess(dev(cls), dev(reg), dev(anc), (256, 256))
    res = rh.postprocess(dev(cls), dev(reg), dev(anc), (256, 256), pre_nms_topk=500)
    assert all(0 < r["scores"].numel() <= 500 for r in res)


def test_postprocess_overflow_retry_and_odd_classes(rh):
    cls, reg, anc = infer_batch(3, 2, 128, 601, mu=-7.0)
    check_post(rh, cls, reg, anc, 128, max_candidates=64)      # forces the exact second pass
    cls, reg, anc = infer_batch(4, 1, 64, 3, mu=-2.0)          # C < 4: generic kernel
    check_post(rh, cls, reg, anc, 64)


def test_hip_graph_replay_of_both_paths(rh):
    """The entry points enqueue work only (no allocation, no host read, no synchronisation), so a
    captured hipGraph of the loss and of the post-processing replays on new data with the results
    of direct calls, bit for bit."""
    F = rh.functional
    size, c = 128, 20
    batches = [make_batch(31 + i, 2, size, c, 5, [5, 2]) for i in range(3)]
    anc = dev(batches[0][2])
    s_cls, s_reg, s_ann = (dev(batches[0][k]) for k in (0, 1, 3))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        loss = F.focal_loss_fwd_bwd(s_cls, s_reg, anc, s_ann)
        post = F.postprocess_raw(s_cls, s_reg, anc, size, size, score_thresh=0.02)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        F.focal_loss_fwd_bwd(s_cls, s_reg, anc, s_ann, out=loss)
        F.postprocess_raw(s_cls, s_reg, anc, size, size, score_thresh=0.02, out=post)
    for cls, reg, _, ann in batches[1:]:
        s_cls.copy_(dev(cls))
        s_reg.copy_(dev(reg))
        s_ann.copy_(dev(ann))
        graph.replay()
        torch.cuda.synchronize()
        want = F.focal_loss_fwd_bwd(dev(cls), dev(reg), anc, dev(ann))
        for k in ("losses", "total", "per_image", "num_pos", "d_classifications", "d_regressions"):
            assert torch.equal(loss[k], want[k]), k
        wpost = F.postprocess_raw(dev(cls), dev(reg), anc, size, size, score_thresh=0.02)
        assert torch.equal(post["counts"], wpost["counts"]) and int(post["counts"][0].min()) > 0
        for b in range(2):
            k = int(post["counts"][0, b])
            for name in ("scores", "labels", "boxes"):
                assert torch.equal(post[name][b, :k], wpost[name][b, :k]), name


def test_two_host_threads_on_their_own_streams(rh):
    """nn.DataParallel drives its replicas from several Python threads of one process
    (train.py:137): the library keeps no mutable global state and the wrappers cache workspaces
    per stream, so two threads working at once on their own streams get the single-thread results."""
    import threading
    F = rh.functional
    data = [make_batch(51 + i, 2, 128, 20, 5, [5, 3]) for i in range(2)]
    anc = dev(data[0][2])
    want = []
    for cls, reg, _, ann in data:
        o = F.focal_loss_fwd_bwd(dev(cls), dev(reg), anc, dev(ann))
        p = F.postprocess_raw(dev(cls), dev(reg), anc, 128, 128, score_thresh=0.02)
        want.append((o["losses"].clone(), o["d_classifications"].clone(), p["counts"].clone(), p["scores"].clone()))
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            cls, reg, _, ann = data[i]
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                c, r, a = dev(cls), dev(reg), dev(ann)
                for _ in range(30):
                    o = F.focal_loss_fwd_bwd(c, r, anc, a)
                    p = F.postprocess_raw(c, r, anc, 128, 128, score_thresh=0.02)
                st.synchronize()
                kept = p["counts"][0]
                ok = (torch.equal(o["losses"], want[i][0]) and torch.equal(o["d_classifications"], want[i][1])
                      and torch.equal(p["counts"], want[i][2])
                      and all(torch.equal(p["scores"][b, :int(kept[b])], want[i][3][b, :int(kept[b])])
                              for b in range(2)))
                if not ok:
                    errors.append(f"thread {i}: results differ")
        except Exception as e:                       # surfaced in the main thread below
            errors.append(f"thread {i}: {e!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    assert not errors, errors


def test_two_host_threads_with_lists_of_different_room(rh):
    """Kernels that use more than 48 KB of dynamic LDS opt in through an attribute of the kernel,
    which every host thread sets before its launch: set to each launch's own size, the thread with
    the smaller lists could lower it just before the other thread's larger launch, which then
    fails.  Two threads post-process at once with room for 4 096 and 8 192 candidates (57 and
    112 KB in the rank sort); every call must succeed and give the single-thread result."""
    import threading
    F = rh.functional
    cls, reg, anc = infer_batch(61, 2, 192, 8, mu=-4.6)
    c, r, a = dev(cls), dev(reg), dev(anc)
    caps = [4096, 8192]
    want = [F.postprocess_raw(c, r, a, 192, 192, cand_cap=cap) for cap in caps]
    want = [{k: v.clone() for k, v in w.items() if torch.is_tensor(v)} for w in want]
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(300):
                    p = F.postprocess_raw(c, r, a, 192, 192, cand_cap=caps[i])
                st.synchronize()
                kept = p["counts"][0]
                if not (torch.equal(p["counts"], want[i]["counts"]) and
                        all(torch.equal(p["boxes"][b, :int(kept[b])], want[i]["boxes"][b, :int(kept[b])])
                            for b in range(2))):
                    errors.append(f"thread {i}: results differ")
        except Exception as e:                       # surfaced in the main thread below
            errors.append(f"thread {i}: {e!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=100)
    assert not errors, errors


def test_zz_tolerance_floors_report(rh):
    """Runs after the gradient tests of this file, of the full-size configurations and of the
    two-rank runs (file and name order) and states what their absolute
    floors were needed for: how many elements passed only thanks to a floor, how much of the
    floor was used, and the worst purely relative error among elements above 1e-3 of the largest
    gradient.  The numbers go to gpurun_out/tolerance_floors.json."""
    import json
    import os
    assert FLOOR_STATS, "no gradient comparison ran before this test"
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "tolerance_floors.json"), "w") as f:
        json.dump(FLOOR_STATS, f, indent=1, sort_keys=True)
    print("\ntolerance floors:", json.dumps(FLOOR_STATS, sort_keys=True))
    cls, reg = FLOOR_STATS["d_cls"], FLOOR_STATS["d_reg"]
    # classification gradients: both terms of dL/dp have the same sign, nothing cancels
    assert cls["needed_floor"] == 0 and cls["worst_rel_above_milli_max"] <= 1e-5
    # regression gradients: the floor is 9 ulp(t) of the quadratic zone, a few 1e-6 of the largest
    assert reg["floor_used_rel"] <= REG_ATOL_REL and reg["needed_floor"] <= 1e-3 * reg["nonzero"]
