#!/usr/bin/env python3
"""Randomised comparison of the fused post-processing with the oracle on box sets of many kinds (sizes, aspect
ratios, densities, image shapes, thresholds from one half up, both modes), sized so that the pair search takes
its centre-hash form.  usage on the GPU box: python3 tools/fuzz_pairs.py [cases] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import retinahead as rh  # noqa: E402
from oracle import dense_head_oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
t0 = time.time()
bad = 0
for case in range(cases):
    h, w = int(rng.integers(200, 1400)), int(rng.integers(200, 1400))
    k = int(rng.integers(1536, 4097))
    c = int(rng.integers(1, 7))
    kind = rng.integers(0, 4)
    if kind == 0:      # uniform centres, log-uniform sizes
        ctr = rng.uniform(0, 1, (k, 2)) * [w, h]
        wh = np.exp(rng.uniform(np.log(4), np.log(min(h, w)), (k, 2)))
    elif kind == 1:    # clusters of near-copies
        m = int(rng.integers(5, 200))
        base_c = rng.uniform(0, 1, (m, 2)) * [w, h]
        base_s = np.exp(rng.uniform(np.log(8), np.log(300), (m, 2)))
        pick = rng.integers(0, m, k)
        jit = rng.uniform(0.01, 0.3)
        ctr = base_c[pick] + rng.normal(0, 1, (k, 2)) * base_s[pick] * jit
        wh = base_s[pick] * np.exp(rng.normal(0, jit, (k, 2)))
    elif kind == 2:    # anchor-like: a few sizes and ratios on a grid
        s = rng.choice([16, 32, 64, 128, 256], k) * rng.choice([1.0, 1.26, 1.59], k)
        r = rng.choice([0.5, 1.0, 2.0], k)
        wh = np.stack([s / np.sqrt(r), s * np.sqrt(r)], 1) * np.exp(rng.normal(0, 0.15, (k, 2)))
        ctr = (rng.integers(0, 64, (k, 2)) + 0.5) * [w / 64, h / 64] + rng.normal(0, 3, (k, 2))
    else:              # areas near powers of two, the edges of the area bins
        a = 2.0 ** rng.integers(6, 16, k) * (1 + rng.choice([-1, 1], k) * 10.0 ** rng.uniform(-7, -2, k))
        r = np.exp(rng.uniform(-1, 1, k))
        wh = np.stack([np.sqrt(a * r), np.sqrt(a / r)], 1)
        ctr = rng.uniform(0, 1, (k, 2)) * [w, h]
        ctr[k // 2:] = ctr[:k - k // 2] + rng.normal(0, 2, (k - k // 2, 2))      # partners nearby
    boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    scores = (rng.permutation(k) / k * 0.9 + 0.06).astype(np.float32)
    if rng.random() < 0.3:
        scores = np.round(scores * 8) / 8 + np.float32(0.06)            # heavy ties
    labels = rng.integers(0, c, k)
    cls = np.zeros((1, k, c), np.float32)
    cls[0, np.arange(k), labels] = scores
    reg = np.zeros((1, k, 4), np.float32)
    anc = boxes[None]
    thr = float(rng.choice([0.5, 0.5, 0.55, 0.6, 0.7, 0.9]))
    mode = "per_class" if rng.random() < 0.4 else "agnostic"
    topk = int(rng.integers(1536, k + 1)) if rng.random() < 0.3 else None      # (the best `topk` only)
    dcls, dreg, danc = (torch.from_numpy(x).to(dev) for x in (cls, reg, anc))
    res = rh.postprocess(dcls, dreg, danc, (h, w), iou_thresh=thr, mode=mode, pre_nms_topk=topk)[0]
    dboxes = rh.ClipBoxes()(rh.BBoxTransform()(danc, dreg), torch.empty(1, 3, h, w, device=dev)).cpu().numpy()
    s, l, bx = O.postprocess_image(cls[0], reg[0], anc, h, w, boxes=dboxes[0], iou_thresh=thr, mode=mode,
                                   pre_nms_topk=topk)
    same = (np.array_equal(res["labels"].cpu().numpy(), l) and np.array_equal(res["scores"].cpu().numpy(), s)
            and np.array_equal(res["boxes"].cpu().numpy(), bx))
    if not same:
        bad += 1
        print(f"case {case}: DIFFERS (kind {kind}, {h}x{w}, k {k}, thr {thr}, {mode}, top-k {topk}; kept {res['scores'].numel()} want {len(s)})")
    if case % 20 == 19:
        print(f"{case + 1} cases, {bad} differ, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz: {cases} cases, {bad} differ")
sys.exit(1 if bad else 0)
