#!/usr/bin/env python3
"""Stream-stage time of the loss at the 1024 px configuration against the annotation rows M and the
number of valid boxes among them (the split form of the assignment, M > 128, leaves four times as
many per-block positive counts for every tile to sum)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import bench, retinahead as rh
from retinahead import _lib, functional as F
dev = torch.device("cuda:0")
bsz, size, c = 4, 1024, 80
anchors = rh.Anchors()(torch.zeros(1, 3, size, size, device=dev))
n = anchors.shape[1]
rng = np.random.default_rng(4000)
cls_np, reg_np, ann_full = bench.synth_train(rng, bsz, n, c, size, 256)
ann_full[:] = -1
for j in range(bsz):       # 256 valid boxes per image
    v = 256
    cx, cy = rng.uniform(0, size, v), rng.uniform(0, size, v)
    w = np.exp(rng.uniform(np.log(16), np.log(size / 2), v)); h = np.exp(rng.uniform(np.log(16), np.log(size / 2), v))
    x1, y1 = np.clip(cx - w / 2, 0, size - 1), np.clip(cy - h / 2, 0, size - 1)
    x2, y2 = np.clip(cx + w / 2, x1 + 1, size), np.clip(cy + h / 2, y1 + 1, size)
    ann_full[j, :, :4] = np.stack([x1, y1, x2, y2], 1); ann_full[j, :, 4] = rng.integers(0, c, v)
cls, reg = torch.from_numpy(cls_np).to(dev), torch.from_numpy(reg_np).to(dev)
for m, valid in ((256, 256), (256, 8), (128, 128), (128, 8), (256, 64), (128, 64)):
    a = ann_full[:, :m].copy(); a[:, valid:] = -1
    ann = torch.from_numpy(a).to(dev)
    bufs = F.focal_loss_fwd_bwd(cls, reg, anchors, ann)
    fn = lambda: F.focal_loss_fwd_bwd(cls, reg, anchors, ann, stages=_lib.STAGE_STREAM, out=bufs)
    for _ in range(10): fn()
    ms, mn = bench.event_train(fn, 20, 10)
    afn = lambda: F.focal_loss_fwd_bwd(cls, reg, anchors, ann, stages=_lib.STAGE_ASSIGN, out=bufs)
    for _ in range(5): afn()
    ams, amn = bench.event_train(afn, 20, 5)
    print(f"M={m:4d} valid={valid:4d} positives={int(bufs['num_pos'].sum()):6d}  stream {ms*1e3:7.1f} us (best train {mn*1e3:7.1f})  assign {ams*1e3:6.1f} us")
