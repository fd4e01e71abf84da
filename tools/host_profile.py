#!/usr/bin/env python3
"""Host-side (Python) cost of one FocalLoss forward+backward step: cProfile over N steps."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import retinahead as rh  # noqa: E402

dev = torch.device("cuda:0")
SIZE = int(sys.argv[1]) if len(sys.argv) > 1 else 608      # a tiny size isolates the host cost
anchors = rh.Anchors()(torch.zeros(1, 3, SIZE, SIZE, device=dev))
n = anchors.shape[1]
cls_np, reg_np, ann_np = bench.synth_train(np.random.default_rng(0), 16, n, 80, SIZE, 20)
cls = torch.from_numpy(cls_np).to(dev).requires_grad_(True)
reg = torch.from_numpy(reg_np).to(dev).requires_grad_(True)
ann = torch.from_numpy(ann_np).to(dev)
mod = rh.FocalLoss()


def step():
    cls.grad = None
    reg.grad = None
    out = mod(cls, reg, anchors, ann)
    loss = sum(l for l in out.values())
    loss.backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
# pure host cost: time to ENQUEUE (GPU may lag behind)
t0 = time.perf_counter()
for _ in range(200):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / 200:.1f} us/step, with drain {1e6 * (t2 - t0) / 200:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
