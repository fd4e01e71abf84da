#!/usr/bin/env python3
"""Launches rh.nms for several K (random clustered boxes) so that rocprofv3 can time the
sort / mask / sweep kernels per K (kernel order in the trace = K order below)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import retinahead as rh  # noqa: E402

KS = [64, 128, 256, 512, 1024, 2048, 4096, 8192]
rng = np.random.default_rng(0)
for k in KS:
    c = rng.uniform(0, 600, (k, 2))
    wh = rng.uniform(20, 120, (k, 2))
    boxes = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)).cuda()
    scores = torch.from_numpy(rng.random(k).astype(np.float32)).cuda()
    for _ in range(6):
        keep = rh.nms(boxes, scores, 0.5)
    print(k, "kept", keep.numel())
