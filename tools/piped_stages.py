#!/usr/bin/env python3
"""Which stage costs the pipelined post-processing its overlap: the bench workload with three batches in
flight on three streams, running only the scan, scan + rank sort, or the whole call in every slot
(usage on the GPU box: python3 tools/piped_stages.py [streams])."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import retinahead._lib as L  # noqa: E402
if os.environ.get("RH_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["RH_LIB"])
import bench  # noqa: E402
import retinahead as rh  # noqa: E402
from retinahead import functional as F  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 3
CAP = int(os.environ.get("RH_CAP", "4096"))      # room of the candidate lists
dev = torch.device("cuda:0")
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
sets = []
for i in range(ns):
    c, r = bench.synth_infer(np.random.default_rng(3000 + i), 16, n, 80)
    sets.append((torch.from_numpy(c).to(dev), torch.from_numpy(r).to(dev)))
streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
outs = []
for s, (c, r) in zip(streams, sets):
    with torch.cuda.stream(s):
        outs.append(F.postprocess_raw(c, r, anchors, 608, 608, cand_cap=CAP))
torch.cuda.synchronize()


def run(k, stages):
    for _ in range(k):
        for s, o, (c, r) in zip(streams, outs, sets):
            with torch.cuda.stream(s):
                F.postprocess_raw(c, r, anchors, 608, 608, cand_cap=CAP, out=o, stages=stages)
    torch.cuda.synchronize()


for name, stages in (("scan", L.POST_DECODE), ("scan + rank sort", L.POST_DECODE | L.POST_SORT),
                     ("rank sort + pairs + resolve (on what the scan left)", L.POST_SORT | L.POST_NMS),
                     ("whole call", L.POST_ALL)):
    run(100, stages)
    tr = []
    for _ in range(5):
        t1 = time.perf_counter()
        run(100, stages)
        tr.append((time.perf_counter() - t1) / (100 * ns) * 1e6)
    print(f"{ns} in flight, {name}: {sorted(round(x, 1) for x in tr)} us per batch")
