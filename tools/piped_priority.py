#!/usr/bin/env python3
"""Three batches in flight with the stages behind the score scan on a HIGH-priority stream of their own
(per slot: scan on a normal stream, rank sort + pairs + resolve on a high-priority one, events between
them), against the usual one stream per slot.  (usage on the GPU box: python3 tools/piped_priority.py)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import retinahead._lib as L  # noqa: E402
import bench  # noqa: E402
import retinahead as rh  # noqa: E402
from retinahead import functional as F  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
sets = []
for i in range(ns):
    c, r = bench.synth_infer(np.random.default_rng(3000 + i), 16, n, 80)
    sets.append((torch.from_numpy(c).to(dev), torch.from_numpy(r).to(dev)))
lo = [torch.cuda.Stream(device=dev, priority=0) for _ in range(ns)]
hi = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(ns)]
outs = []
for s, (c, r) in zip(lo, sets):
    with torch.cuda.stream(s):
        outs.append(F.postprocess_raw(c, r, anchors, 608, 608))
torch.cuda.synchronize()
ref = [{k: v.clone() for k, v in o.items() if torch.is_tensor(v)} for o in outs]
ev_scan = [torch.cuda.Event() for _ in range(ns)]
ev_tail = [torch.cuda.Event() for _ in range(ns)]


def run_plain(k):
    for _ in range(k):
        for s, o, (c, r) in zip(lo, outs, sets):
            with torch.cuda.stream(s):
                F.postprocess_raw(c, r, anchors, 608, 608, out=o)
    torch.cuda.synchronize()


def run_split(k, tails):
    for _ in range(k):
        for i, (o, (c, r)) in enumerate(zip(outs, sets)):
            with torch.cuda.stream(lo[i]):
                lo[i].wait_event(ev_tail[i])          # the slot's lists are free again
                F.postprocess_raw(c, r, anchors, 608, 608, out=o, stages=L.POST_DECODE)
                ev_scan[i].record(lo[i])
            with torch.cuda.stream(tails[i]):
                tails[i].wait_event(ev_scan[i])
                F.postprocess_raw(c, r, anchors, 608, 608, out=o, stages=L.POST_SORT | L.POST_NMS)
                ev_tail[i].record(tails[i])
    torch.cuda.synchronize()


def timed(fn, *a):
    fn(100, *a)
    tr = []
    for _ in range(5):
        t1 = time.perf_counter()
        fn(100, *a)
        tr.append((time.perf_counter() - t1) / (100 * ns) * 1e6)
    return sorted(round(x, 1) for x in tr)


for i in range(ns):
    ev_tail[i].record(lo[i])
torch.cuda.synchronize()
print(f"{ns} slots, one stream per slot:                     ", timed(run_plain), "us per batch")
print(f"{ns} slots, tails on a high-priority stream per slot:", timed(run_split, hi), "us per batch")
mid = [torch.cuda.Stream(device=dev, priority=0) for _ in range(ns)]
print(f"{ns} slots, tails on a second normal stream per slot:", timed(run_split, mid), "us per batch")
for o, w in zip(outs, ref):
    for k in ("scores", "labels", "boxes", "counts"):
        assert torch.equal(o[k], w[k]), k
print("results as with one stream per slot")
