#!/usr/bin/env python3
"""Does a PostprocessPipeline depend on the streams the process has used before it?  A fresh process, five
pipelines one after the other, each warmed by 100 batches and timed over 400; optionally some streams drawn
from torch's pool first (RH_USE_DRAWN=1: and used).  With streams from torch's pool, the pipeline that
came after ~32 used streams ran at 80-93 us per batch instead of 65; with streams of its own, none does.
usage: python3 tools/pipe_first.py [streams drawn first]"""
import os
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
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
sets = []
for i in range(3):
    c, r = bench.synth_infer(np.random.default_rng(3000 + i), 16, n, 80)
    sets.append((torch.from_numpy(c).to(dev), torch.from_numpy(r).to(dev)))
drawn = [torch.cuda.Stream(device=dev) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0)]
if os.environ.get("RH_USE_DRAWN"):          # ... and used, so that each has a hardware queue
    from retinahead import functional as F
    for s_ in drawn:
        with torch.cuda.stream(s_):
            F.postprocess_raw(sets[0][0], sets[0][1], anchors, 608, 608)
torch.cuda.synchronize()
for trial in range(5):
    pipe = rh.PostprocessPipeline(depth=6, streams=3)

    def run(k):
        for i in range(k):
            if len(pipe) == 6:
                pipe.results()
            pipe.submit(sets[i % 3][0], sets[i % 3][1], anchors, (608, 608))
        while len(pipe):
            pipe.results()
    run(100)
    ts = []
    for _ in range(3):
        t1 = time.perf_counter()
        run(400)
        ts.append(round((time.perf_counter() - t1) / 400 * 1e6, 1))
    print(f"pipeline {trial}: {ts} us per batch; streams {[hex(s.cuda_stream)[-6:] for s in pipe.streams]}")
