#!/usr/bin/env python3
"""PostprocessPipeline rate with one shared input tensor against one tensor per submission."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import bench, retinahead as rh
dev = torch.device("cuda:0")
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
c_np, r_np = bench.synth_infer(np.random.default_rng(3000), 16, n, 80)
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cs = [torch.from_numpy(c_np).to(dev)]; rs = [torch.from_numpy(r_np).to(dev)]
for k in range(1, sets):
    cs.append(torch.roll(cs[0], k, 0).contiguous()); rs.append(torch.roll(rs[0], k, 0).contiguous())
for mode in ("same", "rotate", "same", "rotate"):
    pipe = rh.PostprocessPipeline(depth=6, streams=3)
    def run(k):
        for i in range(k):
            if len(pipe) == 6: pipe.results()
            j = i % sets if mode == "rotate" else 0
            pipe.submit(cs[j], rs[j], anchors, (608, 608))
        while len(pipe): pipe.results()
    run(100)
    t = time.perf_counter(); run(400); print(mode, (time.perf_counter() - t) / 400 * 1e6, "us/batch")
