#!/usr/bin/env python3
"""Marginal cost of each post-processing stage with three, two and one batches in flight: the pipeline is run
with the score scan alone, with scan + sort, and with all stages."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import bench
import retinahead as rh
from retinahead import _lib, functional as F
dev = torch.device("cuda:0")
B = 16
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
icls_np, ireg_np = bench.synth_infer(np.random.default_rng(3000), B, n, 80)
icls, ireg = torch.from_numpy(icls_np).to(dev), torch.from_numpy(ireg_np).to(dev)
for ns in (3, 2, 1):
    for label, st in (("scan", _lib.POST_DECODE), ("scan+sort", _lib.POST_DECODE | _lib.POST_SORT), ("all", _lib.POST_ALL)):
        post = lambda out=None: F.postprocess_raw(icls, ireg, anchors, 608, 608, cand_cap=4096, out=out, stages=st)
        full = lambda out=None: F.postprocess_raw(icls, ireg, anchors, 608, 608, cand_cap=4096, out=out)
        res = []
        for trial in range(5):
            streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
            outs = []
            for s in streams:
                s.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(s):
                    outs.append(full())
            torch.cuda.synchronize()
            def run(k):
                for _ in range(k):
                    for s, o in zip(streams, outs):
                        with torch.cuda.stream(s):
                            post(o)
                torch.cuda.synchronize()
            run(60)
            t1 = time.perf_counter()
            run(100)
            res.append((time.perf_counter() - t1) / (100 * ns) * 1e6)
        print(f"streams={ns} {label:10s} median {np.median(res):6.1f} us/batch   trials {[round(x, 1) for x in res]}", flush=True)
