#!/usr/bin/env python3
"""Decode + NMS throughput with several batches in flight on separate HIP streams."""
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
from retinahead import functional as F  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
icls_np, ireg_np = bench.synth_infer(np.random.default_rng(3000), B, n, 80)
icls, ireg = torch.from_numpy(icls_np).to(dev), torch.from_numpy(ireg_np).to(dev)
post = lambda out=None: F.postprocess_raw(icls, ireg, anchors, 608, 608, cand_cap=4096, out=out)
for ns in (1, 2, 3, 4, 6, 8):
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    outs = []
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            outs.append(post())
    torch.cuda.synchronize()
    rounds = max(120 // ns, 10)
    t1 = time.perf_counter()
    for _ in range(rounds):
        for st, o in zip(streams, outs):
            with torch.cuda.stream(st):
                post(o)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) / (rounds * ns) * 1e3
    print(f"B={B} streams={ns}: {ms * 1e3:7.1f} us/batch  {B / ms * 1e3:9.0f} img/s")

# host cost of one call (nothing else queued), and the same pipeline replayed from hipGraphs
t1 = time.perf_counter()
for _ in range(200):
    post(outs[0])
host_us = (time.perf_counter() - t1) / 200 * 1e6
torch.cuda.synchronize()
print(f"host time per postprocess_raw call (enqueue only): {host_us:.1f} us")
for ns in (1, 2, 3, 4, 6):
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    graphs, outs = [], []
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            o = post()
            post(o)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            post(o)
        graphs.append(g)
        outs.append(o)
    torch.cuda.synchronize()
    rounds = max(120 // ns, 10)
    t1 = time.perf_counter()
    for _ in range(rounds):
        for st, g in zip(streams, graphs):
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) / (rounds * ns) * 1e3
    print(f"B={B} streams={ns} (hipGraph replay): {ms * 1e3:7.1f} us/batch  {B / ms * 1e3:9.0f} img/s")
