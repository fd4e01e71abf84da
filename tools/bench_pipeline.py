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


# One stream runs the score scans back to back (the HBM-bound stage); the rest of each batch
# (sort, mask, sweep) follows on one of a few other streams; a ring of buffer sets keeps
# several batches in flight.
from retinahead import _lib  # noqa: E402
for n_tail, ring in ((1, 3), (2, 4), (3, 6), (4, 8)):
    scan_stream = torch.cuda.Stream(device=dev)
    tails = [torch.cuda.Stream(device=dev) for _ in range(n_tail)]
    slots = []
    for _ in range(ring):
        o = post()
        slots.append(dict(out=o, ws=torch.empty(o["workspace_bytes"], dtype=torch.uint8, device=dev),
                          scanned=torch.cuda.Event(), done=torch.cuda.Event()))
    torch.cuda.synchronize()

    def run(batches):
        for b in range(batches):
            sl = slots[b % ring]
            with torch.cuda.stream(scan_stream):
                if b >= ring:
                    scan_stream.wait_event(sl["done"])          # the slot's previous batch is finished
                F.postprocess_raw(icls, ireg, anchors, 608, 608, cand_cap=4096, stages=_lib.POST_DECODE,
                                  out=sl["out"], ws=sl["ws"])
                sl["scanned"].record(scan_stream)
            tl = tails[b % n_tail]
            with torch.cuda.stream(tl):
                tl.wait_event(sl["scanned"])
                F.postprocess_raw(icls, ireg, anchors, 608, 608, cand_cap=4096,
                                  stages=_lib.POST_SORT | _lib.POST_NMS, out=sl["out"], ws=sl["ws"])
                sl["done"].record(tl)
    run(2 * ring)
    torch.cuda.synchronize()
    nb = 240
    t1 = time.perf_counter()
    run(nb)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) / nb * 1e3
    ref = post()
    torch.cuda.synchronize()
    same = True
    for si, sl in enumerate(slots):
        for k in ("counts", "scores", "labels", "boxes"):
            if k == "counts":
                eq = torch.equal(sl["out"][k], ref[k])
            else:      # rows beyond an image's count are not written
                kept = ref["counts"][0]
                eq = all(torch.equal(sl["out"][k][i, :int(kept[i])], ref[k][i, :int(kept[i])]) for i in range(B))
            if not eq:
                same = False
                print("   slot", si, k, "differs")
    print(f"B={B} scan stream + {n_tail} tail streams, ring {ring}: {ms * 1e3:7.1f} us/batch  "
          f"{B / ms * 1e3:9.0f} img/s  results {'identical' if same else 'DIFFER'}")


# (Giving the scan stream and the tail streams disjoint shares of the compute units would be the
# next step, but hipExtStreamCreateWithCUMask fails on this pool: "out of memory".)
