"""Host time of PostprocessPipeline.submit / results at the bench workload, and its throughput."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import bench
import retinahead as rh
dev = torch.device("cuda:0")
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
icls_np, ireg_np = bench.synth_infer(np.random.default_rng(3000), 16, n, 80)
icls, ireg = torch.from_numpy(icls_np).to(dev), torch.from_numpy(ireg_np).to(dev)
for depth, width in ((4, 4), (8, 4), (12, 4), (3, 3), (6, 3), (9, 3), (8, 4), (6, 3), (8, 4), (6, 3)):
    pipe = rh.PostprocessPipeline(depth=depth, streams=width)
    ts, tr = [], []
    def run(k):
        for _ in range(k):
            if len(pipe) == depth:
                t0 = time.perf_counter(); pipe.results(); tr.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); pipe.submit(icls, ireg, anchors, (608, 608)); ts.append(time.perf_counter() - t0)
        while len(pipe):
            pipe.results()
    run(100)
    del ts[:], tr[:]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run(400)
    total = (time.perf_counter() - t1) / 400 * 1e6
    print(f"depth {depth} on {width} streams: {total:.1f} us/batch; submit median {np.median(ts)*1e6:.1f} us, results median {np.median(tr)*1e6:.1f} us (p90 {np.percentile(tr,90)*1e6:.1f})")
