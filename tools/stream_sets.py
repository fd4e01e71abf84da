"""Is the rate of four batches in flight a property of WHICH four streams are used?  Eight sets of
four consecutive streams from torch's pool, each measured three times, in two orders."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import bench
import retinahead as rh
from retinahead import functional as F
dev = torch.device("cuda:0")
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
icls_np, ireg_np = bench.synth_infer(np.random.default_rng(3000), 16, n, 80)
icls, ireg = torch.from_numpy(icls_np).to(dev), torch.from_numpy(ireg_np).to(dev)
post = lambda out=None: F.postprocess_raw(icls, ireg, anchors, 608, 608, cand_cap=4096, out=out)
streams = [torch.cuda.Stream(device=dev) for _ in range(32)]
print("stream handles:", [hex(s.cuda_stream & 0xFFFFFF) for s in streams[:8]])
outs = {}
def measure(idx):
    ss = [streams[i] for i in idx]
    for s in ss:
        if s not in outs:
            with torch.cuda.stream(s):
                outs[s] = post()
    torch.cuda.synchronize()
    def run(k):
        for _ in range(k):
            for s in ss:
                with torch.cuda.stream(s):
                    post(outs[s])
        torch.cuda.synchronize()
    run(60)
    t1 = time.perf_counter()
    run(100)
    return (time.perf_counter() - t1) / 400 * 1e6
measure([0, 1, 2, 3])
for rnd in range(3):
    print("consecutive sets:", [round(measure(list(range(4 * k, 4 * k + 4))), 1) for k in range(8)])
print("strided sets (i, i+8, i+16, i+24):", [round(measure([k, k + 8, k + 16, k + 24]), 1) for k in range(8)])
print("strided sets (i, i+1, i+16, i+17):", [round(measure([k, k + 1, k + 16, k + 17]), 1) for k in range(0, 16, 2)])
print("pairs 0+k alone (2 streams):", [round(measure([0, k]) , 1) for k in range(1, 9)])
