"""Four batches in flight: does ordering the HBM-bound score scans one after another (an event from
each scan gates the next batch's scan) remove the two rates (67 / 72 us) the free-running streams
settle into?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import bench
import retinahead as rh
from retinahead import functional as F, _lib
dev = torch.device("cuda:0")
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
icls_np, ireg_np = bench.synth_infer(np.random.default_rng(3000), 16, n, 80)
icls, ireg = torch.from_numpy(icls_np).to(dev), torch.from_numpy(ireg_np).to(dev)
def post(out=None, stages=_lib.POST_ALL):
    return F.postprocess_raw(icls, ireg, anchors, 608, 608, cand_cap=4096, out=out, stages=stages)
streams = [torch.cuda.Stream(device=dev) for _ in range(24)]
outs = {}
def measure(ss, mode):
    for s in ss:
        if s not in outs:
            with torch.cuda.stream(s):
                outs[s] = post()
    evs = [torch.cuda.Event() for _ in range(len(ss))]
    torch.cuda.synchronize()
    def run(k):
        prev = None
        for _ in range(k):
            for i, s in enumerate(ss):
                with torch.cuda.stream(s):
                    if mode == "free":
                        post(outs[s])
                    else:
                        if prev is not None:
                            s.wait_event(prev)
                        post(outs[s], _lib.POST_DECODE)
                        evs[i].record(s)
                        prev = evs[i]
                        post(outs[s], _lib.POST_SORT | _lib.POST_NMS)
        torch.cuda.synchronize()
    run(60)
    t1 = time.perf_counter()
    run(100)
    return (time.perf_counter() - t1) / (100 * len(ss)) * 1e6
import sys as _s
if len(_s.argv) > 1 and _s.argv[1] == "widths":
    measure(streams[:4], "free")
    for w in (3, 4, 5, 6, 8):
        print(w, "streams:", [round(measure(streams[j:j + w], "free"), 1) for j in (0, 8, 16, 0, 8, 16)], flush=True)
    raise SystemExit
measure(streams[:4], "free")
for mode in ("free", "chain", "free", "chain"):
    print(mode, [round(measure(streams[4 * k:4 * k + 4], mode), 1) for k in range(1, 6)], flush=True)
print("chain, 3 streams", [round(measure(streams[4 * k:4 * k + 3], "chain"), 1) for k in range(1, 6)], flush=True)
print("chain, 2 streams", [round(measure(streams[4 * k:4 * k + 2], "chain"), 1) for k in range(1, 6)], flush=True)
