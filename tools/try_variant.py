"""Post-processing timings (bench workload) against another build of the library:
   tools/try_variant.py <path to .so>  ->  sort stage alone, whole call on one stream, three batches in flight."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import retinahead._lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
import retinahead as rh
from retinahead import functional as F
dev = torch.device("cuda:0")
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
icls_np, ireg_np = bench.synth_infer(np.random.default_rng(3000), 16, n, 80)
icls, ireg = torch.from_numpy(icls_np).to(dev), torch.from_numpy(ireg_np).to(dev)
def post(out=None, stages=L.POST_ALL):
    return F.postprocess_raw(icls, ireg, anchors, 608, 608, cand_cap=int(os.environ.get("RH_CAP", "4096")), out=out, stages=stages)
res = post()
ref = {k: v.clone() for k, v in res.items() if torch.is_tensor(v)}
def train(fn, launches=20, reps=10):
    for _ in range(launches):
        fn()
    ms = bench.event_train(fn, launches, reps)[0]
    return ms * 1e3
print(os.path.basename(L.LIB_PATH), "kept", int(res["counts"][0].sum()))
print(f"  sort alone {train(lambda: post(res, L.POST_SORT)):.1f} us   nms alone {train(lambda: post(res, L.POST_NMS)):.1f} us   whole call {train(lambda: post(res)):.1f} us")
NS = int(os.environ.get("RH_STREAMS", "3"))       # batches in flight
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
outs = []
for s in streams:
    with torch.cuda.stream(s):
        outs.append(post())
torch.cuda.synchronize()
def run(k):
    for _ in range(k):
        for s, o in zip(streams, outs):
            with torch.cuda.stream(s):
                post(o)
    torch.cuda.synchronize()
run(150)
tr = []
for _ in range(5):
    t1 = time.perf_counter(); run(100); tr.append((time.perf_counter() - t1) / (100 * NS) * 1e6)
print(f"  {NS} in flight:" if NS != 3 else "  three in flight:", [round(x, 1) for x in tr])
for k in ("scores", "labels", "boxes", "counts"):
    assert torch.equal(outs[0][k], ref[k]), k
