"""PostprocessPipeline on HIP streams created directly (not torch's pool): rate per batch."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import bench
import retinahead as rh
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
def own_stream():
    h = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0      # hipStreamNonBlocking
    return torch.cuda.ExternalStream(h.value, device=dev)
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
icls_np, ireg_np = bench.synth_infer(np.random.default_rng(3000), 16, n, 80)
icls, ireg = torch.from_numpy(icls_np).to(dev), torch.from_numpy(ireg_np).to(dev)
for kind in ("own", "pool", "own", "pool", "own", "own"):
    pipe = rh.PostprocessPipeline(depth=8, streams=4)
    if kind == "own":
        pipe.streams = [own_stream() for _ in range(4)]
    def run(k):
        for _ in range(k):
            if len(pipe) == 8:
                pipe.results()
            pipe.submit(icls, ireg, anchors, (608, 608))
        while len(pipe):
            pipe.results()
    run(100)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run(400)
    print(kind, f"{(time.perf_counter() - t1) / 400 * 1e6:.1f} us/batch", flush=True)
