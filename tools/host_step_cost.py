"""Host time of one eager FocalLoss step (announce the next batch, forward, sum, backward) at the
bench workload: the enqueue loop alone (the GPU lags behind), then a cProfile of it."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import bench
import retinahead as rh
dev = torch.device("cuda:0")
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
cls_np, reg_np, ann_np = bench.synth_train(np.random.default_rng(2000), 16, n, 80, 608, 20)
cls, reg, ann = (torch.from_numpy(x).to(dev) for x in (cls_np, reg_np, ann_np))
cls.requires_grad_(True); reg.requires_grad_(True)
mod = rh.FocalLoss(batch_divisor=16)
def step():
    cls.grad = None
    reg.grad = None
    mod.prefetch_assignment(anchors, ann, 80)
    out = mod(cls, reg, anchors, ann)
    loss = sum(l for l in out.values())
    loss.backward()
for _ in range(20):
    step()
torch.cuda.synchronize()
for reps in (10, 20, 40):          # short bursts: the queue does not fill, so this is host time
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        step()
    host = (time.perf_counter() - t) / reps
    torch.cuda.synchronize()
    total = (time.perf_counter() - t) / reps
    print(f"{reps} steps: host {host*1e6:.1f} us/step, with the GPU's tail {total*1e6:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
