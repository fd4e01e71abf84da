#!/usr/bin/env python3
"""Soak of the one-launch training step (stream + in-launch reduction + the next step's assignment,
two captured graphs replayed in turn) at the bench workload: every replay must leave the plain
step's losses and gradient checksum, bit for bit."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import bench
import retinahead as rh
from retinahead import functional as F

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda:0")
anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
n = anchors.shape[1]
cls_np, reg_np, ann_np = bench.synth_train(np.random.default_rng(2000), 16, n, 80, 608, 20)
cls = torch.from_numpy(cls_np).to(dev).requires_grad_(True)
reg = torch.from_numpy(reg_np).to(dev).requires_grad_(True)
ann = torch.from_numpy(ann_np).to(dev)
ref = F.focal_loss_fwd_bwd(cls.detach(), reg.detach(), anchors, ann)
torch.cuda.synchronize()
ref_losses = ref["losses"].clone()
ref_sum = ref["d_classifications"].double().sum().item()
mod = rh.FocalLoss()

def step():
    cls.grad = None
    reg.grad = None
    mod.prefetch_assignment(anchors, ann, 80)
    out = mod(cls, reg, anchors, ann)
    sum(out.values()).backward()
    return out["classification_loss"], out["regression_loss"], cls.grad

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(4):
        step()
torch.cuda.synchronize()
graphs, outs = [], []
for _ in range(2):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        outs.append(step())
    graphs.append(g)
bad = 0
t0 = time.time()
for r in range(reps):
    graphs[r % 2].replay()
    if r % 97 == 96 or r == reps - 1:
        torch.cuda.synchronize()
        c, g_, d = outs[r % 2]
        ok = (torch.equal(torch.cat([c, g_]), ref_losses[:2]) and d.double().sum().item() == ref_sum)
        bad += not ok
    if r % 5000 == 4999:
        print(f"rep {r + 1}: {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
print("soak", "FAILED" if bad else "ok", f"({reps} replays, mismatches {bad})")
sys.exit(1 if bad else 0)
