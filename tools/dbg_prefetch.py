import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import retinahead as rh
from oracle import dense_head_oracle as O
from test_gpu_parity import make_batch, dev
size, c, m = 160, 20, 6
batches = [make_batch(50 + i, 3, size, c, m, [m, 0, 2]) for i in range(4)]
anc = dev(batches[0][2])
data = [(dev(b[0]), dev(b[1]), dev(b[3])) for b in batches]
refs = [O.focal_loss_fwd_bwd(b[0], b[1], b[2], b[3], grad_cls=1.0, grad_reg=1.0) for b in batches]
for i, r in enumerate(refs):
    print("oracle", i, float(r["classification_loss"][0]), float(r["regression_loss"][0]))
plain = rh.FocalLoss()
for i, (cls, reg, ann) in enumerate(data):
    out = plain(cls, reg, anc, ann)
    torch.cuda.synchronize()
    print("plain", i, float(out["classification_loss"]), float(out["regression_loss"]), plain.last["per_image"].cpu().numpy().tolist())
mod = rh.FocalLoss()
for i, (cls, reg, ann) in enumerate(data):
    if i + 1 < len(data):
        mod.prefetch_assignment(anc, data[i + 1][2], c)
    out = mod(cls, reg, anc, ann)
    torch.cuda.synchronize()
    print("ahead", i, float(out["classification_loss"]), float(out["regression_loss"]), mod.last["per_image"].cpu().numpy().tolist())
