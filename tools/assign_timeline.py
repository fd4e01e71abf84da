#!/usr/bin/env python3
"""In-kernel stamps of assign_kernel (diagnostic build: make -C pytorch-retinanet_amd/csrc timeline):
when its workgroups start, how long the three parts of one take (anchors + bounding box, the walk over
the annotation rows, merge + stores) and when the last one ends.
usage: tools/assign_timeline.py [B,size,C,M]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import retinahead._lib as L
L.LIB_PATH = os.environ.get("RH_TL_LIB", os.path.join(ROOT, "pytorch-retinanet_amd", "csrc", "build", "libretinahead_tl.so"))
import bench
import retinahead as rh
from retinahead import functional as F
bsz, size, c, m = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4,1024,80,256").split(","))
dev = torch.device("cuda:0")
anchors = rh.Anchors()(torch.zeros(1, 3, size, size, device=dev))
n = anchors.shape[1]
cls_np, reg_np, ann_np = bench.synth_train(np.random.default_rng(4000), bsz, n, c, size, m)
cls, reg, ann = (torch.from_numpy(v).to(dev) for v in (cls_np, reg_np, ann_np))
bufs = F.focal_loss_fwd_bwd(cls, reg, anchors, ann)
for _ in range(5):
    F.focal_loss_fwd_bwd(cls, reg, anchors, ann, stages=L.STAGE_ASSIGN, out=bufs)
torch.cuda.synchronize()
lib = L.load()
lib.rh_dbg_read.restype = ctypes.c_int
lib.rh_dbg_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
nblk = (n + 127) // 128 if m > 128 else (n + 511) // 512
buf = np.zeros((bsz * nblk, 8), np.int64)
assert lib.rh_dbg_read(buf.ctypes.data, buf.size) == 0
t = buf[:, :4].astype(np.float64) / 100.0       # us
t0 = t[:, 0].min()
start, end = t[:, 0] - t0, t[:, 3] - t0
parts = np.diff(t, axis=1)
q = lambda v: "  ".join(f"p{p} {np.percentile(v, p):6.2f}" for p in (10, 50, 90, 99, 100))
print(f"B,size,C,M = {bsz},{size},{c},{m}: {bsz * nblk} workgroups; last one ends {end.max():.2f} us after the first one starts")
print("workgroup starts   ", q(start))
print("workgroup life     ", q(end - start))
for k, name in enumerate(("anchors + bounding box", "walk over the rows    ", "merge + stores         ")):
    print(name, q(parts[:, k]))
# by block index within the image (blockIdx.y counts from the last block down)
by = buf.reshape(bsz, nblk, 8)[0]
life = (by[:, 3] - by[:, 0]) / 100.0
for lo in range(0, nblk, max(1, nblk // 12)):
    hi = min(nblk, lo + max(1, nblk // 12))
    print(f"  launch order {lo:5d}..{hi - 1:5d} of image 0: life median {np.median(life[lo:hi]):6.2f} max {life[lo:hi].max():6.2f}  start median {np.median((by[lo:hi, 0] / 100.0) - t0):6.2f}")
