#!/usr/bin/env python3
"""In-kernel stamps of topk_select_kernel at the dense-GT configuration (4 x 1024 px, top-1000),
from the diagnostic build (make -C pytorch-retinanet_amd/csrc timeline)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import retinahead._lib as L
L.LIB_PATH = os.path.join(ROOT, "pytorch-retinanet_amd", "csrc", "build", "libretinahead_tl.so")
import bench
import retinahead as rh
from retinahead import functional as F
size, bsz, c = 1024, 4, 80
dev = torch.device("cuda:0")
anchors = rh.Anchors()(torch.zeros(1, 3, size, size, device=dev))
n = anchors.shape[1]
icls_np, ireg_np = bench.synth_infer(np.random.default_rng(4000), bsz, n, c)
icls, ireg = torch.from_numpy(icls_np).to(dev), torch.from_numpy(ireg_np).to(dev)
res = F.postprocess_raw(icls, ireg, anchors, size, size, cand_cap=8192, pre_nms_topk=1000)
for _ in range(5):
    F.postprocess_raw(icls, ireg, anchors, size, size, cand_cap=8192, pre_nms_topk=1000, out=res)
torch.cuda.synchronize()
print("counts", res["counts"].cpu().numpy().tolist())
lib = L.load()
lib.rh_tl_read.restype = ctypes.c_int
lib.rh_tl_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros((3, 8192, 12), np.int64)
assert lib.rh_tl_read(buf.ctypes.data, buf.size) == 0
names = ["start", "loads issued + LDS zeroed", "histogram done", "cut found", "positions known", "written"]
for j in range(bsz):
    t = buf[0, j, 5:11]
    print(f"image {j}: " + "  ".join(f"{nm} {(v - t[0]) / 100:.2f}" for nm, v in zip(names, t)))
    r = buf[0, j, 0:5]
    print(f"   rank block 0: start {(r[0] - t[0]) / 100:.2f} end {(r[4] - t[0]) / 100:.2f} us after the selection's start")
