"""Mean duration of the finalize stage alone at the benchmark's shape (trains of 50 launches)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import bench
import retinahead as rh
from retinahead import _lib, functional as F
if os.environ.get("RH_LIB"):        # another build of the library (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["RH_LIB"])
if len(sys.argv) > 1:               # B,size,C,M instead of the benchmark's shape
    bench.BATCH, bench.SIZE, bench.CLASSES, bench.MAX_GT = (int(v) for v in sys.argv[1].split(","))
dev = torch.device("cuda", 0)
anchors = rh.Anchors()(torch.zeros(1, 3, bench.SIZE, bench.SIZE, device=dev))
n = anchors.shape[1]
cls_np, reg_np, ann_np = bench.synth_train(np.random.default_rng(2000), bench.BATCH, n, bench.CLASSES, bench.SIZE, bench.MAX_GT)
cls, reg, ann = (torch.from_numpy(x).to(dev) for x in (cls_np, reg_np, ann_np))
bufs = F.focal_loss_fwd_bwd(cls, reg, anchors, ann)
for name, st in (("finalize", _lib.STAGE_FINALIZE), ("assign", _lib.STAGE_ASSIGN)):
    fn = lambda: F.focal_loss_fwd_bwd(cls, reg, anchors, ann, stages=st, out=bufs)
    for _ in range(20):
        fn()
    mean, best = bench.event_train(fn, 50, 10)
    print(f"{name}: mean {mean*1e3:.2f} us  best {best*1e3:.2f} us (back-to-back launches, includes the kernel boundary)")
