"""Back-to-back launches of one kernel on one stream, for a kernel trace: how long after one
launch has ended does the next begin?  `run` enqueues the launches; `gaps <dir>` reads rocprofv3's
kernel trace and prints, per kernel, the duration and the idle time before the next launch."""
import csv, glob, os, sys
import numpy as np

def run():
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "pytorch-retinanet_amd")):
        sys.path.insert(0, p)
    import bench
    import retinahead as rh
    from retinahead import _lib, functional as F
    dev = torch.device("cuda:0")
    anchors = rh.Anchors()(torch.zeros(1, 3, 608, 608, device=dev))
    n = anchors.shape[1]
    cls_np, reg_np, ann_np = bench.synth_train(np.random.default_rng(2000), 16, n, 80, 608, 20)
    cls, reg, ann = (torch.from_numpy(x).to(dev) for x in (cls_np, reg_np, ann_np))
    bufs = F.focal_loss_fwd_bwd(cls, reg, anchors, ann)
    for st in (_lib.STAGE_STREAM, _lib.STAGE_STREAM | _lib.STAGE_FINALIZE):
        for _ in range(150):
            F.focal_loss_fwd_bwd(cls, reg, anchors, ann, stages=st, out=bufs)
        torch.cuda.synchronize()
    icls_np, ireg_np = bench.synth_infer(np.random.default_rng(3000), 16, n, 80)
    icls, ireg = torch.from_numpy(icls_np).to(dev), torch.from_numpy(ireg_np).to(dev)
    out = F.postprocess_raw(icls, ireg, anchors, 608, 608, cand_cap=4096)
    for _ in range(150):
        F.postprocess_raw(icls, ireg, anchors, 608, 608, cand_cap=4096, stages=_lib.POST_DECODE, out=out)
    torch.cuda.synchronize()
    for _ in range(150):
        F.postprocess_raw(icls, ireg, anchors, 608, 608, cand_cap=4096, out=out)
    torch.cuda.synchronize()

def gaps(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    stat = {}
    for a, b in zip(rows, rows[1:]):
        key = (a["Kernel_Name"].split("(")[0][-40:], b["Kernel_Name"].split("(")[0][-40:])
        gap = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if gap < 50000:      # a host pause is not a boundary
            stat.setdefault(key, []).append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"]), gap))
    for key, v in stat.items():
        if len(v) >= 20:
            v = np.array(v)
            print(f"{key[0]:42s} -> {key[1]:42s} n={len(v):4d} duration {np.median(v[:,0])/1e3:7.2f} us, "
                  f"idle before the next: median {np.median(v[:,1])/1e3:5.2f} p10 {np.percentile(v[:,1],10)/1e3:5.2f} p90 {np.percentile(v[:,1],90)/1e3:5.2f} us")

if __name__ == "__main__":
    run() if sys.argv[1] == "run" else gaps(sys.argv[2])
