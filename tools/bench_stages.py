#!/usr/bin/env python3
"""Per-stage HIP-event timings of the loss and post-processing kernels for the benchmark
configurations (2: COCO-80 608 B=16, 3: 601 classes 800 px B=8, 4: 256 GT 1024 px B=4)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import retinahead as rh  # noqa: E402
from retinahead import _lib, functional as F  # noqa: E402

if os.environ.get("RH_LIB"):        # another build of the library (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["RH_LIB"])

CFG = {5: (128, 256, 80, 20), 1: (16, 512, 80, 20), 2: (16, 608, 80, 20), 3: (8, 800, 601, 8), 4: (4, 1024, 80, 256)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--infer", action="store_true")
    ap.add_argument("--logits", action="store_true", help="loss on logits (sigmoid folded in)")
    ap.add_argument("--levels", action="store_true", help="loss on the heads' [B, A*C, H, W] outputs")
    ap.add_argument("--shape", help="B,size,C,M instead of --cfg")
    ap.add_argument("--topk", type=int, default=0, help="pre-NMS top-k (post-processing)")
    ap.add_argument("--cap", type=int, default=4096, help="candidate capacity per image (post-processing)")
    ap.add_argument("--mu", type=float, default=-7.0, help="mean of the inference logits (post-processing)")
    a = ap.parse_args()
    bsz, size, c, m = [int(x) for x in a.shape.split(",")] if a.shape else CFG[a.cfg]
    dev = torch.device("cuda:0")
    anchors = rh.Anchors()(torch.zeros(1, 3, size, size, device=dev))
    n = anchors.shape[1]
    rng = np.random.default_rng(1000 * a.cfg)
    if not a.infer:
        cls_np, reg_np, ann_np = bench.synth_train(rng, bsz, n, c, size, m)
        if a.cfg == 4:   # every row valid
            pass
        cls, reg, ann = (torch.from_numpy(x).to(dev) for x in (cls_np, reg_np, ann_np))
        if a.logits:
            cls = torch.logit(cls.double()).float().contiguous()
        if a.levels:
            shapes = [((size + 2 ** l - 1) // 2 ** l,) * 2 for l in (3, 4, 5, 6, 7)]
            cls_lv, reg_lv, at = [], [], 0
            for (h, w) in shapes:
                k = h * w * 9
                cls_lv.append(cls[:, at:at + k].reshape(bsz, h, w, 9 * c).permute(0, 3, 1, 2).contiguous())
                reg_lv.append(reg[:, at:at + k].reshape(bsz, h, w, 36).permute(0, 3, 1, 2).contiguous())
                at += k
            bufs = F.focal_loss_levels_fwd_bwd(cls_lv, reg_lv, anchors, ann, from_logits=a.logits)
            print("num_pos", bufs["num_pos"].cpu().numpy().tolist())
            algo = bsz * n * (8 * c + 48)
            for name, st in (("assign", _lib.STAGE_ASSIGN), ("stream", _lib.STAGE_STREAM),
                             ("finalize", _lib.STAGE_FINALIZE), ("all", _lib.STAGE_ALL)):
                fn = lambda: F.focal_loss_levels_fwd_bwd(cls_lv, reg_lv, anchors, ann, stages=st, out=bufs,
                                                         from_logits=a.logits)
                for _ in range(10):
                    fn()
                mean, mn = bench.event_times(fn, a.iters)
                extra = f"  algo {algo / mean / 1e6:8.1f} GB/s (mean) {algo / mn / 1e6:8.1f} (min)" \
                    if name in ("stream", "all") else ""
                print(f"{name:9s} mean {mean * 1e3:8.1f} us  min {mn * 1e3:8.1f} us{extra}")
            return
        bufs = F.focal_loss_fwd_bwd(cls, reg, anchors, ann, from_logits=a.logits)
        print("num_pos", bufs["num_pos"].cpu().numpy().tolist())
        algo = bsz * n * (8 * c + 48)
        for name, st in (("assign", _lib.STAGE_ASSIGN), ("stream", _lib.STAGE_STREAM),
                         ("finalize", _lib.STAGE_FINALIZE), ("all", _lib.STAGE_ALL)):
            fn = lambda: F.focal_loss_fwd_bwd(cls, reg, anchors, ann, stages=st, out=bufs,
                                               from_logits=a.logits)
            for _ in range(10):
                fn()
            mean, mn = bench.event_times(fn, a.iters)
            extra = f"  algo {algo / mean / 1e6:8.1f} GB/s (mean) {algo / mn / 1e6:8.1f} (min)" \
                if name in ("stream", "all") else ""
            print(f"{name:9s} mean {mean * 1e3:8.1f} us  min {mn * 1e3:8.1f} us{extra}")
        if not a.logits:
            # the step as a loop that announces the next batch runs it: one launch (stream, the
            # reduction inside it, the next step's assignment at the end of its grid), two
            # workspaces in turn; trains of 20 back-to-back steps
            wss = [F.focal_workspace(bsz, n, c, m, dev) for _ in range(2)]
            outs = [F.focal_loss_fwd_bwd(cls, reg, anchors, ann, ws=wss[i]) for i in range(2)]
            turn = [0]

            def step():
                i = turn[0] & 1
                turn[0] += 1
                F.focal_loss_fwd_bwd(cls, reg, anchors, ann, stages=_lib.STAGE_STREAM | _lib.STAGE_FINALIZE,
                                     out=outs[i], ws=wss[i], ahead=(ann, wss[1 - i], outs[1 - i]))
            for _ in range(10):
                step()
            mean, mn = bench.event_train(step, 20, 10)
            print(f"step, look-ahead, one launch: mean {mean * 1e3:8.1f} us  best train {mn * 1e3:8.1f} us"
                  f"  algo {algo / mean / 1e6:8.1f} GB/s = {algo / mean / 1e6 / 8000:.3f} of the HBM peak")
            ref = F.focal_loss_fwd_bwd(cls, reg, anchors, ann)
            assert torch.equal(ref["losses"], outs[0]["losses"]) and torch.equal(ref["losses"], outs[1]["losses"])
            # the step of a loop that cannot announce: the assignment's launch, then the stream's
            plain = lambda: F.focal_loss_fwd_bwd(cls, reg, anchors, ann, out=bufs)
            for _ in range(10):
                plain()
            mean, mn = bench.event_train(plain, 20, 10)
            print(f"step, no look-ahead, two launches: mean {mean * 1e3:8.1f} us  best train {mn * 1e3:8.1f} us"
                  f"  = {algo / mean / 1e6 / 8000:.3f} of the HBM peak")
            asg = lambda: F.focal_loss_fwd_bwd(cls, reg, anchors, ann, stages=_lib.STAGE_ASSIGN, out=bufs)
            mean, mn = bench.event_train(asg, 20, 10)
            print(f"assignment launches back to back: mean {mean * 1e3:8.1f} us")
            assert torch.equal(ref["d_classifications"], outs[0]["d_classifications"])
    else:
        icls_np, ireg_np = bench.synth_infer(rng, bsz, n, c, a.mu)
        icls, ireg = torch.from_numpy(icls_np).to(dev), torch.from_numpy(ireg_np).to(dev)
        if a.levels:
            shapes = [((size + 2 ** l - 1) // 2 ** l,) * 2 for l in (3, 4, 5, 6, 7)]
            cls_lv, reg_lv, at = [], [], 0
            for (h, w) in shapes:
                k = h * w * 9
                cls_lv.append(icls[:, at:at + k].reshape(bsz, h, w, 9 * c).permute(0, 3, 1, 2).contiguous())
                reg_lv.append(ireg[:, at:at + k].reshape(bsz, h, w, 36).permute(0, 3, 1, 2).contiguous())
                at += k
            res = F.postprocess_levels_raw(cls_lv, reg_lv, anchors, size, size, cand_cap=4096)
            print("counts", res["counts"].cpu().numpy()[:, :4].tolist())
            algo = bsz * n * (4 * c + 32)
            for name, st in (("decode", _lib.POST_DECODE), ("sort", _lib.POST_SORT),
                             ("nms", _lib.POST_NMS), ("all", _lib.POST_ALL)):
                fn = lambda: F.postprocess_levels_raw(cls_lv, reg_lv, anchors, size, size, cand_cap=4096,
                                                      stages=st, out=res)
                for _ in range(5):
                    fn()
                mean, mn = bench.event_times(fn, a.iters)
                extra = f"  algo {algo / mean / 1e6:8.1f} GB/s" if name in ("decode", "all") else ""
                print(f"{name:9s} mean {mean * 1e3:8.1f} us  min {mn * 1e3:8.1f} us{extra}")
            return
        res = F.postprocess_raw(icls, ireg, anchors, size, size, cand_cap=a.cap, pre_nms_topk=a.topk)
        print("counts", res["counts"].cpu().numpy()[:, :4].tolist())
        algo = bsz * n * (4 * c + 32)
        for name, st in (("decode", _lib.POST_DECODE), ("sort", _lib.POST_SORT),
                         ("nms", _lib.POST_NMS), ("all", _lib.POST_ALL)):
            fn = lambda: F.postprocess_raw(icls, ireg, anchors, size, size, cand_cap=a.cap, stages=st,
                                           pre_nms_topk=a.topk, out=res)
            for _ in range(5):
                fn()
            mean, mn = bench.event_times(fn, a.iters)
            extra = f"  algo {algo / mean / 1e6:8.1f} GB/s" if name in ("decode", "all") else ""
            print(f"{name:9s} mean {mean * 1e3:8.1f} us  min {mn * 1e3:8.1f} us{extra}")


if __name__ == "__main__":
    main()
