#!/usr/bin/env python3
"""Calibrates bench.py's CPU baseline: times the reference's own loss
(models/retinanet/losses.py, loaded by path with the two process-local shims of
tools/gen_golden.py) beside oracle/torch_cpu_path.py on the same inputs and thread count, forward
+ backward, and prints the ratio.  Build container only (needs /root/reference); CPU only.

  python tools/time_reference_cpu.py [--images 4] [--threads 8] [--runs 5]

The inputs are bench.py's synthetic batch (608 px, C=80, up to 20 GT boxes per image), cut to
--images images so that a run takes seconds here.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import gen_golden  # noqa: E402
from oracle import dense_head_oracle as O  # noqa: E402
from oracle import torch_cpu_path as T  # noqa: E402


def best_of(fn, runs):
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return min(times), float(np.median(times))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--runs", type=int, default=5)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    gen_golden.install_shims()
    losses = gen_golden.load("ref_losses", "models/retinanet/losses.py")
    ref = losses.FocalLoss()

    anc = O.anchors_for_image(bench.SIZE, bench.SIZE)
    n = anc.shape[1]
    cls_np, reg_np, ann_np = bench.synth_train(np.random.default_rng(2000), a.images, n, bench.CLASSES,
                                               bench.SIZE, bench.MAX_GT)
    anchors, ann = torch.from_numpy(anc), torch.from_numpy(ann_np)
    cls = torch.from_numpy(cls_np).requires_grad_(True)
    reg = torch.from_numpy(reg_np).requires_grad_(True)
    res = {}

    def run_ref():
        cls.grad = reg.grad = None
        out = ref(cls, reg, anchors, ann)
        (out["classification_loss"] + out["regression_loss"]).sum().backward()
        res["ref"] = (float(out["classification_loss"].detach()), float(out["regression_loss"].detach()), cls.grad.clone(), reg.grad.clone())

    def run_port():
        cls.grad = reg.grad = None
        out = T.dense_head_loss(cls, reg, anchors, ann)
        (out["classification_loss"] + out["regression_loss"]).sum().backward()
        res["port"] = (float(out["classification_loss"].detach()), float(out["regression_loss"].detach()), cls.grad.clone(), reg.grad.clone())

    run_ref()
    run_port()
    same = (np.allclose(res["ref"][:2], res["port"][:2], rtol=1e-6)
            and torch.allclose(res["ref"][2], res["port"][2], rtol=1e-5, atol=0)
            and torch.allclose(res["ref"][3], res["port"][3], rtol=1e-5, atol=1e-9))
    ref_best, ref_med = best_of(run_ref, a.runs)
    port_best, port_med = best_of(run_port, a.runs)
    rate = lambda s: a.images * n / s
    print(json.dumps({
        "images": a.images, "anchors_per_image": n, "classes": bench.CLASSES, "threads": a.threads,
        "runs": a.runs, "results_agree": bool(same),
        "reference_s_best": ref_best, "reference_s_median": ref_med,
        "port_s_best": port_best, "port_s_median": port_med,
        "reference_anchors_per_s": rate(ref_best), "port_anchors_per_s": rate(port_best),
        "port_over_reference_speed": ref_best / port_best,
    }, indent=1))


if __name__ == "__main__":
    main()
