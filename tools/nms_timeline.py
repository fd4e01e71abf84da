#!/usr/bin/env python3
"""Where the time goes inside the post-processing kernels: runs the bench workload's
post-processing against the diagnostic build of the library (make -C pytorch-retinanet_amd/csrc
timeline), which stamps a 100 MHz wall clock at marked places of every workgroup, and prints per
kernel the span from its first workgroup's start and the median / maximum stamp of each mark."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import retinahead._lib as L  # noqa: E402

L.LIB_PATH = os.path.join(ROOT, "pytorch-retinanet_amd", "csrc", "build", os.environ.get("RH_TL_LIB", "libretinahead_tl.so"))
import bench  # noqa: E402
import retinahead as rh  # noqa: E402
from retinahead import functional as F  # noqa: E402

KERNELS = {0: ("sort_rank", ["start", "count pass", "list in LDS", "ranks counted", "end", "bins laid out", "bins filled"]),
           1: ("nms_pairs (hash form: boxes binned / table built)",
               ["start", "counters read", "own cluster loaded", "end"]),
           2: ("nms_resolve", ["start", "words loaded", "rounds done", "end", "round 1 done", None,
                               "round 2 done", "round 3 done"])}
BLOCKS, SLOTS = 8192, 12


def main():
    size, bsz, c = 608, 16, 80
    dev = torch.device("cuda:0")
    anchors = rh.Anchors()(torch.zeros(1, 3, size, size, device=dev))
    n = anchors.shape[1]
    icls_np, ireg_np = bench.synth_infer(np.random.default_rng(3000), bsz, n, c)
    icls, ireg = torch.from_numpy(icls_np).to(dev), torch.from_numpy(ireg_np).to(dev)
    res = F.postprocess_raw(icls, ireg, anchors, size, size, cand_cap=int(os.environ.get("RH_CAND_CAP", "4096")))
    for _ in range(5):
        F.postprocess_raw(icls, ireg, anchors, size, size, cand_cap=int(os.environ.get("RH_CAND_CAP", "4096")), out=res)
    torch.cuda.synchronize()
    lib = L.load()
    lib.rh_tl_read.restype = ctypes.c_int
    lib.rh_tl_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    buf = np.zeros((3, BLOCKS, SLOTS), np.int64)
    assert lib.rh_tl_read(buf.ctypes.data, buf.size) == 0
    for k, (name, marks) in KERNELS.items():
        t = buf[k]
        live = t[:, 0] > 0
        t = t[live]
        t0 = t[:, 0].min()
        print(f"{name}: {live.sum()} workgroups, starts spread over {(t[:, 0].max() - t0) / 100:.2f} us")
        for m, label in enumerate(marks):
            if label is None:
                continue
            v = t[:, m]
            v = v[v > 0]
            if v.size:
                rel = (v - t0) / 100.0
                print(f"   {label:22s} median {np.median(rel):7.2f}  p90 {np.percentile(rel, 90):7.2f}  max {rel.max():7.2f} us"
                      f"   ({v.size} workgroups)")
        if k in (1, 2):
            a, b = (2, 3) if k == 1 else (0, 2)
            ok = (t[:, 9] > 0) & (t[:, b] > t[:, a])
            if ok.any():
                ghz = (t[ok, 9] - t[ok, 8]) / ((t[ok, b] - t[ok, a]) * 10.0)
                print(f"   shader clock while it ran: median {np.median(ghz):.2f} GHz (min {ghz.min():.2f}, max {ghz.max():.2f})")
        if k == 1 and not (t[:, 9] > 0).any():       # the hash form stamps the four marks only
            walk = (t[:, 3] - t[:, 2]) / 100.0
            print(f"   walk of the owners (wave 0): median {np.median(walk):.2f}  p90 {np.percentile(walk, 90):.2f}  max {walk.max():.2f} us")
        elif k == 1:
            done = t[t[:, 3] > 0]
            dur = (done[:, 3] - done[:, 2]) / 100.0
            print(f"   work per workgroup (wave 0): median {np.median(dur):.2f} max {dur.max():.2f} us;"
                  f" cluster pairs median {np.median(done[:, 4]):.0f} max {done[:, 4].max()},"
                  f" survivors median {np.median(done[:, 5]):.0f} max {done[:, 5].max()}, total pairs {done[:, 4].sum()}")
            print(f"   inside the cluster pairs (wave 0): survivor loops median {np.median(done[:, 6]) / 100:.2f} "
                  f"max {done[:, 6].max() / 100:.2f} us")
            st, ns = done[:, 7] // 1000, done[:, 7] % 1000
            print(f"   rounds of 4 with a possible hit (wave 0): median {np.median(ns):.0f} max {ns.max()}; time in them median "
                  f"{np.median(st) / 100:.2f} max {st.max() / 100:.2f} us; worst loops: " +
                  ", ".join(f"{a / 100:.1f}us/{b}surv/{c}hitrounds/{d / 100:.1f}us" for a, b, c, d in
                            sorted(zip(done[:, 6], done[:, 5], ns, st), reverse=True)[:6]))
            full = buf[k]
            idx = np.nonzero(full[:, 3] > 0)[0]
            slices = int(os.environ.get("RH_SLICES", "1"))
            a_of = (idx // bsz) // slices
            print("   by cluster A: blocks, start median, end median / max (us), survivors (wave 0) median / max, loop us max")
            for a in sorted(set(a_of.tolist())):
                r = full[idx[a_of == a]]
                print(f"     A={a:3d}: {len(r):3d}  {np.median(r[:, 0] - t0) / 100:6.2f}  {np.median(r[:, 3] - t0) / 100:6.2f} / "
                      f"{(r[:, 3] - t0).max() / 100:6.2f}   {np.median(r[:, 5]):4.0f} / {r[:, 5].max():4d}   {r[:, 6].max() / 100:6.2f}")
        if k == 2:
            print(f"   rounds: {sorted(set(t[:, 5].tolist()))}")


if __name__ == "__main__":
    main()
