"""usage (on the GPU box): python3 tools/check_post_variant.py <lib.so> ...
Post-processing of a 608 px batch by library variants (tools/build_variant.sh), every image
against the oracle fed with the device's decoded boxes; prints one line per library."""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def check(lib):
    sys.path.insert(0, os.path.join(ROOT, "pytorch-retinanet_amd"))
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from retinahead import _lib
    _lib.LIB_PATH = os.path.abspath(lib)
    import retinahead as rh
    from oracle import dense_head_oracle as O
    dev = torch.device("cuda:0")
    size, c, bsz = 608, 80, 6
    anchors = rh.Anchors()(torch.zeros(1, 3, size, size, device=dev))
    n = anchors.shape[1]
    gen = torch.Generator(device=dev).manual_seed(0)
    cls = torch.sigmoid(torch.randn((bsz, n, c), device=dev, generator=gen) * 1.2 - 7.0)
    reg = torch.randn((bsz, n, 4), device=dev, generator=gen) * 0.5
    boxes = rh.ClipBoxes()(rh.BBoxTransform()(anchors, reg), torch.zeros(1, 3, size, size, device=dev))
    bad = 0
    for mode in ("agnostic", "per_class"):
        res = rh.postprocess(cls, reg, anchors, (size, size), mode=mode)
        for b in range(bsz):
            so, lo, bo = O.postprocess_image(cls[b].cpu().numpy(), reg[b].cpu().numpy(), anchors.cpu().numpy(),
                                             size, size, mode=mode, boxes=boxes[b].cpu().numpy())
            got = res[b]
            same = (np.array_equal(got["boxes"].cpu().numpy(), bo) and np.array_equal(got["scores"].cpu().numpy(), so)
                    and np.array_equal(got["labels"].cpu().numpy(), lo))
            bad += not same
    print(f"{os.path.basename(lib)}: {'all images as the oracle' if not bad else f'{bad} IMAGES DIFFER'}")
    return bad


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--one":
        sys.exit(1 if check(sys.argv[2]) else 0)
    rc = 0
    for lib in sys.argv[1:]:          # a process per library: the library is loaded once per process
        rc |= subprocess.call([sys.executable, os.path.abspath(__file__), "--one", lib])
    sys.exit(rc)
