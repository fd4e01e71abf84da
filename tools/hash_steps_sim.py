#!/usr/bin/env python3
"""CPU count of the steps the centre-hash pair search takes on the benchmark's candidate lists: per
owner the runs it reaches and their entries, per thread (four per owner) the steps of the flattened
loop, and the maxima over the lanes of a wave / the waves of a workgroup / the image -- which is
what the kernel's duration follows."""
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/pytorch-retinanet_amd')
import bench
from oracle import dense_head_oracle as O

size, C, G, SHARE, THREADS = 608, 80, 16, 4, 512
anc = O.anchors_for_image(size, size)[0]
n = anc.shape[0]
icls, ireg = bench.synth_infer(np.random.default_rng(3000), 2, n, C)
E0 = 8
top = int(np.floor(np.log2(size * size)))
classes = min(max(top - E0 + 1, 1), 20)
gsz = [max(16.0, np.sqrt(2.0 ** (E0 + c))) for c in range(classes)]
nxs = [int(size / g) + 1 for g in gsz]


def klass(area):
    return int(min(max(np.floor(np.log2(area)) - E0, 0), classes - 1))


for j in range(2):
    sc = icls[j].max(1)
    cand = np.nonzero(sc > 0.05)[0]
    boxes = O.clip_boxes(O.bbox_transform(anc[None, cand], ireg[j][None, cand]), size, size)[0]
    order = np.argsort(-sc[cand], kind="stable")
    b = boxes[order]
    K = len(b)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    cx = (b[:, 0] + b[:, 2]) / 2; cy = (b[:, 1] + b[:, 3]) / 2
    cls_ = np.array([klass(a) for a in area])
    count = {}
    for i in range(K):
        c = cls_[i]
        key = (c, min(int(cy[i] / gsz[c]), nxs[c] - 1), min(int(cx[i] / gsz[c]), nxs[c] - 1))
        count[key] = count.get(key, 0) + 1
    runs = np.zeros(K, int); ents = np.zeros(K, int); steps = np.zeros((K, SHARE), int)
    for i in range(K):
        for c in sorted({cls_[i], klass(area[i] * 2.00002)}):
            g = gsz[c]
            ix0, ix1 = int(b[i, 0] / g), min(int(b[i, 2] / g), nxs[c] - 1)
            iy0, iy1 = int(b[i, 1] / g), min(int(b[i, 3] / g), nxs[c] - 1)
            for iy in range(iy0, iy1 + 1):
                ln = sum(count.get((c, iy, ix), 0) for ix in range(ix0, ix1 + 1))
                runs[i] += 1; ents[i] += ln
                for s in range(SHARE):
                    steps[i, s] += 1 + max(0, (ln - s + SHARE - 1) // SHARE)
    print(f"image {j}: K {K}  runs/owner mean {runs.mean():.1f} max {runs.max()}  entries/owner mean {ents.mean():.1f} "
          f"p90 {np.percentile(ents, 90):.0f} p99 {np.percentile(ents, 99):.0f} max {ents.max()}")
    tsteps = steps.max(1)
    print(f"   steps per owner thread: mean {tsteps.mean():.1f} p90 {np.percentile(tsteps, 90):.0f} max {tsteps.max()}")
    wave_max = []
    for g in range(G):
        own = np.arange(g, K, G)
        per_wave = [tsteps[own[w * 16:(w + 1) * 16]].max() for w in range((len(own) + 15) // 16)]
        # owners beyond one pass of the workgroup go to the same waves again
        waves = THREADS // 64
        tot = [sum(per_wave[w::waves]) for w in range(min(waves, len(per_wave)))]
        wave_max.append(max(tot))
    print(f"   busiest wave per workgroup: mean {np.mean(wave_max):.1f} max {max(wave_max)};"
          f" balanced over all lanes it would be {steps.sum() / (G * THREADS):.1f} steps")
