import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    sys.path.insert(0, p)
import retinahead._lib as _L
if os.environ.get("RH_LIB"): _L.LIB_PATH = os.path.join(ROOT, os.environ["RH_LIB"])
import retinahead as rh
from oracle import dense_head_oracle as O
rng = np.random.default_rng(0)
for k in (40, 64, 100, 300, 1000):
    for rep in range(2):
        c = rng.uniform(0, 300, (k, 2)); wh = rng.uniform(10, 90, (k, 2))
        b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        s = rng.random(k).astype(np.float32)
        got = rh.nms(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), 0.5).cpu().numpy()
        want = O.nms(b, s, 0.5)
        ok = np.array_equal(got, want)
        print(k, rep, "ok" if ok else f"DIFF got {len(got)} want {len(want)}", flush=True)
        if not ok:
            extra = sorted(set(got.tolist()) - set(want.tolist()))
            order = np.argsort(-s, kind="stable"); rank = np.empty(k, int); rank[order] = np.arange(k)
            print("  kept wrongly (index: rank, list cluster):", [(i, int(rank[i]), i // 64) for i in extra[:10]])
            for i in extra[:5]:
                iou = O.calc_iou(b[i:i+1], b)[0]
                sup = [j for j in want if iou[j] > 0.5 and rank[j] < rank[i]]
                print(f"   box {i} (rank {rank[i]}) should be suppressed by", [(j, int(rank[j]), j // 64, j % 64) for j in sup[:6]])
for k in (2000, 3000, 4096):
    c = rng.uniform(0, 600, (k, 2)); wh = rng.uniform(10, 90, (k, 2))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    s = rng.random(k).astype(np.float32)
    got = rh.nms(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), 0.5).cpu().numpy()
    want = O.nms(b, s, 0.5)
    print("nms", k, "ok" if np.array_equal(got, want) else f"DIFF got {len(got)} want {len(want)}", flush=True)
# post-processing: B images, capacity 4096 against the oracle per image
size, c = 256, 8
anc = O.anchors_for_image(size, size)
n = anc.shape[1]
for bsz in (1, 2, 5, 16):
    cls = (1 / (1 + np.exp(-rng.normal(-5.5, 1.2, (bsz, n, c))))).astype(np.float32)
    reg = rng.normal(0, 0.5, (bsz, n, 4)).astype(np.float32)
    tc, tr, ta = torch.from_numpy(cls).cuda(), torch.from_numpy(reg).cuda(), torch.from_numpy(anc).cuda()
    for cap, topk in ((4096, 0), (1024, 0), (4096, 300), (4096, 1000)):
        out = rh.functional.postprocess_raw(tc, tr, ta, size, size, cand_cap=cap, pre_nms_topk=topk)
        cnt = out["counts"].cpu().numpy()
        boxes_all = rh.ClipBoxes()(rh.BBoxTransform()(ta, tr), torch.zeros(1, 3, size, size, device="cuda")).cpu().numpy()
        for j in range(bsz):
            so, lo, bo = O.postprocess_image(cls[j], reg[j], anc, size, size, boxes=boxes_all[j],
                                             pre_nms_topk=topk or None)
            kk = int(cnt[0, j])
            ok = kk == len(so) and np.array_equal(out["scores"][j, :kk].cpu().numpy(), so)
            print(f"post B={bsz} cap={cap} topk={topk} image {j}: cands {cnt[1, j]} overflow {cnt[2, j]} kept {kk} want {len(so)}", "ok" if ok else "DIFF", flush=True)
