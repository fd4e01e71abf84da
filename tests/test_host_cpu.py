"""CPU-only checks: the torch CPU baseline path against the golden vectors, the NMS oracle against a
brute force and hand-computed answers, the C-ABI library's exported symbols, the host logic of the
package (no compute calls: there is no GPU here), and the data-parallel helper under gloo."""
import ctypes
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, PKG_DIR, load_golden
from oracle import dense_head_oracle as O
from oracle import torch_cpu_path as T


# ------------------------------------------------------------------ torch CPU baseline path
@pytest.mark.parametrize("case", ["normal", "holes", "no_pos", "ties", "clamp", "c13", "handmade"])
def test_torch_cpu_path_matches_reference(focal_golden, case):
    d = focal_golden[case]
    cls = torch.from_numpy(d["cls"]).clone().requires_grad_(True)
    reg = torch.from_numpy(d["reg"]).clone().requires_grad_(True)
    out = T.dense_head_loss(cls, reg, torch.from_numpy(d["anchors"]), torch.from_numpy(d["ann"]))
    (float(d["g"][0]) * out["classification_loss"].sum()
     + float(d["g"][1]) * out["regression_loss"].sum()).backward()
    assert np.allclose(out["classification_loss"].detach().numpy(), d["cls_loss"], rtol=1e-6)
    assert np.allclose(out["regression_loss"].detach().numpy(), d["reg_loss"], rtol=1e-6)
    assert np.allclose(cls.grad.numpy(), d["d_cls"], rtol=1e-5, atol=1e-12)
    got_reg = np.zeros_like(d["d_reg"]) if reg.grad is None else reg.grad.numpy()   # no positives
    assert np.allclose(got_reg, d["d_reg"], rtol=1e-5, atol=1e-9)


def test_torch_cpu_postprocess_matches_oracle():
    g = load_golden("decode.npz")
    h, w = int(g["hw"][0]), int(g["hw"][1])
    boxes = T.decode_clip(torch.from_numpy(g["anchors"]), torch.from_numpy(g["deltas"]), h, w)
    scale = np.maximum(np.max(np.abs(g["clipped"]), axis=2, keepdims=True), 1.0)
    assert (np.abs(boxes.numpy().astype(np.float64) - g["clipped"]) <= 2e-6 * scale).all()
    rng = np.random.default_rng(5)
    b = np.concatenate([rng.uniform(0, 80, (300, 2)), rng.uniform(90, 160, (300, 2))], 1).astype(np.float32)
    s = rng.random(300).astype(np.float32)
    keep = T.greedy_nms(torch.from_numpy(b), torch.from_numpy(s), 0.5).numpy()
    assert np.array_equal(keep, O.nms(b, s, 0.5))


# ------------------------------------------------------------------ NMS oracle (parity unpinned)
def test_nms_known_answers():
    boxes = np.array([[0, 0, 10, 10],      # kept (best)
                      [1, 1, 11, 11],      # IoU 81/119 = .68 with 0 -> dropped
                      [0, 0, 10, 5],       # IoU exactly .5 with 0 -> kept (strict >)
                      [20, 20, 30, 30],    # far away -> kept
                      [20, 20, 30, 29]],   # IoU .9 with 3 -> dropped
                     np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.6, 0.5], np.float32)
    assert O.nms(boxes, scores, 0.5).tolist() == [0, 2, 3]
    assert O.nms(boxes, scores, 0.95).tolist() == [0, 1, 2, 3, 4]
    assert O.nms(boxes, scores, 0.0).tolist() == [0, 3]
    # ties in score: the lower index goes first
    assert O.nms(boxes[[1, 0]], np.array([0.5, 0.5], np.float32), 0.5).tolist() == [0]
    # labels: only equal labels suppress
    assert O.nms(boxes, scores, 0.5, labels=np.array([0, 1, 0, 2, 2])).tolist() == [0, 1, 2, 3]
    # a box suppressed by a dropped box survives
    chain = np.array([[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]], np.float32)
    assert O.nms(chain, np.array([0.9, 0.8, 0.7], np.float32), 0.4).tolist() == [0, 2]
    assert O.nms(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.5).shape == (0,)


@pytest.mark.parametrize("seed", range(6))
def test_nms_matches_bruteforce(seed):
    rng = np.random.default_rng(seed)
    k = int(rng.integers(1, 200))
    c = rng.uniform(0, 60, (k, 2))
    wh = rng.uniform(0, 30, (k, 2))                 # includes near-degenerate boxes
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    scores = np.round(rng.random(k), 1).astype(np.float32)      # many ties
    labels = rng.integers(0, 3, k)
    for thr in (0.3, 0.5, 0.7):
        assert np.array_equal(O.nms(boxes, scores, thr), O.nms_bruteforce(boxes, scores, thr))
        assert np.array_equal(O.nms(boxes, scores, thr, labels), O.nms_bruteforce(boxes, scores, thr, labels))


def test_postprocess_oracle_follows_reference_steps():
    g = load_golden("decode.npz")
    h, w = int(g["hw"][0]), int(g["hw"][1])
    s, l, b = O.postprocess_image(g["probs"][0], g["deltas"][0], g["anchors"], h, w, boxes=g["clipped"][0])
    over = np.nonzero(g["over"][0])[0]                          # model.py:325 mask from the reference
    keep = O.nms(g["clipped"][0][over], g["score"][0][over], 0.5)
    assert np.array_equal(s, g["score"][0][over][keep])
    assert np.array_equal(l, g["label"][0][over][keep])
    assert np.array_equal(b, g["clipped"][0][over][keep])
    assert (np.diff(s) <= 0).all()
    s2, _, _ = O.postprocess_image(g["probs"][0], g["deltas"][0], g["anchors"], h, w,
                                   boxes=g["clipped"][0], pre_nms_topk=10, max_det=3)
    assert len(s2) <= 3 and np.array_equal(s2, s[: len(s2)])


# ------------------------------------------------------------------ C ABI surface
def declared_symbols():
    text = open(os.path.join(ROOT, "include", "retinahead.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import retinahead
    from retinahead import _lib
    assert os.path.exists(retinahead.LIB_PATH), "build the library first (__graft_entry__.build())"
    lib = ctypes.CDLL(retinahead.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in retinahead.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert retinahead.load_library().rh_version() == 100


def test_static_nms_limit_matches_header_and_scratch_budget():
    """The entry point that allocates nothing: its stated maximum is the header's macro, and the
    workspace rh_nms needs at that size fits the library's fixed 3 MiB scratch array."""
    import re
    import retinahead
    from retinahead import _lib
    text = open(os.path.join(ROOT, "include", "retinahead.h")).read()
    assert int(re.search(r"#define RH_NMS_STATIC_MAX_BOXES (\d+)", text).group(1)) == _lib.NMS_STATIC_MAX_BOXES
    lib = retinahead.load_library()
    need = lib.rh_nms_workspace_bytes(_lib.NMS_STATIC_MAX_BOXES)
    assert 0 < need <= 3 << 20


def test_host_side_queries_without_gpu():
    """Entry points that touch no device: defaults, counts, workspace sizes, argument errors."""
    import retinahead
    from retinahead import _lib, functional as F
    lib = retinahead.load_library()
    cfg = F.make_anchor_cfg()
    for s, n in ((512, 49104), (608, 69354), (800, 120087), (1024, 196416)):
        assert F.anchor_count(cfg, s, s) == n == O.num_anchors(s, s)
    assert F.anchor_count(F.make_anchor_cfg(pyramid_levels=[3]), 64, 64) == 8 * 8 * 9
    p = _lib.FocalParams()
    lib.rh_focal_params_default(ctypes.byref(p))
    assert (p.alpha, p.neg_thresh, p.pos_thresh, p.batch_divisor) == (0.25, np.float32(0.4), 0.5, 0)
    assert lib.rh_focal_loss_workspace_bytes(16, 69354, 80, 8) > 16 * 69354 * 4
    assert lib.rh_focal_loss_workspace_bytes(0, 69354, 80, 8) == 0
    assert lib.rh_nms_workspace_bytes(2000) >= 2000 * 32 * 8
    # bad arguments are refused on the host, before any launch
    rc = lib.rh_focal_loss_fwd_bwd(None, None, None, None, 1, 100, 4, 2, ctypes.byref(p), None, None,
                                   None, None, None, None, None, 0, 7, None)
    assert rc == -1 and "null" in _lib.last_error()
    rc = lib.rh_anchors_f32(ctypes.byref(cfg), 64, 64, None, 5, None)
    assert rc == -2 and "needs 774" in _lib.last_error()
    with pytest.raises(retinahead.RetinaHeadError):
        retinahead.FocalLoss()(torch.zeros(1, 774, 4), torch.zeros(1, 774, 4), torch.zeros(1, 774, 4),
                               torch.zeros(1, 1, 5))          # CPU tensors: no fallback path
    with pytest.raises(retinahead.RetinaHeadError):
        retinahead.Anchors()(torch.zeros(1, 3, 64, 64))


def test_missing_library_fails_loudly(monkeypatch):
    from retinahead import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libretinahead.so")
    with pytest.raises(_lib.RetinaHeadError):
        _lib.load()


# ------------------------------------------------------------------ data-parallel helper (gloo, world 2)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, q, normalizer="per_image"):
    import torch.distributed as dist
    for p in (ROOT, PKG_DIR):
        if p not in sys.path:
            sys.path.insert(0, p)
    from retinahead import dist as rdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "focal_loss.npz"))
    cls, reg, anc, ann = (g["normal." + k] for k in ("cls", "reg", "anchors", "ann"))
    bsz = cls.shape[0]
    lo, hi = rdist.shard_bounds(bsz, rank, world)

    def local_loss(c, r, a, an):     # the oracle stands in for the HIP kernels on this GPU-less host
        kw = {}
        if normalizer == "global":   # the staged call of the module: assign, exchange, stream
            counts = [O.assign(a.numpy().reshape(-1, 4), row)["num_pos"] for row in an.numpy()]
            total = local_loss.cfg["reduce_total"](torch.tensor(counts, dtype=torch.int32))
            kw = dict(normalizer="global", num_pos_total=int(total.item()))
        o = O.focal_loss_fwd_bwd(c.detach().numpy(), r.detach().numpy(), a.numpy(), an.numpy(),
                                 batch_divisor=bsz, **kw)

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, c_, r_):
                return (torch.from_numpy(o["classification_loss"]).clone(),
                        torch.from_numpy(o["regression_loss"]).clone())

            @staticmethod
            def backward(ctx, gc, gr):
                return (torch.from_numpy(o["d_classifications"]) * gc,
                        torch.from_numpy(o["d_regressions"]) * gr)
        cl, rl = Fn.apply(c, r)
        return {"classification_loss": cl, "regression_loss": rl}

    c = torch.from_numpy(cls[lo:hi]).clone().requires_grad_(True)
    r = torch.from_numpy(reg[lo:hi]).clone().requires_grad_(True)
    local_loss.cfg = dict(normalizer=normalizer, reduce_total=None)
    mod = rdist.DataParallelFocalLoss(local_loss)
    assert (local_loss.cfg["reduce_total"] is not None) == (normalizer == "global")
    out = mod(c, r, torch.from_numpy(anc), torch.from_numpy(ann[lo:hi]))
    (out["classification_loss"] + out["regression_loss"]).sum().backward()
    q.put((rank, lo, hi, out["classification_loss"].item(), out["regression_loss"].item(),
           c.grad.numpy(), r.grad.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_data_parallel_equals_single_process(world):
    """world ranks (gloo), images sharded (2+1 and 1+1+1 of the fixture's three): every rank sees the
    single-process losses, and the ranks' gradients concatenate to the single-process gradients."""
    from retinahead import dist as rdist
    assert [rdist.shard_bounds(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [rdist.shard_bounds(16, r, 8) for r in (0, 7)] == [(0, 2), (14, 16)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = load_golden("focal_loss.npz")
    want_cls, want_reg = float(g["normal.cls_loss"][0]), float(g["normal.reg_loss"][0])
    for rank, lo, hi, cl, rl, gc, gr in res:
        assert np.isclose(cl, want_cls, rtol=1e-5) and np.isclose(rl, want_reg, rtol=1e-5)
    gscale = g["normal.g"]
    full_c = np.concatenate([t[5] for t in res]) * gscale[0]
    full_r = np.concatenate([t[6] for t in res]) * gscale[1]
    assert np.allclose(full_c, g["normal.d_cls"], rtol=1e-5, atol=1e-12)
    assert np.allclose(full_r, g["normal.d_reg"], rtol=1e-5, atol=1e-9)


def test_data_parallel_global_normaliser():
    """normalizer='global' over 2 ranks: one more all-reduce (the positive count) and the ranks'
    losses are shares; the result is the single-process batch-wide result.  (The reference has no
    such mode: the single-process side is the oracle's restatement, parity unpinned.)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q, "global")) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = load_golden("focal_loss.npz")
    cls, reg, anc, ann = (g["normal." + k] for k in ("cls", "reg", "anchors", "ann"))
    want = O.focal_loss_fwd_bwd(cls, reg, anc, ann, normalizer="global")
    for rank, lo, hi, cl, rl, gc, gr in res:
        assert np.isclose(cl, want["classification_loss"][0], rtol=1e-5)
        assert np.isclose(rl, want["regression_loss"][0], rtol=1e-5)
    assert np.allclose(np.concatenate([t[5] for t in res]), want["d_classifications"], rtol=1e-5,
                       atol=1e-12)
    assert np.allclose(np.concatenate([t[6] for t in res]), want["d_regressions"], rtol=1e-5, atol=1e-9)


def _bench_worker(rank, world, port, q):
    """bench.py's own multi-rank skeleton (exchange, fences, rank count, per-block max over ranks)
    under gloo, with a stand-in for the GPU step: each rank's 'launch' leaves (rank + 1) * its step
    count in the four-float exchange vector."""
    import time
    import torch.distributed as dist
    for p in (ROOT, PKG_DIR):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = bench.RankExchange(world, "gloo")
    calls = [0]
    syncs = [0]

    def step():
        calls[0] += 1
        if rank == 1:
            time.sleep(0.002)                      # the slow rank sets every block's time
        ex.submit(torch.full((4,), float((rank + 1) * calls[0])))

    def sync():
        syncs[0] += 1
    seen = bench.count_ranks(world, "gloo", torch.device("cpu"))
    blocks = bench.timed_blocks(step, ex, sync, steps=5, warmup=2, repeats=3, world=world,
                                backend="gloo", dev=torch.device("cpu"))
    q.put((rank, seen, calls[0], syncs[0], blocks.tolist(), ex.last.tolist(), len(ex.pending)))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_multi_rank_skeleton_under_gloo():
    """What `bench.py --gpus N` does around its step, run by two CPU ranks: both ranks are counted,
    every step's vector is summed over the ranks, each timed block is bracketed by fences (two
    synchronisations each) and reported as the slowest rank's time, nothing is left pending."""
    import bench
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, seen, calls, syncs, blocks, last, pending in res:
        assert seen == world and calls == 2 + 3 * 5 and pending == 0
        assert syncs == 2 * (1 + 3)                       # the warm-up fence and one per block
        assert last == [float(3 * calls)] * 4             # (1 + 2) * the step count, on every rank
        assert len(blocks) == 3 and min(blocks) >= 5 * 0.002
    assert res[0][4] == res[1][4]                         # per block, the slowest rank, on all ranks
    # a scaling run that is not what it says is refused
    bench.require_ranks(2, 2, 8, "nccl")
    with pytest.raises(SystemExit):
        bench.require_ranks(1, 2, 8, "nccl")
    with pytest.raises(SystemExit):
        bench.require_ranks(2, 2, 1, "nccl")
    bench.require_ranks(2, 2, 1, "gloo")                  # the one-card rehearsal is asked for by name
