"""FocalLoss.prefetch_assignment: a step whose IoU assignment was carried by the previous step's
stream launch gives the results of the plain step bit for bit -- in eager calls, when the
announced tensor is not the one that arrives, and in two alternating captured graphs (the
benchmark's launch mode)."""
import numpy as np
import pytest
import torch

from test_gpu_parity import dev, make_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rh():
    import retinahead
    retinahead.load_library()
    return retinahead


def run_step(mod, cls, reg, anc, ann):
    cls = cls.detach().clone().requires_grad_(True)
    reg = reg.detach().clone().requires_grad_(True)
    out = mod(cls, reg, anc, ann)
    sum(out.values()).backward()
    return (out["classification_loss"].detach().clone(), out["regression_loss"].detach().clone(),
            cls.grad.clone(), reg.grad.clone(), mod.last["num_pos"].clone(), mod.last["status"].clone())


def same(a, b):
    return all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.fixture
def stages_seen(rh, monkeypatch):
    seen = []
    real = rh.functional.focal_loss_fwd_bwd

    def spy(*a, **kw):
        seen.append((kw.get("stages", rh._lib.STAGE_ALL), kw.get("ahead") is not None))
        return real(*a, **kw)
    monkeypatch.setattr(rh.functional, "focal_loss_fwd_bwd", spy)
    return seen


@pytest.mark.parametrize("normalizer,m", [("per_image", 6), ("global", 6), ("per_image", 200),
                                          ("per_image", 1100)])
def test_prefetched_assignment_gives_the_plain_step(rh, stages_seen, normalizer, m):
    """m = 1100: annotation lists long enough for the split form of the assignment kernel (four
    waves share a block's anchors), which rides on the stream launch like the plain form."""
    size, c = 160, 20
    batches = [make_batch(50 + i, 3, size, c, m, [m, 0, 2]) for i in range(4)]
    anc = dev(batches[0][2])
    data = [(dev(b[0]), dev(b[1]), dev(b[3])) for b in batches]
    plain = rh.FocalLoss(normalizer=normalizer)
    want = [run_step(plain, cls, reg, anc, ann) for cls, reg, ann in data]
    del stages_seen[:]

    mod = rh.FocalLoss(normalizer=normalizer)
    later = rh._lib.STAGE_STREAM | rh._lib.STAGE_FINALIZE
    for i, (cls, reg, ann) in enumerate(data):
        if i + 1 < len(data):
            mod.prefetch_assignment(anc, data[i + 1][2], c)      # carried by this step
        got = run_step(mod, cls, reg, anc, ann)
        assert same(got, want[i]), i
        assert stages_seen[-1] == (rh._lib.STAGE_ALL if i == 0 else later, i + 1 < len(data))
    st = mod.cfg["ahead"].by_device[anc.device.index]
    assert st["ready"] is None and st["asked"] is None and len(st["bufs"]) <= 3


def test_prefetch_that_does_not_match_is_ignored(rh):
    size, c = 128, 8
    b0, b1 = make_batch(60, 2, size, c, 4, [4, 1]), make_batch(61, 2, size, c, 4, [2, 3])
    anc = dev(b0[2])
    cls, reg, ann0, ann1 = dev(b0[0]), dev(b0[1]), dev(b0[3]), dev(b1[3])
    plain = rh.FocalLoss()
    want0, want1 = run_step(plain, cls, reg, anc, ann0), run_step(plain, cls, reg, anc, ann1)
    assert not same(want0, want1)
    mod = rh.FocalLoss()
    mod.prefetch_assignment(anc, ann1, c)
    assert same(run_step(mod, cls, reg, anc, ann0), want0)       # carries ann1's assignment
    assert same(run_step(mod, cls, reg, anc, ann0), want0)       # but ann0 arrives again
    mod.prefetch_assignment(anc, ann0, c)
    assert same(run_step(mod, cls, reg, anc, ann1), want1)       # carries ann0's
    ann0.copy_(ann1)                                             # modified after that
    assert same(run_step(mod, cls, reg, anc, ann0), want1)
    # an out-of-range label is still reported by the step that owns the annotations
    bad = ann1.clone()
    bad[0, 0, 4] = c
    chk = rh.FocalLoss(check_labels=True)
    chk.prefetch_assignment(anc, bad, c)
    chk(cls, reg, anc, ann1)                                     # carries the bad batch: no error yet
    with pytest.raises(IndexError):
        chk(cls, reg, anc, bad)
    # a batch of another size is not carried
    mod.prefetch_assignment(anc, ann1[:1], c)
    assert same(run_step(mod, cls, reg, anc, ann1), want1)
    assert mod.cfg["ahead"].by_device[anc.device.index]["ready"] is None


def test_prefetch_in_two_alternating_graphs(rh):
    """Graph A runs step i on workspace X and assigns step i+1 into Y; graph B the reverse.  Replayed A, B, A, ... on changing data they give the plain results."""
    size, c = 128, 20
    batches = [make_batch(70 + i, 2, size, c, 5, [5, 2]) for i in range(6)]
    anc = dev(batches[0][2])
    s_cls = dev(batches[0][0]).requires_grad_(True)
    s_reg = dev(batches[0][1]).requires_grad_(True)
    # step i reads s_ann[i % 2]; while it runs, the other buffer holds step i+1's annotations
    s_ann = [dev(batches[0][3]), dev(batches[1][3])]
    mod = rh.FocalLoss()

    def step(i):
        s_cls.grad = None
        s_reg.grad = None
        mod.prefetch_assignment(anc, s_ann[(i + 1) % 2], c)
        out = mod(s_cls, s_reg, anc, s_ann[i % 2])
        sum(out.values()).backward()
        return out["classification_loss"], out["regression_loss"], s_cls.grad, s_reg.grad

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(2):
            step(i)              # leaves the assignment of s_ann[0] done, in its workspace
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graphs, outs = [], []
    for i in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            outs.append(step(i))
        graphs.append(g)
    plain = rh.FocalLoss()
    # the assignment of batch 0 must be in place before the first replay: it was issued by the
    # eager steps above for s_ann[0], whose content is still batch 0's
    for i in range(len(batches) - 1):
        s_cls.data.copy_(dev(batches[i][0]))
        s_reg.data.copy_(dev(batches[i][1]))
        s_ann[(i + 1) % 2].copy_(dev(batches[i + 1][3]))
        graphs[i % 2].replay()
        torch.cuda.synchronize()
        want = run_step(plain, dev(batches[i][0]), dev(batches[i][1]), anc, dev(batches[i][3]))
        got = outs[i % 2]
        for k in range(4):
            assert torch.equal(got[k], want[k]), (i, k)


def test_captured_prefetch_buffers_outlive_eager_steps_of_other_shapes(rh):
    """Two graphs are captured as above; then more than three eager steps with growing annotation
    lists make the module allocate new workspaces and rotate its old ones out, the loss tensors
    the captures returned are dropped, and the allocator is asked to give freed blocks back.  The
    graphs' workspaces and small outputs must have stayed alive: replaying them still gives the
    plain results."""
    import gc
    size, c = 128, 20
    batches = [make_batch(90 + i, 2, size, c, 5, [5, 2]) for i in range(5)]
    anc = dev(batches[0][2])
    s_cls = dev(batches[0][0]).requires_grad_(True)
    s_reg = dev(batches[0][1]).requires_grad_(True)
    s_ann = [dev(batches[0][3]), dev(batches[1][3])]
    mod = rh.FocalLoss()

    def step(i):
        s_cls.grad = None
        s_reg.grad = None
        mod.prefetch_assignment(anc, s_ann[(i + 1) % 2], c)
        out = mod(s_cls, s_reg, anc, s_ann[i % 2])
        sum(out.values()).backward()
        return s_cls.grad, s_reg.grad       # the loss tensors are dropped at once

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(2):
            step(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graphs, grads = [], []
    for i in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            grads.append(step(i))
        graphs.append(g)
    st = mod.cfg["ahead"].by_device[anc.device.index]
    assert len(st["pinned"]) == 2
    pinned_ptrs = sorted(b.data_ptr() for b in st["pinned"])
    # eager steps of other shapes: every one needs a larger workspace than any the module holds
    for k, m in enumerate((40, 90, 150, 220)):
        cls, reg, _, ann = make_batch(95 + k, 2, size, c, m, [m, 1])
        nxt = dev(make_batch(99 + k, 2, size, c, m + 20, [3, 3])[3])
        mod.prefetch_assignment(anc, nxt, c)
        run_step(mod, dev(cls), dev(reg), anc, dev(ann))
    assert sorted(b.data_ptr() for b in st["pinned"]) == pinned_ptrs
    assert all(any(b is q for q in st["bufs"]) for b in st["pinned"])
    gc.collect()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]   # reuse freed blocks
    plain = rh.FocalLoss()
    mod.cfg["ahead"].by_device[anc.device.index]["ready"] = None
    # batch 0's assignment has to be in the workspace graph 0 reads: graph 1 writes it
    s_ann[0].copy_(dev(batches[0][3]))
    graphs[1].replay()
    for i in range(len(batches) - 1):
        s_cls.data.copy_(dev(batches[i][0]))
        s_reg.data.copy_(dev(batches[i][1]))
        s_ann[(i + 1) % 2].copy_(dev(batches[i + 1][3]))
        graphs[i % 2].replay()
        torch.cuda.synchronize()
        want = run_step(plain, dev(batches[i][0]), dev(batches[i][1]), anc, dev(batches[i][3]))
        assert torch.equal(grads[i % 2][0], want[2]) and torch.equal(grads[i % 2][1], want[3]), i
    del junk


def test_annotations_uploaded_on_a_side_stream(rh):
    """A loader uploads the next batch's annotations on a stream of its own (pinned memory,
    non_blocking) while the current step runs; ordered before the step that carries their
    assignment by wait_stream, as the docstring of prefetch_assignment asks, the announced steps
    give the plain results."""
    size, c, m = 160, 20, 6
    batches = [make_batch(40 + i, 2, size, c, m, [m, 3]) for i in range(4)]
    anc = dev(batches[0][2])
    plain = rh.FocalLoss()
    want = [run_step(plain, dev(b[0]), dev(b[1]), anc, dev(b[3])) for b in batches]
    loader = torch.cuda.Stream()
    host = [torch.from_numpy(b[3]).pin_memory() for b in batches]
    slots = [torch.empty_like(dev(batches[0][3])) for _ in range(2)]

    def upload(i):
        loader.wait_stream(torch.cuda.current_stream())      # the slot's previous reader is done
        with torch.cuda.stream(loader):
            slots[i % 2].copy_(host[i], non_blocking=True)

    mod = rh.FocalLoss()
    upload(0)
    for i in range(len(batches)):
        if i + 1 < len(batches):
            upload(i + 1)
        torch.cuda.current_stream().wait_stream(loader)       # both uploads precede this step's launch
        if i + 1 < len(batches):
            mod.prefetch_assignment(anc, slots[(i + 1) % 2], c)
        got = run_step(mod, dev(batches[i][0]), dev(batches[i][1]), anc, slots[i % 2])
        assert same(got, want[i]), i


def test_assignment_stage_on_its_own_stream(rh):
    """functional.focal_assign: the assignment stage issued apart (here on a second stream, ordered
    by events) into a workspace of its own; the remaining stages then give the plain call's bits."""
    F, L = rh.functional, rh._lib
    cls, reg, anc, ann = make_batch(80, 3, 160, 20, 6, [6, 0, 2])
    d_cls, d_reg, d_anc, d_ann = dev(cls), dev(reg), dev(anc), dev(ann)
    want = F.focal_loss_fwd_bwd(d_cls, d_reg, d_anc, d_ann)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out, ws = F.focal_assign(d_anc, d_ann, 20)
    torch.cuda.current_stream().wait_stream(side)
    got = F.focal_loss_fwd_bwd(d_cls, d_reg, d_anc, d_ann, stages=L.STAGE_STREAM | L.STAGE_FINALIZE,
                               out=out, ws=ws)
    for k in ("losses", "per_image", "num_pos", "status", "d_classifications", "d_regressions"):
        assert torch.equal(got[k], want[k]), k
    with pytest.raises(rh.RetinaHeadError):
        F.focal_loss_fwd_bwd(d_cls, d_reg, d_anc, d_ann, ws=ws[:16])
