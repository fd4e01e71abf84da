"""rh_nms_static, the entry point that allocates nothing: its keep lists must equal rh_nms's
(the entry point it is a variant of) bit for bit, on two calls in a row that share the
library's fixed scratch array, the second with a smaller input, and sizes beyond the stated
maximum must be refused.  The file name makes pytest collect it last, so that a failure here
hides no other test under `pytest -x`.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rh():
    import retinahead
    retinahead.load_library()
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return retinahead


def make_boxes(rng, k, extent):
    c = rng.uniform(0, extent, (k, 2))
    wh = rng.uniform(8, 90, (k, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    scores = rng.random(k).astype(np.float32)
    scores[k // 3] = scores[k // 2]                 # a tie in score: the lower index goes first
    labels = rng.integers(0, 5, k).astype(np.int32)
    return boxes, scores, labels


def run_static(rh, boxes, scores, labels, thr, keep, count):
    from retinahead import functional as F
    F.nms_static(boxes, scores, thr, keep, count, labels=labels)
    torch.cuda.synchronize()
    n = int(count.item())
    assert 0 <= n <= boxes.shape[0]
    return keep[:n].clone()


@pytest.mark.parametrize("per_class", [False, True])
def test_static_nms_matches_nms_on_consecutive_calls(rh, per_class):
    from retinahead import _lib
    from oracle import dense_head_oracle as O
    rng = np.random.default_rng(11 + per_class)
    big, small = _lib.NMS_STATIC_MAX_BOXES, 300     # the maximum first, then a much smaller input
    keep = torch.empty(big, dtype=torch.int64, device="cuda")
    count = torch.empty(1, dtype=torch.int32, device="cuda")
    for k, extent in ((big, 1500.0), (small, 400.0), (1, 10.0)):
        b_np, s_np, l_np = make_boxes(rng, k, extent)
        b, s = torch.from_numpy(b_np).cuda(), torch.from_numpy(s_np).cuda()
        l = torch.from_numpy(l_np).cuda() if per_class else None
        keep.fill_(-7)
        got = run_static(rh, b, s, l, 0.5, keep, count)
        want = rh.batched_nms(b, s, l, 0.5) if per_class else rh.nms(b, s, 0.5)
        assert torch.equal(got, want), k
        assert bool((keep[got.numel():k] == -7).all())          # nothing written past the count
        ref = O.nms(b_np, s_np, 0.5, labels=l_np if per_class else None)
        assert np.array_equal(got.cpu().numpy(), ref), k
        assert 0 < got.numel() < k or k == 1


def test_static_nms_limits(rh):
    from retinahead import _lib, functional as F
    k = _lib.NMS_STATIC_MAX_BOXES + 1
    b = torch.zeros((k, 4), device="cuda")
    s = torch.zeros((k,), device="cuda")
    keep = torch.empty(k, dtype=torch.int64, device="cuda")
    count = torch.full((1,), -5, dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.RetinaHeadError, match="exceeds"):
        F.nms_static(b, s, 0.5, keep, count)
    torch.cuda.synchronize()
    assert int(count.item()) == -5                               # refused before any launch
    F.nms_static(b[:0], s[:0], 0.5, keep, count)                 # no boxes: count 0
    torch.cuda.synchronize()
    assert int(count.item()) == 0


@pytest.mark.parametrize("k", [1000, 4096])
def test_static_nms_under_stream_capture(rh, k):
    """The reason the entry point exists: it can be captured into a hipGraph (after one call
    outside capture has looked up the scratch address) and replayed on new data; also at the
    entry point's limit, where the rank sort asks for more than the default share of LDS."""
    from retinahead import functional as F
    rng = np.random.default_rng(5)
    b_np, s_np, _ = make_boxes(rng, k, 600.0)
    b, s = torch.from_numpy(b_np).cuda(), torch.from_numpy(s_np).cuda()
    keep = torch.empty(k, dtype=torch.int64, device="cuda")
    count = torch.empty(1, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        F.nms_static(b, s, 0.5, keep, count)                     # warm-up outside capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        F.nms_static(b, s, 0.5, keep, count)
    for seed in (1, 2):
        b2, s2, _ = make_boxes(np.random.default_rng(seed), k, 600.0)
        b.copy_(torch.from_numpy(b2))
        s.copy_(torch.from_numpy(s2))
        graph.replay()
        torch.cuda.synchronize()
        n = int(count.item())
        assert torch.equal(keep[:n], rh.nms(b, s, 0.5))
