"""PostprocessPipeline: batches in flight on several streams give what the synchronous
``postprocess`` gives, in submission order, including a batch that overflows its candidate room."""
import numpy as np
import pytest
import torch

from test_gpu_parity import dev, infer_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rh():
    import retinahead
    retinahead.load_library()
    return retinahead


def same_detections(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        for k in ("scores", "labels", "boxes"):
            assert torch.equal(g[k], w[k]), k


@pytest.mark.parametrize("kw", [dict(), dict(mode="per_class", pre_nms_topk=300), dict(max_det=20)])
def test_pipeline_matches_the_synchronous_call(rh, kw):
    size, c = 192, 13
    batches = [infer_batch(20 + i, 2, size, c, mu=-4.5) for i in range(7)]
    anc = dev(batches[0][2])
    data = [(dev(b[0]), dev(b[1])) for b in batches]
    want = [rh.postprocess(cls, reg, anc, (size, size), **kw) for cls, reg in data]
    want = [[{k: v.clone() for k, v in img.items()} for img in batch] for batch in want]
    pipe = rh.PostprocessPipeline(depth=3, streams=2, **kw)
    got = []
    for cls, reg in data:
        if len(pipe) == 3:
            got.append([{k: v.clone() for k, v in img.items()} for img in pipe.results()])
        pipe.submit(cls, reg, anc, (size, size))
    while len(pipe):
        got.append([{k: v.clone() for k, v in img.items()} for img in pipe.results()])
    assert len(got) == len(want)
    for g, w in zip(got, want):
        same_detections(g, w)


def test_pipeline_overflow_falls_back_and_limits(rh):
    size, c = 128, 8
    cls, reg, anc = infer_batch(40, 2, size, c, mu=-3.0)          # thousands of candidates per image
    want = rh.postprocess(dev(cls), dev(reg), dev(anc), (size, size))
    pipe = rh.PostprocessPipeline(depth=2, max_candidates=64)
    pipe.submit(dev(cls), dev(reg), dev(anc), (size, size))
    same_detections(pipe.results(), want)
    with pytest.raises(RuntimeError):
        pipe.results()
    pipe.submit(dev(cls), dev(reg), dev(anc), (size, size))
    pipe.submit(dev(cls), dev(reg), dev(anc), (size, size))
    with pytest.raises(RuntimeError):
        pipe.submit(dev(cls), dev(reg), dev(anc), (size, size))
    with pytest.raises(TypeError):
        rh.PostprocessPipeline(bogus=1)


def test_pipeline_streams_are_its_own_and_are_handed_on(rh):
    """A pipeline's streams are created for pipelines, one after the other, outside torch's pool of
    32 (whose streams can share a hardware queue, after which batches no longer overlap); they are
    never destroyed -- the caching allocator remembers the stream of every block -- but handed to
    the next pipeline of the same width when theirs is gone."""
    import gc
    size, c = 128, 6
    cls, reg, anc = infer_batch(70, 2, size, c, mu=-4.0)
    want = rh.postprocess(dev(cls), dev(reg), dev(anc), (size, size))
    pool = {torch.cuda.Stream().cuda_stream for _ in range(40)}          # the whole pool, and around again
    pipe = rh.PostprocessPipeline(depth=4, streams=2)
    if not pipe._own.own:
        pytest.skip("the HIP runtime could not be reached from Python here: streams come from torch's pool")
    first = [s.cuda_stream for s in pipe.streams]
    assert len(set(first)) == 2 and not (set(first) & pool)
    for _ in range(6):
        pipe.submit(dev(cls), dev(reg), dev(anc), (size, size))
        same_detections(pipe.results(), want)
    del pipe
    gc.collect()
    again = rh.PostprocessPipeline(depth=4, streams=2)
    assert [s.cuda_stream for s in again.streams] == first
    other = rh.PostprocessPipeline(depth=3, streams=3)                   # another width: streams of its own
    assert not (set(s.cuda_stream for s in other.streams) & set(first))
    for p in (again, other):
        p.submit(dev(cls), dev(reg), dev(anc), (size, size))
    for p in (again, other):
        same_detections(p.results(), want)
