"""Several batches of inference post-processing in flight at once.

One batch of models/retinanet/model.py:320-340 is four dependent kernels: a score scan bound by
HBM, then a sort, a pair search and a resolve step bound by latency.  Issued one batch after the
other on one stream the GPU idles through the latency-bound part; with a few batches in flight on
streams of their own the next batch's scan fills it (608 px, 80 classes, 16 images: 98 -> 68 us
per batch).  ``PostprocessPipeline`` keeps that bookkeeping -- streams, per-stream outputs and
workspaces, the one host read per batch -- behind two calls.
"""
import collections
import collections.abc
import ctypes as C

import torch

from . import _lib
from . import functional as F
from . import modules


def _hip_runtime():
    """The HIP runtime this process already has loaded (torch's), or None."""
    try:
        with open("/proc/self/maps") as f:
            paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        return None
    if len(paths) != 1:         # none, or two runtimes mapped: not ours to choose between
        return None
    try:
        return C.CDLL(paths.pop())
    except OSError:
        return None


_idle_stream_sets = {}      # (device index, n) -> stream sets whose pipeline is gone


class _OwnStreams:
    """``n`` HIP streams created one after the other for pipelines alone.  Streams drawn from torch's
    pool can share one of HIP's four hardware queues per process -- the pool holds 32 and hands them out
    in turn, so a process that has used a few dozen streams gives a pipeline a set of which two sit on
    one queue, and their batches no longer overlap (80-93 instead of 65 us per batch; streams created in
    a row get queues in a row).  The streams are never destroyed (the caching allocator keeps the
    handle of the stream a block was allocated on): a pipeline that goes hands its set to the next one.
    Falls back to the pool where the runtime cannot be reached."""

    def __init__(self, n, device):
        self.key = (device.index, n)
        idle = _idle_stream_sets.get(self.key)
        if idle:
            self.streams, self.own = idle.pop(), True
            return
        self.streams, self.own = [], True
        hip = _hip_runtime()
        try:
            if hip is None:
                raise OSError("HIP runtime not found")
            with torch.cuda.device(device):
                for _ in range(n):
                    h = C.c_void_p()
                    if hip.hipStreamCreateWithFlags(C.byref(h), 1) != 0 or not h.value:    # 1: non-blocking
                        raise OSError("hipStreamCreateWithFlags failed")
                    self.streams.append(torch.cuda.ExternalStream(h.value, device=device))
        except (OSError, AttributeError, RuntimeError):
            # (streams already created stay unused; nothing is destroyed)
            self.streams, self.own = [torch.cuda.Stream(device=device) for _ in range(n)], False

    def release(self):
        if self.own and self.streams:
            try:
                for st in self.streams:
                    st.synchronize()
                _idle_stream_sets.setdefault(self.key, []).append(self.streams)
            except Exception:       # interpreter shutdown
                pass
        self.streams = []

    def __del__(self):
        self.release()


class Detections(collections.abc.Sequence):
    """A batch's detections, ``[{'boxes','labels','scores'}]`` per image like the list
    ``retinahead.postprocess`` returns, with the per-image views made when asked for (slicing 48
    tensors costs the host more than the GPU spends on the whole batch)."""

    def __init__(self, out, kept):
        self.out, self.kept = out, kept

    def __len__(self):
        return len(self.kept)

    def __getitem__(self, b):
        if isinstance(b, slice):
            return [self[i] for i in range(*b.indices(len(self)))]
        if b < 0:
            b += len(self)
        k = self.kept[b]         # IndexError past the batch, as a list gives
        return {"boxes": self.out["boxes"][b, :k], "labels": self.out["labels"][b, :k],
                "scores": self.out["scores"][b, :k]}


class PostprocessPipeline:
    """``submit`` enqueues a batch and returns at once; ``results`` hands back, in submission order,
    what ``retinahead.postprocess`` returns for it (``[{'boxes','labels','scores'}]`` per image,
    as a sequence that makes each image's views on access).

    depth: batches that may be in flight (output buffers); streams: HIP streams they are spread
    over (created for the pipeline, outside torch's stream pool).  Three streams and the default stream have a hardware queue each with HIP's four queues
    per process: a fourth stream gains 2 % when the queues fall well and loses 5-10 % when two
    streams share one, which is not under the caller's control.  Twice as many slots keep the
    streams fed while the host waits for the oldest batch (608 px, 80 classes, 16 images:
    79 us per batch with three slots, 67 with six (an occasional run 74); 92 one batch at a time).
    The remaining keyword arguments are those of ``retinahead.postprocess`` (``score_thresh``,
    ``iou_thresh``, ``mode``, ``pre_nms_topk``, ``max_det``, ``max_candidates``, ``from_logits``).

    A submitted batch's input tensors must stay unmodified until its results have been taken (the
    kernels read them where they lie).  They must have been produced on the stream that is current
    at ``submit`` (the slot's stream waits for that one, unless it is idle): work queued on any
    other stream has to be ordered before the call by the caller.  The outputs returned for a batch are views of buffers
    that are reused ``depth`` submissions later: copy them if they are kept longer.  ``len(pipe)``
    is the number of batches submitted and not yet taken.
    """

    def __init__(self, depth=6, streams=3, device=None, **kw):
        if depth < 1 or streams < 1:
            raise ValueError("depth and streams must be at least 1")
        streams = min(streams, depth)
        self.kw = dict(score_thresh=0.05, iou_thresh=0.5, mode="agnostic", pre_nms_topk=None,
                       max_det=None, max_candidates=4096, from_logits=False)
        unknown = set(kw) - set(self.kw)
        if unknown:
            raise TypeError(f"unknown arguments {sorted(unknown)}")
        self.kw.update(kw)
        if self.kw["mode"] not in ("agnostic", "per_class"):
            raise ValueError("mode must be 'agnostic' or 'per_class'")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._own = _OwnStreams(streams, self.device)
        self.streams = self._own.streams
        self.slots = [None] * depth        # per slot: (shape key, device outputs, pinned counts)
        self.pending = collections.deque()
        self.turn = 0

    def __len__(self):
        return len(self.pending)

    def submit(self, classification, regression, anchors, img_batch):
        """Enqueues one batch (same inputs as ``retinahead.postprocess``, flat tensors)."""
        if len(self.pending) == len(self.slots):
            raise RuntimeError("every slot is in flight: take results() first")
        if hasattr(img_batch, "shape"):
            height, width = int(img_batch.shape[2]), int(img_batch.shape[3])
        else:
            height, width = int(img_batch[0]), int(img_batch[1])
        k = self.turn % len(self.slots)
        self.turn += 1
        st = self.streams[k % len(self.streams)]     # (slots of one stream share its workspace, in order)
        kw = self.kw
        n = classification.shape[1]
        cap = min(n, int(kw["max_candidates"]))
        key = (tuple(classification.shape), cap, kw["pre_nms_topk"], kw["max_det"])
        out, host_counts = self.slots[k][1:] if self.slots[k] is not None and self.slots[k][0] == key else (None, None)
        # the producer of the inputs works on the caller's stream; when that stream is idle (a
        # host query) the wait is left out: a wait is a barrier packet in the slot's stream, and
        # those cost the pipeline more than the query costs the host
        cur = torch.cuda.current_stream(self.device)
        if not cur.query():
            st.wait_stream(cur)
        fast = (out is not None and "call" in out and classification.dtype is torch.float32
                and regression.dtype is torch.float32 and classification.is_contiguous()
                and regression.is_contiguous() and classification.device == self.device
                and regression.device == self.device and anchors.data_ptr() == out["call"][0]
                and torch.cuda.current_device() == self.device.index   # the C call launches on it
                and (height, width) == out["call"][1])
        if fast:
            # The slot has served this shape before: same checks, same buffers -- one C call.  Its
            # three count rows go straight to the slot's pinned host buffer (the resolve kernel
            # writes them; pinned memory is mapped into the device's address space), so no copy
            # has to be queued either.
            lib, args, done = out["call"][2:5]
            _lib.check(lib.rh_postprocess(classification.data_ptr(), regression.data_ptr(), *args,
                                          st.cuda_stream), "rh_postprocess")
            done.record(st)
        else:
            with torch.cuda.stream(st):
                out = F.postprocess_raw(classification, regression, anchors, height, width,
                                        kw["score_thresh"], kw["iou_thresh"], kw["mode"] == "per_class",
                                        kw["pre_nms_topk"] or 0, kw["max_det"] or 0, cap, out=out,
                                        from_logits=kw["from_logits"])
                if host_counts is None:      # pinned, so that the copy does not block the host
                    host_counts = torch.empty(out["counts"].shape, dtype=out["counts"].dtype,
                                              pin_memory=True)
                host_counts.copy_(out["counts"], non_blocking=True)
                done = torch.cuda.Event()
                done.record(st)
            self._remember_call(out, classification, anchors, height, width, cap, st, host_counts)
        self.slots[k] = (key, out, host_counts)
        self.pending.append((out, host_counts, done, (classification, regression, anchors, (height, width))))

    def _remember_call(self, out, classification, anchors, height, width, cap, st, host_counts):
        """Everything of the C call that does not change while a slot keeps serving one shape:
        later submissions of that shape skip the Python-side checks and make one ctypes call."""
        kw = self.kw
        anc = F._anchors2d(anchors)
        if anc.dtype is not torch.float32 or not anc.is_contiguous():
            return
        bsz, n, c = classification.shape
        params = _lib.PostParams(float(height), float(width), float(kw["score_thresh"]),
                                 float(kw["iou_thresh"]), int(kw["mode"] == "per_class"),
                                 int(kw["pre_nms_topk"] or 0), int(kw["max_det"] or 0), cap,
                                 int(bool(kw["from_logits"])))
        ws = F.workspace(self.device, out["workspace_bytes"], "post", st.cuda_stream)
        args = (anc.data_ptr(), bsz, n, c, C.byref(params), out["scores"].data_ptr(),
                out["labels"].data_ptr(), out["boxes"].data_ptr(), host_counts[0].data_ptr(),
                host_counts[1].data_ptr(), host_counts[2].data_ptr(), ws.data_ptr(), ws.numel(),
                int(_lib.POST_ALL))
        # (params, ws and anc ride along so that the pointers in args stay valid; one event per
        # slot is enough: a slot is reused only after its previous batch has been taken)
        out["call"] = (anchors.data_ptr(), (height, width), _lib.load(), args, torch.cuda.Event(),
                       (params, ws, anc))

    def results(self):
        """The oldest submitted batch's detections (waits for that batch only)."""
        if not self.pending:
            raise RuntimeError("nothing has been submitted")
        out, counts, done, inputs = self.pending.popleft()
        done.synchronize()
        if int(counts[2].max()) > 0:
            # more candidates than max_candidates holds: the synchronous call repeats with room
            cls, reg, anc, shape = inputs
            kw = self.kw
            return modules.postprocess(cls, reg, anc, shape, kw["score_thresh"], kw["iou_thresh"],
                                       kw["mode"], kw["pre_nms_topk"], kw["max_det"],
                                       kw["max_candidates"], from_logits=kw["from_logits"])
        kept = counts[0].tolist()
        if min(kept) < 0:
            raise _lib.RetinaHeadError("NMS resolve gave up (internal error)")
        return Detections(out, kept)
