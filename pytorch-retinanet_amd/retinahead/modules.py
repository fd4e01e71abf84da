"""Drop-in operators with the reference's call signatures.

Each class/function names the reference symbol it replaces; replacing the four
attribute assignments at models/retinanet/model.py:231-237 and the ``nms``
import at model.py:8 with these is the whole integration.
"""
from __future__ import annotations

import threading

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import functional as F
from ._lib import RetinaHeadError


def calc_iou(a, b):
    """[N,M] IoU of a[N,4] and b[M,4]; replaces calc_iou (models/retinanet/losses.py:6-23)."""
    return F.calc_iou(a, b)


class _LossTensor(torch.Tensor):
    """A [1] loss tensor that knows its partner.  ``train.py:185`` adds the two losses
    (``sum(loss_dict.values())``: ``0 + classification_loss + regression_loss``); the finalize
    kernel has formed that float32 sum already, so ``0 + loss`` returns the loss itself and
    ``classification_loss + regression_loss`` returns the stored total, a third output of the
    same autograd node: two one-element launches less per step, same values, same gradients.
    ``loss.backward()`` without an explicit gradient on one of these tensors (train.py:190 when
    ``iterations`` is 1) seeds autograd with a cached one-element tensor of ones instead of
    filling a new one, and the node recognises it: the stored gradients were computed for an
    upstream gradient of exactly that, so nothing is left to launch in the backward pass.
    Every other operation behaves as on a plain tensor and returns plain tensors."""

    _token = None      # shared by the two losses of one forward call (and by nothing else)
    _total = None      # their sum as the finalize kernel stored it

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is _BACKWARD and len(args) == 1 and kwargs.get("gradient") is None \
                and args[0]._token is not None:
            with torch._C.DisableTorchFunctionSubclass():
                return torch.autograd.backward(
                    args[0], grad_tensors=_unit_gradient(args[0].device),
                    retain_graph=kwargs.get("retain_graph"),
                    create_graph=kwargs.get("create_graph", False), inputs=kwargs.get("inputs"))
        if not kwargs and len(args) == 2 and func in _ADD_FUNCS:
            a, b = args
            if isinstance(a, _LossTensor) and _is_zero(b):
                return a
            if isinstance(b, _LossTensor) and _is_zero(a):
                return b
            if isinstance(a, _LossTensor) and isinstance(b, _LossTensor) and a is not b \
                    and a._total is not None and a._token is b._token:
                # handed out once: train.py:187 divides the sum in place, and a second sum of
                # the same dict must not see that
                total, a._total, b._total = a._total, None, None
                return total
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        return out.as_subclass(torch.Tensor) if isinstance(out, _LossTensor) else out


_ADD_FUNCS = (torch.Tensor.add, torch.Tensor.__add__, torch.Tensor.__radd__, torch.add)
_BACKWARD = torch.Tensor.backward
_UNIT = {}      # device index -> float32 [1] tensor of ones that only this module holds
_UNIT_LOCK = threading.Lock()


def _unit_gradient(dev):
    unit = _UNIT.get(dev.index)
    if unit is None:
        with _UNIT_LOCK:        # replicas on several devices start from several threads
            unit = _UNIT.get(dev.index)
            if unit is None:
                unit = _UNIT[dev.index] = torch.ones(1, dtype=torch.float32, device=dev)
    return unit


class _CallState:
    """Small outputs of a module's latest call.  nn.DataParallel (the reference's multi-GPU path,
    train.py:137) replicates a module by copying its attribute dictionary, so every replica thread
    shares this object: results are kept per thread (what ``check_labels`` reads right after its
    own call) and per device (what the caller of the wrapped module finds afterwards)."""

    def __init__(self):
        self._tls = threading.local()
        self.by_device = {}

    def __reduce__(self):       # copies and pickles of a module start with an empty record
        return (_CallState, ())

    def set(self, dev, out):
        self._tls.last = out
        self.by_device[dev.index] = out

    @property
    def mine(self):
        """This thread's latest call, else the latest call on any device."""
        out = getattr(self._tls, "last", None)
        if out is None and self.by_device:
            out = next(reversed(self.by_device.values()))
        return out


class _AssignAhead:
    """Assignments asked for ahead of their step (``FocalLoss.prefetch_assignment``).  Per device:
    the request not yet carried out (it rides on the next ``forward``'s stream launch), the
    assignment carried out and waiting for its step, and up to three workspaces that take turns.
    Shared by the replicas nn.DataParallel makes; each works on its own device."""

    def __init__(self):
        self.by_device = {}
        self.lock = threading.Lock()

    def __reduce__(self):
        return (_AssignAhead, ())

    def _state(self, dev):
        st = self.by_device.get(dev.index)
        if st is None:
            with self.lock:
                st = self.by_device.setdefault(dev.index, {"bufs": [], "asked": None, "ready": None})
        return st

    @staticmethod
    def key(anchors, annotations, num_classes, cfg):
        return (annotations.data_ptr(), annotations._version, tuple(annotations.shape),
                anchors.data_ptr(), tuple(anchors.shape), int(num_classes), cfg["neg_thresh"],
                cfg["pos_thresh"], cfg["normalizer"])

    def ask(self, anchors, annotations, num_classes, cfg):
        # the tensor is kept alive so that its address cannot come to name another one
        self._state(annotations.device)["asked"] = {
            "key": self.key(anchors, annotations, num_classes, cfg), "ann": annotations}

    def take(self, anchors, annotations, num_classes, cfg):
        """The finished assignment of this step, if the previous step carried it."""
        st = self.by_device.get(annotations.device.index)
        if st is None:
            return None
        hit, st["ready"] = st["ready"], None
        if hit is not None and hit["key"] == self.key(anchors, annotations, num_classes, cfg):
            return hit
        return None

    def next_for(self, cls, anchors, using):
        """What the step now starting carries for the next one: (annotations, workspace, small
        outputs), or None.  ``using``: the workspace this step itself reads, if it is one of ours.
        Under stream capture the workspace and the small outputs handed out here are pinned: a
        replay of the graph writes to their addresses, so they are neither dropped from the
        rotation nor freed while the process lives."""
        st = self.by_device.get(cls.device.index)
        asked = st and st["asked"]
        if not asked:
            return None
        st["asked"] = None
        key, ann = asked["key"], asked["ann"]
        if (ann.dim() != 3 or ann.shape[0] != cls.shape[0] or key[5] != cls.shape[-1]
                or key[3] != anchors.data_ptr() or ann._version != key[1]):
            return None                 # another batch size, class count or anchor set: not ours
        lib = _lib.load()
        need = F._focal_ws_bytes(lib, cls.shape[0], cls.shape[1], cls.shape[2], ann.shape[1])
        pinned = st.setdefault("pinned", [])
        ws = next((b for b in st["bufs"] if b is not using and b.numel() >= need), None)
        if ws is None:
            ws = F.focal_workspace(cls.shape[0], cls.shape[1], cls.shape[2], ann.shape[1], cls.device)
            st["bufs"].append(ws)
            # three eager workspaces take turns; the ones a captured graph addresses stay
            old, st["bufs"] = st["bufs"][:-3], st["bufs"][-3:]
            st["bufs"][:0] = [b for b in old if any(b is q for q in pinned)]
        out = F.new_small_outputs(cls.shape[0], cls.device)
        if torch.cuda.is_current_stream_capturing():
            self.pin(st, ws, out)
        st["ready"] = {"key": key, "ann": ann, "ws": ws, "out": out}
        return ann, ws, out

    @staticmethod
    def pin(st, ws, out):
        """Keeps a workspace and a step's small outputs alive for the graphs that captured them."""
        pinned = st.setdefault("pinned", [])
        if not any(ws is q for q in pinned):
            pinned.append(ws)
        F.keep_for_graphs(ws, dict.__getitem__(out, "small"))


def _is_unit(g):
    """True when ``g`` is known on the host to hold exactly 1.0: it is the cached seed."""
    unit = _UNIT.get(g.device.index) if g is not None else None
    return unit is not None and g.numel() == 1 and g.data_ptr() == unit.data_ptr()


def _is_zero(x):
    return isinstance(x, (int, float)) and not isinstance(x, bool) and x == 0


def _as_loss_dict(cls_loss, reg_loss, total):
    # no reference from one loss to the other (that would be a cycle through the autograd node,
    # which owns the gradient buffers, and only the cyclic collector would ever free them)
    c, r = cls_loss.as_subclass(_LossTensor), reg_loss.as_subclass(_LossTensor)
    total = total.as_subclass(_LossTensor)
    c._token = r._token = total._token = object()
    c._total = r._total = total
    return {"classification_loss": c, "regression_loss": r}


def _loss_outputs(out):
    """The three [1] outputs of the autograd node on the call's small buffer.  They share its
    storage without being views of it: autograd refuses in-place operations on views returned by
    a node with several outputs, and train.py:187 (``loss /= iterations``) is one."""
    small = out["small"]     # [classification loss, regression loss, positives, images, total | ...]
    store = small.untyped_storage()
    at = small.storage_offset()
    return (small.new_empty(0).set_(store, at, (1,)), small.new_empty(0).set_(store, at + 1, (1,)),
            small.new_empty(0).set_(store, at + 4, (1,)))


def _stored_as_is(ctx, g_cls, g_reg, g_total):
    """The stored gradients are final: they were computed for an upstream gradient of 1 on both
    losses, and the only gradient that arrived is the cached seed of ones on their sum."""
    return ctx.expected == 1.0 and g_cls is None and g_reg is None and _is_unit(g_total)


def _upstream(g_part, g_total):
    """Gradient reaching one loss: directly, through the stored total, or both."""
    if g_part is None:
        return g_total
    return g_part if g_total is None else g_part + g_total


def _run_fused(fn, tensors, cfg):
    """Calls a fused-loss wrapper with the module's settings.  With the batch-wide normaliser
    and a ``reduce_total`` hook (data-parallel use) the assignment stage runs first, the hook
    turns the shard's positive counts into the count over all ranks, and the remaining stages
    normalise by that."""
    kw = dict(alpha=cfg["alpha"], neg_thresh=cfg["neg_thresh"], pos_thresh=cfg["pos_thresh"],
              grad_cls=cfg["expected_grad"], grad_reg=cfg["expected_grad"],
              batch_divisor=cfg["batch_divisor"], from_logits=cfg["from_logits"],
              normalizer=cfg["normalizer"])
    hook = cfg.get("reduce_total")
    if cfg["normalizer"] == "global" and hook is not None:
        out = fn(*tensors, stages=_lib.STAGE_ASSIGN, **kw)
        total = hook(out["num_pos"])
        return fn(*tensors, stages=_lib.STAGE_STREAM | _lib.STAGE_FINALIZE, out=out,
                  num_pos_total=total, **kw)
    ahead = cfg.get("ahead")
    if ahead is not None and fn is F.focal_loss_fwd_bwd and ahead.by_device:
        cls, _, anchors, annotations = tensors
        hit = ahead.take(anchors, annotations, cls.shape[-1], cfg)
        nxt = ahead.next_for(cls, anchors, hit["ws"] if hit else None)
        if hit is not None:     # this step's assignment came with the previous step
            if torch.cuda.is_current_stream_capturing():
                # made by an eager (warm-up) step, read and written by every replay from now on
                ahead.pin(ahead.by_device[cls.device.index], hit["ws"], hit["out"])
            return fn(*tensors, stages=_lib.STAGE_STREAM | _lib.STAGE_FINALIZE, out=hit["out"],
                      ws=hit["ws"], ahead=nxt, **kw)
        return fn(*tensors, ahead=nxt, **kw)
    return fn(*tensors, **kw)


class _DenseHeadLossFn(torch.autograd.Function):
    """Forward launches the fused forward+backward kernels and keeps the gradients;
    backward applies the upstream gradients to them in place (one launch, no host read)."""

    @staticmethod
    def forward(ctx, classifications, regressions, anchors, annotations, cfg):
        out = _run_fused(F.focal_loss_fwd_bwd,
                         (classifications, regressions, anchors, annotations), cfg)
        ctx.d_cls = out["d_classifications"]
        ctx.d_reg = out["d_regressions"]
        ctx.expected = cfg["expected_grad"]
        ctx.in_dtypes = (classifications.dtype, regressions.dtype)
        # keep only the small outputs: a second reference to the gradient buffers would make
        # autograd's AccumulateGrad copy them (hundreds of MB) instead of adopting them
        cfg["state"].set(classifications.device, out.small_only())
        ctx.set_materialize_grads(False)
        return _loss_outputs(out)       # fresh storage every call: no copy needed

    @staticmethod
    def backward(ctx, g_cls, g_reg, g_total):
        as_stored = _stored_as_is(ctx, g_cls, g_reg, g_total)
        g_cls, g_reg = _upstream(g_cls, g_total), _upstream(g_reg, g_total)
        if ctx.d_cls is None:
            raise RetinaHeadError("the fused loss keeps one set of gradients: backward() can run "
                                  "once per forward()")
        d_cls, d_reg = ctx.d_cls, ctx.d_reg
        ctx.d_cls = ctx.d_reg = None
        need = ctx.needs_input_grad
        if not as_stored:
            F.apply_upstream(d_cls if need[0] else None, g_cls, d_reg if need[1] else None, g_reg,
                             ctx.expected)
        out_cls = d_cls.to(ctx.in_dtypes[0]) if need[0] else None
        out_reg = d_reg.to(ctx.in_dtypes[1]) if need[1] else None
        return out_cls, out_reg, None, None, None


class _DenseHeadLevelsLossFn(torch.autograd.Function):
    """As _DenseHeadLossFn, for per-level head outputs [B, A*C, H, W] / [B, A*4, H, W]."""

    @staticmethod
    def forward(ctx, cfg, anchors, annotations, num_levels, *levels):
        cls_levels, reg_levels = levels[:num_levels], levels[num_levels:]
        out = _run_fused(F.focal_loss_levels_fwd_bwd,
                         (cls_levels, reg_levels, anchors, annotations), cfg)
        ctx.flat = (out["d_cls_all"], out["d_reg_all"])
        ctx.grads = (out["d_cls_levels"], out["d_reg_levels"])
        ctx.expected = cfg["expected_grad"]
        ctx.in_dtypes = [t.dtype for t in levels]
        cfg["state"].set(anchors.device, out.small_only())
        ctx.set_materialize_grads(False)
        return _loss_outputs(out)

    @staticmethod
    def backward(ctx, g_cls, g_reg, g_total):
        as_stored = _stored_as_is(ctx, g_cls, g_reg, g_total)
        g_cls, g_reg = _upstream(g_cls, g_total), _upstream(g_reg, g_total)
        if ctx.flat is None:
            raise RetinaHeadError("the fused loss keeps one set of gradients: backward() can run "
                                  "once per forward()")
        (d_cls_all, d_reg_all), (d_cls, d_reg) = ctx.flat, ctx.grads
        ctx.flat = ctx.grads = None
        if not as_stored:
            F.apply_upstream(d_cls_all, g_cls, d_reg_all, g_reg, ctx.expected)
        grads = [g.to(dt) for g, dt in zip(list(d_cls) + list(d_reg), ctx.in_dtypes)]
        need = ctx.needs_input_grad[4:]
        return (None, None, None, None) + tuple(g if n else None for g, n in zip(grads, need))


class FocalLoss(nn.Module):
    """Replaces FocalLoss (models/retinanet/losses.py:26-141).

    ``forward(classifications[B,N,C], regressions[B,N,4], anchors[1,N,4],
    annotations[B,M,5])`` returns ``{'classification_loss': Tensor[1],
    'regression_loss': Tensor[1]}``, differentiable with respect to the first two
    arguments.  The reference's constants are keyword arguments with the
    reference values as defaults (alpha .25, IoU thresholds .4/.5; gamma is 2).

    expected_grad: the upstream gradient both losses are expected to receive in
        ``backward`` (1 for ``sum(losses).backward()``, ``1/iterations`` for the
        accumulation loop of train.py:185-190).  The stored gradients are
        pre-multiplied by it, so a matching backward touches no memory; any other
        value is applied by one in-place scaling pass.
    batch_divisor: number of images the batch mean runs over (0: this batch).
        A data-parallel shard passes the global batch size.
    check_labels: read the device status word after the call (one host
        synchronisation) and raise IndexError for a label outside [0, C), as
        losses.py:79 does, and ValueError for a valid row with a NaN or infinite
        coordinate (the reference lets the NaN IoU through: such an anchor is then
        neither negative nor positive; here the non-finite box is skipped by the
        comparisons, so the image's assignment can differ -- hence the flag).
    from_logits: ``classifications`` holds the head's logits instead of probabilities:
        the ``nn.Sigmoid`` of models/retinanet/model.py:177,193 is folded into the loss
        kernel (same loss; the gradient is the one autograd gives through
        ``torch.sigmoid``), which saves the head a full read and write of the [B,N,C]
        tensor in each direction.
    normalizer: 'per_image' is the reference (each image's sums divided by its own positive
        count, losses.py:94 and :134, then the batch mean, :139-141).  'global' divides the
        batch's sums by the positive count of the whole batch instead (no batch mean;
        ``batch_divisor`` unused); under ``retinahead.dist.DataParallelFocalLoss`` that count
        runs over all ranks.  Both agree when every image has the same number of positives.
    """

    def __init__(self, alpha=0.25, neg_thresh=0.4, pos_thresh=0.5, expected_grad=1.0,
                 batch_divisor=0, check_labels=False, from_logits=False, normalizer="per_image"):
        super().__init__()
        if normalizer not in _lib.NORMALIZERS:
            raise ValueError("normalizer must be 'per_image' or 'global'")
        self.cfg = dict(alpha=float(alpha), neg_thresh=float(neg_thresh),
                        pos_thresh=float(pos_thresh), expected_grad=float(expected_grad),
                        batch_divisor=int(batch_divisor), from_logits=bool(from_logits),
                        normalizer=normalizer, reduce_total=None, state=_CallState(),
                        ahead=_AssignAhead())
        self.check_labels = bool(check_labels)

    def prefetch_assignment(self, anchors, annotations, num_classes):
        """Optional: names the annotations of the step AFTER the next ``forward`` call -- batch
        i+1, announced before ``forward`` of batch i.  That call's kernel launch then also does
        batch i+1's IoU assignment (losses.py:57-79, which needs no predictions): its [B,N,C]
        stream waits on HBM and leaves the vector ALUs free, so the assignment costs about a
        fifth of what it costs alone.  The later ``forward`` with the same ``annotations`` tensor
        (unmodified), anchors, batch size and class count skips its assignment stage; any other
        call computes its own, so results never depend on this method.  Not used by
        ``forward_levels`` or when a ``reduce_total`` hook stages the call.
        Stream order: the kernel that carries the assignment reads ``annotations`` on the stream
        of the next ``forward`` call, so the tensor must be complete on that stream by then -- an
        upload made on a loader's side stream has to be ordered before it (``wait_stream`` /
        an event), exactly as for a tensor passed to ``forward`` itself."""
        self.cfg["ahead"].ask(anchors, annotations, num_classes, self.cfg)

    @property
    def last(self):
        """Device tensors of the latest call: per_image [2,B], num_pos [B], status [B], exchange [4]
        (both losses, positive anchors, images: the vector data-parallel ranks sum).  Under
        nn.DataParallel each replica thread sees its own call; ``last_by_device`` has them all."""
        return self.cfg["state"].mine

    @property
    def last_by_device(self):
        """{device index: outputs of the latest call on that device}."""
        return self.cfg["state"].by_device

    def _check_status(self):
        bits = int(self.last["status"].max().item())
        if bits & _lib.RH_STATUS_REDUCTION_GAVE_UP:
            raise RetinaHeadError("the loss reduction inside the stream launch gave up waiting for "
                                  "a record (internal error; the losses of this call are NaN)")
        if bits & _lib.RH_STATUS_LABEL_RANGE:
            raise IndexError("annotation label outside [0, num_classes)")
        if bits & _lib.RH_STATUS_NONFINITE_BOX:
            raise ValueError("annotation box with a NaN or infinite coordinate")

    def forward(self, classifications, regressions, anchors, annotations):
        cls_loss, reg_loss, total = _DenseHeadLossFn.apply(classifications, regressions, anchors,
                                                           annotations, self.cfg)
        if self.check_labels:
            self._check_status()
        return _as_loss_dict(cls_loss, reg_loss, total)

    def forward_levels(self, cls_levels, reg_levels, anchors, annotations):
        """The same loss on the heads' per-level outputs, without the permute / view / cat of
        models/retinanet/model.py:152-154,196-202,311-313: ``cls_levels[l]`` is the classification
        convolution's output [B, A*C, H_l, W_l] (probabilities, or logits with ``from_logits``),
        ``reg_levels[l]`` the regression convolution's [B, A*4, H_l, W_l], levels in the order of
        the anchors.  Gradients arrive in the same layouts."""
        cls_loss, reg_loss, total = _DenseHeadLevelsLossFn.apply(
            self.cfg, anchors, annotations, len(cls_levels), *cls_levels, *reg_levels)
        if self.check_labels:
            self._check_status()
        return _as_loss_dict(cls_loss, reg_loss, total)


class _GenericFocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, alpha, gamma, logits, reduce):
        loss, grad = F.generic_focal_fwd_bwd(inputs, targets, alpha, gamma, logits, reduce)
        ctx.save_for_backward(grad)
        ctx.in_dtype = inputs.dtype
        return loss.clone() if reduce else loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).to(ctx.in_dtype), None, None, None, None, None


class GenericFocalLoss(nn.Module):
    """Replaces FocalLoss (models/focal_loss.py:6-25): alpha*(1-pt)^gamma*BCE on logits or
    probabilities, mean over all elements when ``reduce``."""

    def __init__(self, alpha=0.25, gamma=2, logits=True, reduce=True):
        super().__init__()
        self.alpha, self.gamma, self.logits, self.reduce = alpha, gamma, logits, reduce

    def forward(self, inputs, targets):
        return _GenericFocalFn.apply(inputs, targets, float(self.alpha), float(self.gamma),
                                     bool(self.logits), bool(self.reduce))


class Anchors(nn.Module):
    """Replaces Anchors (models/retinanet/anchors.py:6-45).

    ``forward(image[B,3,H,W])`` returns float32 ``[1,N,4]`` on ``image.device``;
    only the image's spatial shape is read.  Results are cached per (H, W, device).
    Unlike the reference constructor (which keeps a passed argument only when it is
    None, anchors.py:11-20) explicit levels / strides / sizes / ratios / scales are honoured.
    """

    def __init__(self, pyramid_levels=None, strides=None, sizes=None, ratios=None, scales=None):
        super().__init__()
        self.pyramid_levels = [3, 4, 5, 6, 7] if pyramid_levels is None else list(pyramid_levels)
        self.strides = [2 ** x for x in self.pyramid_levels] if strides is None else list(strides)
        self.sizes = [2 ** (x + 2) for x in self.pyramid_levels] if sizes is None else list(sizes)
        self.ratios = np.array([0.5, 1, 2]) if ratios is None else np.asarray(ratios, np.float64)
        self.scales = (np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)]) if scales is None
                       else np.asarray(scales, np.float64))
        self._cfg = None
        self._cache = {}

    def _config(self):
        if self._cfg is None:
            self._cfg = F.make_anchor_cfg(self.pyramid_levels, self.strides, self.sizes,
                                          self.ratios, self.scales)
        return self._cfg

    def num_anchors(self, height, width):
        return F.anchor_count(self._config(), height, width)

    def forward(self, image):
        h, w = int(image.shape[2]), int(image.shape[3])
        key = (h, w, image.device)
        hit = self._cache.get(key)
        if hit is None:
            hit = F.anchors(self._config(), h, w, image.device)
            if len(self._cache) > 64:
                self._cache.clear()
            self._cache[key] = hit
        return hit


class BBoxTransform(nn.Module):
    """Replaces BBoxTransform (models/retinanet/utils.py:83-125)."""

    def __init__(self, mean=None, std=None):
        super().__init__()
        self.mean = [0.0, 0.0, 0.0, 0.0] if mean is None else [float(x) for x in mean]
        self.std = [0.1, 0.1, 0.2, 0.2] if std is None else [float(x) for x in std]

    def forward(self, boxes, deltas):
        return F.bbox_transform(boxes, deltas, self.mean, self.std)


class ClipBoxes(nn.Module):
    """Replaces ClipBoxes (models/retinanet/utils.py:128-142); clamps ``boxes`` in place."""

    def __init__(self, width=None, height=None):
        super().__init__()

    def forward(self, boxes, img):
        height, width = int(img.shape[2]), int(img.shape[3])
        if boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.data_ptr() % 16 == 0:
            return F.clip_boxes_(boxes, height, width)
        tmp = F.clip_boxes_(boxes.float().contiguous().clone(), height, width)
        boxes.copy_(tmp)
        return boxes


def nms(boxes, scores=None, iou_threshold=None):
    """Replaces ``torchvision.ops.nms(boxes[K,4], scores[K] or [K,1], iou_threshold)`` as called at
    models/retinanet/model.py:336.  The legacy form ``nms(dets[K,5], thresh)`` of the removed
    extension (model.py:11-14, 335) is accepted too.  Returns int64 indices of the kept boxes in
    descending score order."""
    if iou_threshold is None:          # legacy (dets, thresh)
        dets, thresh = boxes, scores
        if dets.dim() != 2 or dets.shape[1] != 5:
            raise RetinaHeadError("legacy nms form expects dets [K,5] and a threshold")
        return F.nms(dets[:, :4], dets[:, 4], float(thresh))
    return F.nms(boxes, scores, float(iou_threshold))


def batched_nms(boxes, scores, labels, iou_threshold):
    """Per-class NMS: boxes only suppress boxes with the same label."""
    return F.nms(boxes, scores, float(iou_threshold), labels=labels)


def postprocess(classification, regression, anchors, img_batch, score_thresh=0.05, iou_thresh=0.5,
                mode="agnostic", pre_nms_topk=None, max_det=None, max_candidates=4096,
                output="dicts", from_logits=False):
    """Inference post-processing for the whole batch; replaces models/retinanet/model.py:320-340.

    classification [B,N,C] probabilities, regression [B,N,4], anchors [1,N,4] -- or, to skip the
    permute / view / cat of model.py:152-154,196-202,311-313, lists with the heads' per-level
    outputs ``[B, A*C, H_l, W_l]`` and ``[B, A*4, H_l, W_l]`` (same results); ``img_batch`` is the
    image tensor [B,3,H,W] (only its shape is read) or an ``(H, W)`` pair.
    mode 'agnostic' is the reference behaviour (one NMS over all classes on the max-class score);
    'per_class' lets only equal labels suppress each other.
    Returns ``[{'boxes','labels','scores'}]`` per image (``output='dicts'``, the form
    visualize.py:95-101 reads) or image 0's ``[scores, labels, boxes]`` (``output='tuple'``, the
    form of model.py:340).  ``max_candidates`` sizes the first attempt's buffers; if an image has
    more candidates over the threshold the call is repeated with room for all of them, so results
    never depend on it.  ``from_logits``: ``classification`` holds the head's logits (the
    ``nn.Sigmoid`` of model.py:177,193 left out); scores are their float32 sigmoid and the result
    is the one the probability form gives for ``retinahead.sigmoid(classification)``.
    """
    if mode not in ("agnostic", "per_class"):
        raise ValueError("mode must be 'agnostic' or 'per_class'")
    if hasattr(img_batch, "shape"):
        height, width = int(img_batch.shape[2]), int(img_batch.shape[3])
    else:
        height, width = int(img_batch[0]), int(img_batch[1])
    levels = isinstance(classification, (list, tuple))
    n = anchors.shape[-2] if levels else classification.shape[1]
    raw = F.postprocess_levels_raw if levels else F.postprocess_raw
    cap = min(n, int(max_candidates))
    while True:
        out = raw(classification, regression, anchors, height, width, score_thresh,
                  iou_thresh, mode == "per_class", pre_nms_topk or 0, max_det or 0, cap,
                  from_logits=from_logits)
        counts = out["counts"].cpu()          # the one host read: output sizes are data dependent
        if int(counts[2].max()) == 0 or cap >= n:
            break
        cap = min(n, int(counts[1].max()))
    kept = counts[0].tolist()
    if min(kept) < 0:
        raise RetinaHeadError("NMS sweep gave up waiting inside a workgroup (internal error)")
    if output == "tuple":
        k = kept[0]
        return [out["scores"][0, :k], out["labels"][0, :k], out["boxes"][0, :k]]
    return [{"boxes": out["boxes"][b, :k], "labels": out["labels"][b, :k],
             "scores": out["scores"][b, :k]} for b, k in enumerate(kept)]
