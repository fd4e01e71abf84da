"""Tensor-level wrappers over the C ABI: argument checks, workspace, stream plumbing.

PyTorch is used for device memory, the current stream and autograd bookkeeping
only; every computation below is a libretinahead.so kernel.
"""
from __future__ import annotations

import ctypes as C
import threading
from collections import OrderedDict
from typing import Optional

import torch

from . import _lib
from ._lib import RetinaHeadError

# Caches shared by every thread that calls into the package (nn.DataParallel drives its replicas
# from several Python threads of one process, train.py:137): one lock guards them all.
_cache_lock = threading.Lock()
_workspaces = OrderedDict()     # (device, stream, tag) -> uint8 buffer, least recently used first
_WORKSPACE_SLOTS = 64           # streams come and go: beyond this the oldest entries are dropped
_capture_workspaces = {}        # the same for calls made under stream capture
_captured = []                  # every buffer a captured graph may hold the address of: kept for good


def _dev_check(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RetinaHeadError("retinahead operators need tensors on a HIP device "
                                  "(there is no CPU fallback)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RetinaHeadError(f"tensors on different devices: {dev} and {t.device}")
    return dev


def _f32c(t):
    """float32, contiguous, 16-byte aligned storage offset."""
    if t.dtype != torch.float32:
        t = t.float()
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _on_device:
    """``with _on_device(dev) as stream:`` makes dev current only when it is not already (the
    common single-GPU-per-process case costs one integer compare) and yields the raw stream."""
    __slots__ = ("dev", "guard")

    def __init__(self, dev):
        self.dev = dev
        self.guard = None

    def __enter__(self):
        if torch.cuda.current_device() != self.dev.index:
            self.guard = torch.cuda.device(self.dev)
            self.guard.__enter__()
        return torch.cuda.current_stream(self.dev).cuda_stream

    def __exit__(self, *exc):
        if self.guard is not None:
            self.guard.__exit__(*exc)
        return False


def workspace(dev, nbytes, tag="default", stream=None):
    """A cached byte buffer per (device, stream, tag); grows, never shrinks.

    The C entry points bake the buffer's address into their launches, so a buffer that a stream
    capture has seen must outlive every replay of that graph.  Calls made during capture
    therefore use a cache of their own whose buffers are never freed, never handed to an eager
    call and never replaced in place: a captured call that needs more room gets a new buffer
    (from the graph's memory pool) and the old one stays alive for the graphs that hold its
    address.  The captured shapes pin their workspace; graphs captured on one stream share it and
    must not be replayed concurrently.  Eager entries of streams that are no longer used fall
    out of the cache in least-recently-used order."""
    if stream is None:
        stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev.index, stream, tag)
    capturing = torch.cuda.is_current_stream_capturing()
    with _cache_lock:
        cache = _capture_workspaces if capturing else _workspaces
        buf = cache.get(key)
        if buf is None or buf.numel() < nbytes:
            # grow in steps of a quarter so that slowly growing shapes do not reallocate every call
            want = max(int(nbytes), 256) if buf is None else max(int(nbytes), buf.numel() * 5 // 4)
            # not filled: no stage reads a byte that an earlier stage of the same call has not
            # written (the loss's assignment stage clears the tags of the records that the
            # reduction inside the stream launch polls), and an allocation that fills would
            # become a memset node of its own when made under stream capture
            buf = torch.empty(want, dtype=torch.uint8, device=dev)
            cache[key] = buf
            if capturing:
                _captured.append(buf)
        if not capturing:
            _workspaces.move_to_end(key)
            while len(_workspaces) > _WORKSPACE_SLOTS:
                _workspaces.popitem(last=False)
    return buf


def keep_for_graphs(*tensors):
    """Buffers whose addresses a stream capture has baked into a graph: held until the process
    ends, since any later replay reads and writes them."""
    with _cache_lock:
        for t in tensors:
            if t is not None and not any(t is k for k in _captured):
                _captured.append(t)


# --------------------------------------------------------------------------- anchors
def make_anchor_cfg(pyramid_levels=None, strides=None, sizes=None, ratios=None, scales=None):
    cfg = _lib.AnchorCfg()
    _lib.load().rh_anchor_cfg_default(C.byref(cfg))
    if pyramid_levels is not None:
        levels = [int(l) for l in pyramid_levels]
        if not 1 <= len(levels) <= _lib.MAX_LEVELS:
            raise RetinaHeadError(f"1..{_lib.MAX_LEVELS} pyramid levels supported")
        cfg.num_levels = len(levels)
        for i, l in enumerate(levels):
            cfg.levels[i] = l
            cfg.strides[i] = float(2 ** l)
            cfg.sizes[i] = float(2 ** (l + 2))
    if strides is not None:
        for i, s in enumerate(strides):
            cfg.strides[i] = float(s)
    if sizes is not None:
        for i, s in enumerate(sizes):
            cfg.sizes[i] = float(s)
    if ratios is not None:
        r = [float(x) for x in ratios]
        if not 1 <= len(r) <= _lib.MAX_RATIOS:
            raise RetinaHeadError(f"1..{_lib.MAX_RATIOS} ratios supported")
        cfg.num_ratios = len(r)
        for i, x in enumerate(r):
            cfg.ratios[i] = x
    if scales is not None:
        s = [float(x) for x in scales]
        if not 1 <= len(s) <= _lib.MAX_SCALES:
            raise RetinaHeadError(f"1..{_lib.MAX_SCALES} scales supported")
        cfg.num_scales = len(s)
        for i, x in enumerate(s):
            cfg.scales[i] = x
    return cfg


def anchor_count(cfg, height, width):
    n = _lib.load().rh_anchor_count(C.byref(cfg), int(height), int(width))
    if n < 0:
        _lib.check(int(n), "rh_anchor_count")
    return int(n)


def anchors(cfg, height, width, device):
    device = torch.device(device)
    if device.type != "cuda":
        raise RetinaHeadError("anchors are generated on a HIP device (no CPU fallback)")
    n = anchor_count(cfg, height, width)
    out = torch.empty((1, n, 4), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.load().rh_anchors_f32(C.byref(cfg), int(height), int(width), _p(out), n,
                                              _stream(device)), "rh_anchors_f32")
    return out


# --------------------------------------------------------------------------- IoU / assignment
def calc_iou(a, b):
    dev = _dev_check(a, b)
    a, b = _f32c(a), _f32c(b)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != 4 or b.shape[1] != 4:
        raise RetinaHeadError("calc_iou expects a[N,4], b[M,4]")
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rh_calc_iou_f32(_p(a), a.shape[0], _p(b), b.shape[0], _p(out),
                                               _stream(dev)), "rh_calc_iou_f32")
    return out


def _anchors2d(anchors_t):
    if anchors_t.dim() == 3:
        if anchors_t.shape[0] != 1:
            raise RetinaHeadError("anchors must be [1,N,4] or [N,4]")
        anchors_t = anchors_t[0]
    if anchors_t.dim() != 2 or anchors_t.shape[1] != 4:
        raise RetinaHeadError("anchors must be [1,N,4] or [N,4]")
    return _f32c(anchors_t)


def assign(anchors_t, annotations, num_classes, neg_thresh=0.4, pos_thresh=0.5):
    """Per-anchor assignment for a batch; returns a dict of device tensors."""
    dev = _dev_check(anchors_t, annotations)
    anc = _anchors2d(anchors_t)
    ann = _f32c(annotations)
    if ann.dim() != 3 or ann.shape[2] != 5:
        raise RetinaHeadError("annotations must be [B,M,5]")
    bsz, m, n = ann.shape[0], ann.shape[1], anc.shape[0]
    lib = _lib.load()
    state = torch.empty((bsz, n), dtype=torch.uint8, device=dev)
    argmax = torch.empty((bsz, n), dtype=torch.int32, device=dev)
    label = torch.empty((bsz, n), dtype=torch.int32, device=dev)
    iou_max = torch.empty((bsz, n), dtype=torch.float32, device=dev)
    num_pos = torch.empty((bsz,), dtype=torch.int32, device=dev)
    status = torch.empty((bsz,), dtype=torch.int32, device=dev)
    need = lib.rh_assign_workspace_bytes(bsz, n, m)
    ws = workspace(dev, need, "assign")
    with torch.cuda.device(dev):
        _lib.check(lib.rh_assign(_p(anc), _p(ann), bsz, n, m, int(num_classes), float(neg_thresh),
                                 float(pos_thresh), _p(state), _p(argmax), _p(label), _p(iou_max),
                                 _p(num_pos), _p(status), _p(ws), ws.numel(), _stream(dev)),
                   "rh_assign")
    return dict(state=state, argmax=argmax, label=label, iou_max=iou_max, num_pos=num_pos,
                status=status)


# --------------------------------------------------------------------------- fused loss
_ws_bytes_cache = {}

# Layout of the one small buffer that carries the scalar-ish outputs of a loss call (float32
# words): [classification loss, regression loss, positives, images, total | per_image 2B |
# num_pos B (int32) | status B (int32)].
_SMALL_HEAD = 5
_SMALL_VIEWS = {
    "losses": lambda t, b: t[0:2],
    "exchange": lambda t, b: t[0:4],
    "total": lambda t, b: t[4:5],
    "per_image": lambda t, b: t[5:5 + 2 * b].view(2, b),
    "num_pos": lambda t, b: t[5 + 2 * b:5 + 3 * b].view(torch.int32),
    "status": lambda t, b: t[5 + 3 * b:5 + 4 * b].view(torch.int32),
}


class LossOutputs(dict):
    """The outputs of a loss call.  Views of the small buffer (``losses``, ``exchange``, ``total``,
    ``per_image``, ``num_pos``, ``status``) are made when first asked for: slicing a tensor costs
    microseconds of host time each, and a training step reads two of the six."""

    def __missing__(self, key):
        make = _SMALL_VIEWS.get(key)
        if make is None:
            raise KeyError(key)
        small = dict.__getitem__(self, "small")
        view = make(small, (small.numel() - _SMALL_HEAD) // 4)
        self[key] = view
        return view

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in _SMALL_VIEWS

    def small_only(self):
        """The same small outputs without the gradient buffers."""
        return LossOutputs(small=dict.__getitem__(self, "small"))


def _small_ptrs(out, bsz):
    small = dict.__getitem__(out, "small") if isinstance(out, LossOutputs) else None
    if small is None:       # a plain dict from a caller: take its views as they are
        return (out["losses"].data_ptr(), out["per_image"].data_ptr(), out["num_pos"].data_ptr(),
                out["status"].data_ptr())
    base = small.data_ptr()
    return base, base + 4 * _SMALL_HEAD, base + 4 * (_SMALL_HEAD + 2 * bsz), base + 4 * (_SMALL_HEAD + 3 * bsz)


_params_cache = {}


def _focal_params(alpha, neg_thresh, pos_thresh, grad_cls, grad_reg, batch_divisor, from_logits,
                  normalizer, num_pos_total, dev):
    if num_pos_total is None:       # the usual case: one structure per setting, built once
        key = (alpha, neg_thresh, pos_thresh, grad_cls, grad_reg, batch_divisor, from_logits, normalizer)
        hit = _params_cache.get(key)
        if hit is None:
            hit = _build_focal_params(*key, None, dev)
            with _cache_lock:
                if len(_params_cache) > 256:
                    _params_cache.clear()
                _params_cache[key] = hit
        return hit
    return _build_focal_params(alpha, neg_thresh, pos_thresh, grad_cls, grad_reg, batch_divisor,
                               from_logits, normalizer, num_pos_total, dev)


def _build_focal_params(alpha, neg_thresh, pos_thresh, grad_cls, grad_reg, batch_divisor, from_logits,
                        normalizer, num_pos_total, dev):
    try:
        norm = _lib.NORMALIZERS[normalizer]
    except KeyError:
        raise ValueError("normalizer must be 'per_image' or 'global'") from None
    total = None
    if num_pos_total is not None:
        if (num_pos_total.device != dev or num_pos_total.dtype != torch.float32
                or num_pos_total.numel() != 1):
            raise RetinaHeadError("num_pos_total must be one float32 on the inputs' device")
        total = num_pos_total.data_ptr()
    return _lib.FocalParams(float(alpha), float(neg_thresh), float(pos_thresh), float(grad_cls),
                            float(grad_reg), int(batch_divisor), int(bool(from_logits)), norm, total)


def _focal_ws_bytes(lib, bsz, n, c, m):
    shape_key = (bsz, n, c, m)
    need = _ws_bytes_cache.get(shape_key)
    if need is None:
        need = _ws_bytes_cache[shape_key] = lib.rh_focal_loss_workspace_bytes(bsz, n, c, m)
    return need


def new_small_outputs(bsz, dev):
    """The small outputs of one loss call over ``bsz`` images, unwritten."""
    return LossOutputs(small=torch.empty((_SMALL_HEAD + 4 * bsz,), dtype=torch.float32, device=dev))


def focal_workspace(bsz, n, num_classes, m, dev):
    """A workspace of its own for one loss step over [bsz, n, num_classes] with m annotation rows."""
    need = _focal_ws_bytes(_lib.load(), bsz, n, int(num_classes), m)
    if need == 0:
        raise RetinaHeadError(f"unsupported shape B={bsz} N={n} C={num_classes} M={m}")
    return torch.empty((need,), dtype=torch.uint8, device=dev)


def focal_assign(anchors_t, annotations, num_classes, neg_thresh=0.4, pos_thresh=0.5,
                 normalizer="per_image", ws=None, out=None):
    """The assignment stage of a later ``focal_loss_fwd_bwd`` call, run now on the current stream:
    it reads the anchors and the annotations only (losses.py:48-79), so a trainer can issue it for
    the next batch on a second stream while the current step's stream stage runs.  Returns
    ``(out, ws)``: the small outputs (status, and num_pos under the batch-wide normaliser, are
    written) and the step's workspace, both to be handed to
    ``focal_loss_fwd_bwd(..., stages=STAGE_STREAM | STAGE_FINALIZE, out=out, ws=ws)`` with the
    same annotations, thresholds and normaliser.  ``ws``: a buffer to reuse (a uint8 tensor of at
    least the workspace size), else a new one; ``out``: small outputs allocated by the caller
    (``new_small_outputs``), e.g. on the stream that will run the rest of the step."""
    dev = _dev_check(anchors_t, annotations)
    anc, ann = _anchors2d(anchors_t), _f32c(annotations)
    if ann.dim() != 3 or ann.shape[2] != 5:
        raise RetinaHeadError("annotations must be [B,M,5]")
    bsz, m, n, c = ann.shape[0], ann.shape[1], anc.shape[0], int(num_classes)
    lib = _lib.load()
    need = _focal_ws_bytes(lib, bsz, n, c, m)
    if need == 0:
        raise RetinaHeadError(f"unsupported shape B={bsz} N={n} C={c} M={m}")
    if ws is None or ws.device != dev or ws.dtype != torch.uint8 or ws.numel() < need:
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    if out is None:
        out = new_small_outputs(bsz, dev)
    params = _focal_params(0.25, neg_thresh, pos_thresh, 1.0, 1.0, 0, False, normalizer, None, dev)
    p_losses, p_per_image, p_num_pos, p_status = _small_ptrs(out, bsz)
    with _on_device(dev) as stream:
        _lib.check(lib.rh_focal_loss_fwd_bwd(
            None, None, anc.data_ptr(), ann.data_ptr(), bsz, n, c, m, C.byref(params), None, None,
            p_losses, p_per_image, p_num_pos, p_status, ws.data_ptr(), ws.numel(),
            int(_lib.STAGE_ASSIGN), stream), "rh_focal_loss_fwd_bwd")
    return out, ws


def focal_loss_fwd_bwd(classifications, regressions, anchors_t, annotations, alpha=0.25,
                       neg_thresh=0.4, pos_thresh=0.5, grad_cls=1.0, grad_reg=1.0,
                       batch_divisor=0, stages=_lib.STAGE_ALL, out=None, from_logits=False,
                       normalizer="per_image", num_pos_total=None, ws=None, ahead=None):
    """Runs the fused loss kernels.  Returns a dict of device tensors:
    losses [2], exchange [4] (the losses, the call's positive anchors and its images: what
    data-parallel ranks sum), total [1] (the losses' sum), per_image [2,B], num_pos [B], status [B],
    d_classifications, d_regressions.
    ``out`` may carry buffers from an earlier call (used when stages are run one at a time).
    ``normalizer='global'``: one positive count for the whole batch instead of the reference's
    per-image counts and batch mean; ``num_pos_total`` (device float32 [1]) supplies a count
    formed elsewhere (over all ranks of a data-parallel job), else the call counts its own.
    ``ws``: the step's workspace (uint8 tensor) when the caller keeps it, as ``focal_assign``
    does for a step whose assignment ran ahead; else the stream's cached one.
    ``ahead``: ``(annotations, ws, out)`` of the NEXT step -- its annotations [B,M',5], its
    workspace (``focal_workspace``) and its small outputs (``new_small_outputs``): the next
    step's assignment stage is carried by this call's stream launch, and the next call runs
    ``stages=STAGE_STREAM | STAGE_FINALIZE`` with ``ws=`` and ``out=`` set to those.
    """
    dev = _dev_check(classifications, regressions, anchors_t, annotations)
    cls, reg = _f32c(classifications), _f32c(regressions)
    anc, ann = _anchors2d(anchors_t), _f32c(annotations)
    if cls.dim() != 3 or reg.dim() != 3 or reg.shape[2] != 4 or cls.shape[:2] != reg.shape[:2]:
        raise RetinaHeadError("expected classifications [B,N,C] and regressions [B,N,4]")
    if ann.dim() != 3 or ann.shape[2] != 5 or ann.shape[0] != cls.shape[0]:
        raise RetinaHeadError("annotations must be [B,M,5]")
    if anc.shape[0] != cls.shape[1]:
        raise RetinaHeadError(f"{anc.shape[0]} anchors but {cls.shape[1]} prediction rows")
    bsz, n, c = cls.shape
    m = ann.shape[1]
    lib = _lib.load()
    if out is None:
        # one small allocation for all scalar-ish outputs:
        # [losses 2, positives 1, images 1, total 1 | per_image 2B | num_pos B | status B]
        out = LossOutputs(
            small=torch.empty((_SMALL_HEAD + 4 * bsz,), dtype=torch.float32, device=dev),
            d_classifications=torch.empty_like(cls),
            d_regressions=torch.empty_like(reg),
        )
    elif not dict.__contains__(out, "d_classifications"):    # from focal_assign: small outputs only
        out["d_classifications"] = torch.empty_like(cls)
        out["d_regressions"] = torch.empty_like(reg)
    params = _focal_params(alpha, neg_thresh, pos_thresh, grad_cls, grad_reg, batch_divisor,
                           from_logits, normalizer, num_pos_total, dev)
    need = _focal_ws_bytes(lib, bsz, n, c, m)
    if ws is not None and (ws.device != dev or ws.dtype != torch.uint8 or ws.numel() < need):
        raise RetinaHeadError(f"ws must be a uint8 tensor of at least {need} bytes on {dev}")
    p_losses, p_per_image, p_num_pos, p_status = _small_ptrs(out, bsz)
    with _on_device(dev) as stream:
        if ws is None:
            ws = workspace(dev, need, "focal", stream)
        if ahead is None:
            _lib.check(lib.rh_focal_loss_fwd_bwd(
                cls.data_ptr(), reg.data_ptr(), anc.data_ptr(), ann.data_ptr(), bsz, n, c, m,
                C.byref(params), out["d_classifications"].data_ptr(), out["d_regressions"].data_ptr(),
                p_losses, p_per_image, p_num_pos, p_status, ws.data_ptr(), ws.numel(), int(stages), stream),
                "rh_focal_loss_fwd_bwd")
        else:
            n_ann, n_ws, n_out = ahead
            n_ann = _f32c(n_ann)
            if n_ann.dim() != 3 or n_ann.shape[2] != 5 or n_ann.shape[0] != bsz or n_ann.device != dev:
                raise RetinaHeadError("the next step's annotations must be [B,M,5] for the same B and device")
            _, _, n_num_pos, n_status = _small_ptrs(n_out, bsz)
            nxt = _lib.AssignAhead(n_ann.data_ptr(), n_ann.shape[1], n_ws.data_ptr(), n_ws.numel(),
                                   n_status, n_num_pos)
            _lib.check(lib.rh_focal_loss_fwd_bwd_ahead(
                cls.data_ptr(), reg.data_ptr(), anc.data_ptr(), ann.data_ptr(), bsz, n, c, m,
                C.byref(params), out["d_classifications"].data_ptr(), out["d_regressions"].data_ptr(),
                p_losses, p_per_image, p_num_pos, p_status, ws.data_ptr(), ws.numel(), int(stages),
                C.byref(nxt), stream), "rh_focal_loss_fwd_bwd_ahead")
    return out


def focal_loss_levels_fwd_bwd(cls_levels, reg_levels, anchors_t, annotations, alpha=0.25,
                              neg_thresh=0.4, pos_thresh=0.5, grad_cls=1.0, grad_reg=1.0,
                              batch_divisor=0, stages=_lib.STAGE_ALL, out=None, from_logits=False,
                              normalizer="per_image", num_pos_total=None):
    """The fused loss on the heads' own outputs: ``cls_levels[l]`` is [B, A*C, H_l, W_l] and
    ``reg_levels[l]`` is [B, A*4, H_l, W_l], the tensors the reference permutes, reshapes and
    concatenates into [B,N,C] / [B,N,4] (models/retinanet/model.py:152-154,196-202,311-313).
    Returns the dict of :func:`focal_loss_fwd_bwd` with ``d_cls_levels`` / ``d_reg_levels`` (lists,
    layouts of the inputs; views of the flat buffers ``d_cls_all`` / ``d_reg_all``) instead of
    the flat gradients."""
    if len(cls_levels) != len(reg_levels) or not cls_levels:
        raise RetinaHeadError("need as many classification as regression levels (at least one)")
    dev = _dev_check(*cls_levels, *reg_levels, anchors_t, annotations)
    cls = [_f32c(t) for t in cls_levels]
    reg = [_f32c(t) for t in reg_levels]
    anc, ann = _anchors2d(anchors_t), _f32c(annotations)
    bsz = cls[0].shape[0]
    if any(t.dim() != 4 or t.shape[0] != bsz for t in cls + reg):
        raise RetinaHeadError("level tensors must be [B, channels, H, W]")
    hs = [int(t.shape[2]) for t in cls]
    wsz = [int(t.shape[3]) for t in cls]
    cells = sum(h * w for h, w in zip(hs, wsz))
    if anc.shape[0] % cells:
        raise RetinaHeadError(f"{anc.shape[0]} anchors do not divide over {cells} cells")
    a = anc.shape[0] // cells
    if cls[0].shape[1] % a:
        raise RetinaHeadError(f"{cls[0].shape[1]} classification channels with {a} anchors per cell")
    c = cls[0].shape[1] // a
    for tc, tr, h, w in zip(cls, reg, hs, wsz):
        if tuple(tc.shape) != (bsz, a * c, h, w) or tuple(tr.shape) != (bsz, a * 4, h, w):
            raise RetinaHeadError("inconsistent level shapes")
    if ann.dim() != 3 or ann.shape[2] != 5 or ann.shape[0] != bsz:
        raise RetinaHeadError("annotations must be [B,M,5]")
    m = ann.shape[1]
    nl = len(cls)
    lib = _lib.load()
    if out is None:
        d_cls_all = torch.empty((sum(t.numel() for t in cls),), dtype=torch.float32, device=dev)
        d_reg_all = torch.empty((sum(t.numel() for t in reg),), dtype=torch.float32, device=dev)

        def views(flat, like):
            res, at = [], 0
            for t in like:
                res.append(flat[at:at + t.numel()].view(t.shape))
                at += t.numel()
            return res
        out = LossOutputs(
            small=torch.empty((_SMALL_HEAD + 4 * bsz,), dtype=torch.float32, device=dev),
            d_cls_all=d_cls_all, d_reg_all=d_reg_all,
            d_cls_levels=views(d_cls_all, cls), d_reg_levels=views(d_reg_all, reg),
        )
    params = _focal_params(alpha, neg_thresh, pos_thresh, grad_cls, grad_reg, batch_divisor,
                           from_logits, normalizer, num_pos_total, dev)
    ptrs = lambda ts: (C.c_void_p * nl)(*[t.data_ptr() for t in ts])
    shape_key = ("levels", bsz, tuple(hs), tuple(wsz), a, m)
    hit = _ws_bytes_cache.get(shape_key)
    if hit is None:
        ints = lambda vs: (C.c_int32 * nl)(*vs)
        h_arr, w_arr = ints(hs), ints(wsz)
        need = lib.rh_focal_loss_levels_workspace_bytes(bsz, nl, h_arr, w_arr, a, m)
        if need == 0:
            raise RetinaHeadError("focal_loss_levels: " + _lib.last_error())
        hit = _ws_bytes_cache[shape_key] = (h_arr, w_arr, need)
    h_arr, w_arr, need = hit
    with _on_device(dev) as stream:
        ws = workspace(dev, need, "focal_levels", stream)
        _lib.check(lib.rh_focal_loss_levels_fwd_bwd(
            ptrs(cls), ptrs(reg), anc.data_ptr(), ann.data_ptr(), bsz, nl, h_arr, w_arr, a, c, m,
            C.byref(params), ptrs(out["d_cls_levels"]), ptrs(out["d_reg_levels"]),
            *_small_ptrs(out, bsz), ws.data_ptr(), ws.numel(), int(stages), stream),
            "rh_focal_loss_levels_fwd_bwd")
    return out


def apply_upstream(d_cls, g_cls, d_reg, g_reg, assumed=1.0):
    """In place and without a host read: d_cls *= g_cls/assumed, d_reg *= g_reg/assumed (one
    launch; an array whose scale equals ``assumed`` is not touched).  A ``None`` gradient zeroes
    its array, a ``None`` array is skipped."""
    dev = _dev_check(d_cls, g_cls, d_reg, g_reg)
    for t in (d_cls, d_reg):
        if t is not None and (not t.is_contiguous() or t.dtype != torch.float32):
            raise RetinaHeadError("apply_upstream needs contiguous float32 tensors")
    gc = None if g_cls is None else _f32c(g_cls.reshape(-1)[:1])
    gr = None if g_reg is None else _f32c(g_reg.reshape(-1)[:1])
    ptr = lambda t: 0 if t is None else t.data_ptr()
    with _on_device(dev) as stream:
        _lib.check(_lib.load().rh_apply_upstream_f32(
            ptr(d_cls), 0 if d_cls is None else d_cls.numel(), ptr(gc),
            ptr(d_reg), 0 if d_reg is None else d_reg.numel(), ptr(gr), float(assumed), stream),
            "rh_apply_upstream_f32")


def generic_focal_fwd_bwd(inputs, targets, alpha=0.25, gamma=2.0, logits=True, reduce=True):
    """Returns (loss, grad): loss is [1]-less scalar tensor when reduce else inputs-shaped."""
    dev = _dev_check(inputs, targets)
    x, t = _f32c(inputs), _f32c(targets)
    if x.shape != t.shape:
        raise RetinaHeadError("inputs and targets must have the same shape")
    lib = _lib.load()
    grad = torch.empty_like(x)
    loss_elem = None if reduce else torch.empty_like(x)
    loss_mean = torch.empty((), dtype=torch.float32, device=dev)
    ws = workspace(dev, lib.rh_generic_focal_workspace_bytes(x.numel()), "generic")
    with torch.cuda.device(dev):
        _lib.check(lib.rh_generic_focal_fwd_bwd(_p(x), _p(t), x.numel(), float(alpha), float(gamma),
                                                int(bool(logits)), int(bool(reduce)), _p(loss_elem),
                                                _p(loss_mean), _p(grad), _p(ws), ws.numel(),
                                                _stream(dev)), "rh_generic_focal_fwd_bwd")
    return (loss_mean if reduce else loss_elem), grad


# --------------------------------------------------------------------------- decode / clip
def bbox_transform(boxes, deltas, mean=None, std=None):
    dev = _dev_check(boxes, deltas)
    anc, d = _anchors2d(boxes), _f32c(deltas)
    if d.dim() != 3 or d.shape[2] != 4 or d.shape[1] != anc.shape[0]:
        raise RetinaHeadError("expected boxes [1,N,4] and deltas [B,N,4]")
    out = torch.empty_like(d)
    arr = lambda v: None if v is None else (C.c_float * 4)(*[float(x) for x in v])
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rh_bbox_transform_f32(_p(anc), _p(d), d.shape[0], d.shape[1], arr(mean),
                                                     arr(std), _p(out), _stream(dev)),
                   "rh_bbox_transform_f32")
    return out


def sigmoid(logits):
    """float32 sigmoid as the logits-in modes of the loss and of post-processing compute it (the
    head's output activation, models/retinanet/model.py:177,193)."""
    dev = _dev_check(logits)
    x = _f32c(logits)
    out = torch.empty_like(x)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rh_sigmoid_f32(_p(x), _p(out), x.numel(), _stream(dev)),
                   "rh_sigmoid_f32")
    return out


def collate_annotations(targets, min_rows=1, min_size=None, box_format="xyxy", return_counts=False):
    """Pads per-image targets to the ``[B, M, 5]`` annotation tensor the loss reads (rows
    ``x1, y1, x2, y2, label``, padding rows all -1), on the device; replaces the annotation half
    of ``collater`` (models/retinanet/model.py:27,43-55).  ``targets`` is a list of dicts with
    device tensors ``'boxes'`` [m,4] and ``'labels'`` [m] or [m,1]; M is the largest m (at least
    ``min_rows``: the reference pads an all-empty batch to one row).  No host read of device data.

    ``min_size`` / ``box_format='xywh'`` fold in what the reference's datasets do while loading
    (csv_dataset.py:129-136, coco_dataset.py:97-108): annotations narrower or lower than
    ``min_size`` (the reference: 1) are dropped, ``[x, y, w, h]`` rows become corners; survivors keep
    their order.  ``return_counts`` adds the int32 [B] tensor of rows kept per image."""
    if box_format not in ("xyxy", "xywh"):
        raise ValueError("box_format must be 'xyxy' or 'xywh'")
    if not targets:
        raise RetinaHeadError("collate_annotations needs at least one image")
    boxes = [t["boxes"].reshape(-1, 4) for t in targets]
    labels = [t["labels"].reshape(-1) for t in targets]
    dev = _dev_check(*boxes, *labels)
    counts = [int(b.shape[0]) for b in boxes]
    if any(int(l.shape[0]) != c for l, c in zip(labels, counts)):
        raise RetinaHeadError("boxes and labels disagree in length")
    m = max(max(counts), int(min_rows), 1)
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    all_boxes = _f32c(torch.cat(boxes)) if offs[-1] else None
    all_labels = _f32c(torch.cat(labels)) if offs[-1] else None
    offsets = torch.tensor(offs, dtype=torch.int32).to(dev, non_blocking=True)
    out = torch.empty((len(targets), m, 5), dtype=torch.float32, device=dev)
    plain = min_size is None and box_format == "xyxy"
    kept = None
    with torch.cuda.device(dev):
        if plain:
            _lib.check(_lib.load().rh_collate_annotations_f32(
                _p(all_boxes), _p(all_labels), _p(offsets), len(targets), m, _p(out), _stream(dev)),
                "rh_collate_annotations_f32")
        else:
            kept = torch.empty((len(targets),), dtype=torch.int32, device=dev)
            floor = float("-inf") if min_size is None else float(min_size)
            _lib.check(_lib.load().rh_sanitize_collate_annotations_f32(
                _p(all_boxes), _p(all_labels), _p(offsets), len(targets), m, floor,
                int(box_format == "xywh"), _p(out), _p(kept), _stream(dev)),
                "rh_sanitize_collate_annotations_f32")
    if return_counts:
        if kept is None:
            kept = (offsets[1:] - offsets[:-1]).to(torch.int32)
        return out, kept
    return out


def match_detections(det_boxes, det_labels, det_count, annotations, iou_threshold=0.5):
    """uint8 [B,D]: 1 where a detection is a true positive (csv_eval.py:188-208).  ``det_boxes``
    [B,D,4] / ``det_labels`` [B,D] hold each image's detections in descending score order,
    ``det_count`` [B] how many are valid, ``annotations`` [B,M,5] the padded ground truth."""
    dev = _dev_check(det_boxes, det_labels, det_count, annotations)
    boxes, ann = _f32c(det_boxes), _f32c(annotations)
    labels = det_labels.to(torch.int64).contiguous()
    count = det_count.to(torch.int32).contiguous()
    if boxes.dim() != 3 or boxes.shape[2] != 4 or tuple(labels.shape) != tuple(boxes.shape[:2]) \
            or ann.dim() != 3 or ann.shape[2] != 5 or ann.shape[0] != boxes.shape[0]:
        raise RetinaHeadError("expected det_boxes [B,D,4], det_labels [B,D], annotations [B,M,5]")
    bsz, d = boxes.shape[:2]
    tp = torch.zeros((bsz, d), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rh_match_detections(_p(boxes), _p(labels), _p(count), _p(ann), bsz, d,
                                                   ann.shape[1], float(iou_threshold), _p(tp),
                                                   _stream(dev)), "rh_match_detections")
    return tp


def clip_boxes_(boxes, height, width):
    """In place (utils.py:136-140)."""
    dev = _dev_check(boxes)
    if boxes.dtype != torch.float32 or not boxes.is_contiguous() or boxes.shape[-1] != 4 \
            or boxes.data_ptr() % 16:
        raise RetinaHeadError("clip_boxes_ needs a contiguous, aligned float32 [...,4] tensor")
    with torch.cuda.device(dev):
        _lib.check(_lib.load().rh_clip_boxes_f32(_p(boxes), boxes.numel() // 4, float(width),
                                                 float(height), _stream(dev)), "rh_clip_boxes_f32")
    return boxes


# --------------------------------------------------------------------------- NMS / post-processing
def nms(boxes, scores, iou_threshold, labels: Optional[torch.Tensor] = None):
    """Kept indices (int64, descending score).  One host read of the kept count."""
    dev = _dev_check(boxes, scores, labels)
    b = _f32c(boxes)
    s = _f32c(scores.reshape(-1))
    if b.dim() != 2 or b.shape[1] != 4 or s.shape[0] != b.shape[0]:
        raise RetinaHeadError("nms expects boxes [K,4] and scores [K] or [K,1]")
    k = b.shape[0]
    if k == 0:
        return torch.zeros((0,), dtype=torch.int64, device=dev)
    lab = None
    if labels is not None:
        lab = labels.reshape(-1).to(torch.int32).contiguous()
    lib = _lib.load()
    keep = torch.empty((k,), dtype=torch.int64, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    need = lib.rh_nms_workspace_bytes(k)
    if need == 0:
        raise RetinaHeadError(f"nms: k={k} out of range")
    ws = workspace(dev, need, "nms")
    with torch.cuda.device(dev):
        _lib.check(lib.rh_nms(_p(b), _p(s), _p(lab), k, float(iou_threshold), _p(keep), _p(count),
                              _p(ws), ws.numel(), _stream(dev)), "rh_nms")
    kept = int(count.item())
    if kept < 0:
        raise _lib.RetinaHeadError("NMS sweep gave up waiting inside a workgroup (internal error)")
    return keep[:kept]


def nms_static(boxes, scores, iou_threshold, keep, count, labels=None):
    """``rh_nms_static``: the NMS of :func:`nms` with nothing allocated anywhere -- the caller
    brings the outputs (``keep`` int64 [K], ``count`` int32 [1]; the first ``count`` entries of
    ``keep`` are valid) and the library's temporaries sit in a fixed per-GPU array, so the call
    can be used under stream capture or where device allocation is forbidden.  At most
    ``_lib.NMS_STATIC_MAX_BOXES`` boxes; calls that overlap on one GPU race.  Inputs must
    already be float32 / int32, contiguous and 16-byte aligned (nothing is converted or copied)
    and nothing is read back: returns ``(keep, count)``."""
    dev = _dev_check(boxes, scores, labels, keep, count)
    s = scores.reshape(-1)
    k = boxes.shape[0]
    for t, dt in ((boxes, torch.float32), (s, torch.float32), (labels, torch.int32),
                  (keep, torch.int64), (count, torch.int32)):
        if t is not None and (t.dtype != dt or not t.is_contiguous()):
            raise RetinaHeadError("nms_static needs contiguous float32 boxes / scores, int32 labels, "
                                  "int64 keep and int32 count")
    if boxes.dim() != 2 or boxes.shape[1] != 4 or s.shape[0] != k or keep.numel() < k \
            or count.numel() < 1 or (labels is not None and labels.numel() != k):
        raise RetinaHeadError("nms_static expects boxes [K,4], scores [K], keep [>=K], count [1]")
    with _on_device(dev) as stream:
        _lib.check(_lib.load().rh_nms_static(_p(boxes), _p(s), _p(labels), k, float(iou_threshold),
                                             _p(keep), _p(count), stream), "rh_nms_static")
    return keep, count


# Largest post-processing workspace a call may ask for.  The NMS keeps a predecessor bitmap of
# boxes x boxes / 8 bytes per image, so tens of thousands of boxes per image reaching the NMS
# (no pre_nms_topk, a low score threshold on an untrained model) would need gigabytes.
MAX_POST_WORKSPACE_BYTES = 2 << 30


def _check_post_workspace(need, bsz, box_cap):
    if need > MAX_POST_WORKSPACE_BYTES:
        raise RetinaHeadError(
            f"post-processing would need {need / 2**30:.1f} GiB of workspace: up to {box_cap} boxes per "
            f"image reach the NMS, whose bitmap grows with the square of that ({bsz} images).  Pass "
            f"pre_nms_topk (a few thousand loses nothing in practice) or raise the score threshold; "
            f"retinahead.functional.MAX_POST_WORKSPACE_BYTES lifts the limit")


def postprocess_raw(classifications, regressions, anchors_t, height, width, score_thresh=0.05,
                    iou_thresh=0.5, per_class=False, pre_nms_topk=0, max_det=0, cand_cap=4096,
                    stages=_lib.POST_ALL, out=None, from_logits=False, ws=None):
    """One pass of the batched post-processing kernels; no host synchronisation.
    ``out`` carries the output buffers of an earlier call; ``ws`` (a uint8 device tensor of at least
    ``out["workspace_bytes"]`` bytes) replaces the per-stream cached workspace, which callers
    need when they run the stages of one batch on different streams.

    Returns dict(scores [B,cap], labels [B,cap] i64, boxes [B,cap,4], counts [3,B] i32) where
    counts rows are (kept, candidates over threshold, candidates dropped for lack of room).
    """
    dev = _dev_check(classifications, regressions, anchors_t)
    cls, reg, anc = _f32c(classifications), _f32c(regressions), _anchors2d(anchors_t)
    if cls.dim() != 3 or reg.dim() != 3 or reg.shape[2] != 4 or cls.shape[:2] != reg.shape[:2]:
        raise RetinaHeadError("expected classifications [B,N,C] and regressions [B,N,4]")
    if anc.shape[0] != cls.shape[1]:
        raise RetinaHeadError(f"{anc.shape[0]} anchors but {cls.shape[1]} prediction rows")
    bsz, n, c = cls.shape
    cand_cap = max(1, min(int(cand_cap), n))
    topk = int(pre_nms_topk or 0)
    max_det = int(max_det or 0)
    box_cap = min(cand_cap, topk) if topk > 0 else cand_cap
    out_cap = max_det if max_det > 0 else box_cap
    params = _lib.PostParams(float(height), float(width), float(score_thresh), float(iou_thresh),
                             int(bool(per_class)), topk, max_det, cand_cap, int(bool(from_logits)))
    lib = _lib.load()
    if out is None:
        out = dict(
            scores=torch.empty((bsz, out_cap), dtype=torch.float32, device=dev),
            labels=torch.empty((bsz, out_cap), dtype=torch.int64, device=dev),
            boxes=torch.empty((bsz, out_cap, 4), dtype=torch.float32, device=dev),
            counts=torch.zeros((3, bsz), dtype=torch.int32, device=dev),
        )
    need = lib.rh_postprocess_workspace_bytes(bsz, n, c, C.byref(params))
    if need == 0:
        raise RetinaHeadError("postprocess: " + _lib.last_error())
    _check_post_workspace(need, bsz, box_cap)
    out["workspace_bytes"] = int(need)
    if ws is None:
        ws = workspace(dev, need, "post")
    elif ws.numel() < need or ws.dtype != torch.uint8 or ws.device != dev:
        raise RetinaHeadError(f"workspace must be a uint8 tensor of at least {need} bytes on {dev}")
    cnt = out["counts"]
    with torch.cuda.device(dev):
        _lib.check(lib.rh_postprocess(
            _p(cls), _p(reg), _p(anc), bsz, n, c, C.byref(params), _p(out["scores"]),
            _p(out["labels"]), _p(out["boxes"]), _p(cnt[0]), _p(cnt[1]), _p(cnt[2]), _p(ws),
            ws.numel(), int(stages), _stream(dev)), "rh_postprocess")
    return out


def _level_shapes(cls_levels, reg_levels, anc):
    """Checks per-level head tensors [B, A*C, H, W] / [B, A*4, H, W]; returns (B, A, C, heights, widths)."""
    if len(cls_levels) != len(reg_levels) or not cls_levels:
        raise RetinaHeadError("need as many classification as regression levels (at least one)")
    bsz = cls_levels[0].shape[0]
    if any(t.dim() != 4 or t.shape[0] != bsz for t in list(cls_levels) + list(reg_levels)):
        raise RetinaHeadError("level tensors must be [B, channels, H, W]")
    hs = [int(t.shape[2]) for t in cls_levels]
    wsz = [int(t.shape[3]) for t in cls_levels]
    cells = sum(h * w for h, w in zip(hs, wsz))
    if anc.shape[0] % cells:
        raise RetinaHeadError(f"{anc.shape[0]} anchors do not divide over {cells} cells")
    a = anc.shape[0] // cells
    if cls_levels[0].shape[1] % a:
        raise RetinaHeadError(f"{cls_levels[0].shape[1]} classification channels with {a} anchors per cell")
    c = cls_levels[0].shape[1] // a
    for tc, tr, h, w in zip(cls_levels, reg_levels, hs, wsz):
        if tuple(tc.shape) != (bsz, a * c, h, w) or tuple(tr.shape) != (bsz, a * 4, h, w):
            raise RetinaHeadError("inconsistent level shapes")
    return bsz, a, c, hs, wsz


def postprocess_levels_raw(cls_levels, reg_levels, anchors_t, height, width, score_thresh=0.05,
                           iou_thresh=0.5, per_class=False, pre_nms_topk=0, max_det=0, cand_cap=4096,
                           stages=_lib.POST_ALL, out=None, from_logits=False):
    """:func:`postprocess_raw` on the heads' per-level outputs ``[B, A*C, H_l, W_l]`` /
    ``[B, A*4, H_l, W_l]`` (no permute / view / cat, models/retinanet/model.py:152-154,196-202,
    311-313).  Same outputs as the flat form."""
    dev = _dev_check(*cls_levels, *reg_levels, anchors_t)
    cls = [_f32c(t) for t in cls_levels]
    reg = [_f32c(t) for t in reg_levels]
    anc = _anchors2d(anchors_t)
    bsz, a, c, hs, wsz = _level_shapes(cls, reg, anc)
    n = anc.shape[0]
    nl = len(cls)
    cand_cap = max(1, min(int(cand_cap), n))
    topk = int(pre_nms_topk or 0)
    max_det = int(max_det or 0)
    box_cap = min(cand_cap, topk) if topk > 0 else cand_cap
    out_cap = max_det if max_det > 0 else box_cap
    params = _lib.PostParams(float(height), float(width), float(score_thresh), float(iou_thresh),
                             int(bool(per_class)), topk, max_det, cand_cap, int(bool(from_logits)))
    lib = _lib.load()
    if out is None:
        out = dict(
            scores=torch.empty((bsz, out_cap), dtype=torch.float32, device=dev),
            labels=torch.empty((bsz, out_cap), dtype=torch.int64, device=dev),
            boxes=torch.empty((bsz, out_cap, 4), dtype=torch.float32, device=dev),
            counts=torch.zeros((3, bsz), dtype=torch.int32, device=dev),
        )
    ptrs = lambda ts: (C.c_void_p * nl)(*[t.data_ptr() for t in ts])
    ints = lambda vs: (C.c_int32 * nl)(*vs)
    h_arr, w_arr = ints(hs), ints(wsz)
    need = lib.rh_postprocess_levels_workspace_bytes(bsz, nl, h_arr, w_arr, a, c, C.byref(params))
    if need == 0:
        raise RetinaHeadError("postprocess_levels: " + _lib.last_error())
    _check_post_workspace(need, bsz, box_cap)
    ws = workspace(dev, need, "post_levels")
    cnt = out["counts"]
    with torch.cuda.device(dev):
        _lib.check(lib.rh_postprocess_levels(
            ptrs(cls), ptrs(reg), _p(anc), bsz, nl, h_arr, w_arr, a, c, C.byref(params),
            _p(out["scores"]), _p(out["labels"]), _p(out["boxes"]), _p(cnt[0]), _p(cnt[1]), _p(cnt[2]),
            _p(ws), ws.numel(), int(stages), _stream(dev)), "rh_postprocess_levels")
    return out
