"""Data-parallel use of the dense-head loss: one process per GPU, images sharded.

Every quantity of the loss is per image (models/retinanet/losses.py:43-136); the
only cross-image step is the batch mean (losses.py:139-140).  So a rank runs the
fused kernels on its own images with ``batch_divisor`` = global batch size, which
already scales its losses and gradients by 1/B_global, and ONE all-reduce (sum) of
a 4-float vector (classification loss, regression loss, number of positive anchors,
number of images) over RCCL/xGMI yields the global losses and the global counts.
Gradients with respect to the predictions need no exchange.  The reference's own multi-GPU mechanism is
nn.DataParallel (train.py:137).

With ``FocalLoss(normalizer='global')`` the normaliser itself is global: each rank runs the
assignment stage, ONE more all-reduce sums the positive counts over all ranks before the
streaming stage scales losses and gradients by it, and the ranks' losses are then shares that the
loss all-reduce adds up.
"""
from __future__ import annotations

import functools

import torch
import torch.distributed as dist_


def shard_bounds(global_batch, rank, world_size):
    """Contiguous image range [lo, hi) of ``rank``; the first ranks take the remainder."""
    if not 0 <= rank < world_size:
        raise ValueError("rank outside world")
    base, rem = divmod(int(global_batch), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_reduce_sum_(t, group=None):
    """In-place sum of ``t`` over the ranks of ``group``.  RCCL (backend "nccl") reduces device
    tensors directly; a host-side backend (gloo: CPU tests, several ranks rehearsing on one
    card) gets a device tensor staged through the host."""
    if t.is_cuda and dist_.get_backend(group) != "nccl":
        host = t.detach().cpu()
        dist_.all_reduce(host, op=dist_.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist_.all_reduce(t, op=dist_.ReduceOp.SUM, group=group)
    return t


class _GlobalValue(torch.autograd.Function):
    """Forward: the value summed over all ranks; backward: the gradient goes to the local share
    unchanged (each rank back-propagates its own images).  No launch either way."""

    @staticmethod
    def forward(ctx, local, value):
        return value.view_as(local)

    @staticmethod
    def backward(ctx, g):
        return g, None


def all_reduce_losses(local_losses, group=None):
    """Sums the loss vectors (any length; entries without grad are plain sums) of all ranks.
    The value is the global sum, the gradient is that of the local share (each rank
    back-propagates its own images)."""
    if not (dist_.is_available() and dist_.is_initialized()) or dist_.get_world_size(group) == 1:
        return local_losses
    total = all_reduce_sum_(local_losses.detach().clone(), group)
    return _GlobalValue.apply(local_losses, total)


def all_reduce_num_pos(num_pos, group=None):
    """float32 [1]: positives over all ranks from this rank's per-image counts (exact below
    2^24)."""
    total = num_pos.sum(dtype=torch.float32).reshape(1)
    if dist_.is_available() and dist_.is_initialized() and dist_.get_world_size(group) > 1:
        all_reduce_sum_(total, group)
    return total


class DataParallelFocalLoss(torch.nn.Module):
    """Wraps a loss callable with the FocalLoss signature for sharded batches.

    ``loss_fn(classifications, regressions, anchors, annotations)`` must return the
    two-entry dict of losses computed with ``batch_divisor = global_batch`` (for
    ``retinahead.FocalLoss`` pass ``batch_divisor`` at construction).
    """

    def __init__(self, loss_fn, group=None):
        super().__init__()
        self.loss_fn = loss_fn
        self.group = group
        cfg = getattr(loss_fn, "cfg", None)
        if isinstance(cfg, dict) and cfg.get("normalizer") == "global":
            cfg["reduce_total"] = functools.partial(all_reduce_num_pos, group=group)

    def forward(self, classifications, regressions, anchors, annotations):
        out = self.loss_fn(classifications, regressions, anchors, annotations)
        cls_l, reg_l = out["classification_loss"], out["regression_loss"]
        last = getattr(self.loss_fn, "last", None)
        if not (dist_.is_available() and dist_.is_initialized()) or dist_.get_world_size(self.group) == 1:
            if last is not None and "exchange" in last:
                self.last_global = {"num_pos": last["exchange"][2], "images": last["exchange"][3]}
            return out
        if last is not None and "exchange" in last:
            # the finalize kernel left (both losses, positive anchors, images) side by side
            vec = last["exchange"].clone()
        else:
            vec = torch.cat([cls_l.detach().reshape(1), reg_l.detach().reshape(1), cls_l.new_zeros(2)])
        all_reduce_sum_(vec, self.group)
        # global counts of the step, device tensors: positive anchors and images over all ranks
        self.last_global = {"num_pos": vec[2], "images": vec[3]}
        return {"classification_loss": _GlobalValue.apply(cls_l.reshape(1), vec[0:1]),
                "regression_loss": _GlobalValue.apply(reg_l.reshape(1), vec[1:2])}
