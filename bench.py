#!/usr/bin/env python3
"""Benchmark of the dense-head hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one fused focal-loss forward+backward (FocalLoss module call + backward)
over one synthetic batch of the COCO-80 608 px configuration: B=16 images per GPU,
N=69354 anchors, C=80 classes, up to 20 ground-truth boxes per image.  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

The timed block (exactly --steps steps between two barrier + synchronize fences) is repeated
--repeats times; ms_per_step and value come from the median block (max over ranks per block),
ms_per_step_p10_p90 gives the spread.

Besides the headline metric the line carries
  roofline      the dominant kernel (focal_stream_kernel): algorithmic bytes
                B*N*(8C+48) over its mean duration from HIP events on the launch stream
  cpu_baseline  a torch-CPU restatement of the reference loss (oracle/torch_cpu_path.py)
                timed on this host on a bounded sample of the same workload
  decode_nms    images/s of the batched decode + score threshold + sort + NMS path
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "pytorch-retinanet_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, spec
SIZE, BATCH, CLASSES, MAX_GT = 608, 16, 80, 20


def synth_train(rng, bsz, n, c, size, max_gt):
    """Training-like inputs: probabilities sigmoid(N(-4.595, 2^2)) (prior 0.01), regressions N(0,1),
    1..max_gt boxes per image with log-uniform sides in [16, size/2], padded rows -1."""
    z = rng.standard_normal((bsz, n, c), dtype=np.float32) * 2.0 - 4.595
    cls = (1.0 / (1.0 + np.exp(-z))).astype(np.float32)
    reg = rng.standard_normal((bsz, n, 4), dtype=np.float32)
    ann = -np.ones((bsz, max_gt, 5), np.float32)
    for j in range(bsz):
        v = int(rng.integers(1, max_gt + 1))
        cx, cy = rng.uniform(0, size, v), rng.uniform(0, size, v)
        w = np.exp(rng.uniform(np.log(16), np.log(size / 2), v))
        h = np.exp(rng.uniform(np.log(16), np.log(size / 2), v))
        x1, y1 = np.clip(cx - w / 2, 0, size - 1), np.clip(cy - h / 2, 0, size - 1)
        x2, y2 = np.clip(cx + w / 2, x1 + 1, size), np.clip(cy + h / 2, y1 + 1, size)
        ann[j, :v, :4] = np.stack([x1, y1, x2, y2], 1)
        ann[j, :v, 4] = rng.integers(0, c, v)
    return cls, reg, ann


def synth_infer(rng, bsz, n, c, mu=-7.0):
    """Inference-like inputs: sigmoid(N(mu, 1.2^2)); mu = -7 leaves about 2000 anchors per image
    over 0.05 at 608 px with 80 classes (with 600 classes, -8 does)."""
    z = rng.standard_normal((bsz, n, c), dtype=np.float32) * 1.2 + np.float32(mu)
    return (1.0 / (1.0 + np.exp(-z))).astype(np.float32), \
        (rng.standard_normal((bsz, n, 4), dtype=np.float32) * 0.5)


def kernel_source_hash():
    """sha256 over the loss kernels' sources: ties the committed PMC traffic figure to the code."""
    import hashlib
    h = hashlib.sha256()
    for name in ("rh_loss.hip", "rh_common.h"):
        with open(os.path.join(ROOT, "pytorch-retinanet_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def event_times(fn, iters):
    """Mean / min milliseconds of fn() from one HIP event pair per call on the current stream."""
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
             for _ in range(iters)]
    for a, b in pairs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in pairs]
    event_times.last = ms
    return float(np.mean(ms)), float(np.min(ms))


def event_train(fn, launches, reps):
    """Milliseconds per launch from event pairs that each bracket `launches` back-to-back calls
    (mean over reps, and the best rep): the fixed cost of an event pair around a single ~0.1 ms
    kernel (~8 us here) would otherwise be charged to the kernel."""
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
             for _ in range(reps)]
    for a, b in pairs:
        a.record()
        for _ in range(launches):
            fn()
        b.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) / launches for a, b in pairs]
    event_train.last = ms
    return float(np.mean(ms)), float(np.min(ms))


def cpu_baseline(cls, reg, anc, ann, sample_images, threads, budget_s):
    """anchors/s of the torch-CPU port on `sample_images` images: best run within `budget_s`
    seconds of CPU work (at least three runs)."""
    from oracle import torch_cpu_path as T
    torch.set_num_threads(threads)
    c = torch.from_numpy(cls[:sample_images]).clone().requires_grad_(True)
    r = torch.from_numpy(reg[:sample_images]).clone().requires_grad_(True)
    a, an = torch.from_numpy(anc), torch.from_numpy(ann[:sample_images])
    best, runs, began = float("inf"), 0, time.perf_counter()
    while runs < 3 or time.perf_counter() - began < budget_s:
        c.grad = r.grad = None
        t0 = time.perf_counter()
        out = T.dense_head_loss(c, r, a, an)
        (out["classification_loss"] + out["regression_loss"]).sum().backward()
        best = min(best, time.perf_counter() - t0)
        runs += 1
    # the port's results, for the parity check of the measured path on these same inputs
    cpu_baseline.last = dict(cls_loss=float(out["classification_loss"].detach()), reg_loss=float(out["regression_loss"].detach()),
                             d_cls=c.grad.numpy(), d_reg=r.grad.numpy(), runs=runs)
    return sample_images * cls.shape[1] / best, best


def cpu_baseline_post(cls, reg, anc, size, sample_images, threads, budget_s):
    """images/s of the torch-CPU port of decode + threshold + NMS on `sample_images` images."""
    from oracle import torch_cpu_path as T
    torch.set_num_threads(threads)
    c, r, a = torch.from_numpy(cls[:sample_images]), torch.from_numpy(reg[:sample_images]), torch.from_numpy(anc)
    best, runs, began = float("inf"), 0, time.perf_counter()
    with torch.no_grad():
        while runs < 3 or time.perf_counter() - began < budget_s:
            t0 = time.perf_counter()
            T.postprocess(c, r, a, size, size)
            best = min(best, time.perf_counter() - t0)
            runs += 1
    cpu_baseline_post.runs = runs
    return sample_images / best, best


class RankExchange:
    """The one exchange of a data-parallel step: the all-reduce of the 4-float vector (loss sums,
    positive anchors, images) that each rank's launch leaves behind.  RCCL: asynchronous, with a
    bounded backlog; rehearsal backends (gloo): blocking, through the host."""

    def __init__(self, world, backend):
        import torch.distributed as dist
        self.dist, self.world, self.backend, self.pending, self.last = dist, world, backend, [], None

    def submit(self, vec):
        if self.world <= 1:
            return
        vec = vec.clone()                    # a graph replay rewrites its own copy every step
        if self.backend == "nccl":
            self.pending.append((vec, self.dist.all_reduce(vec, op=self.dist.ReduceOp.SUM, async_op=True)))
            if len(self.pending) > 4:        # bounded backlog; values are the global losses
                self.last = self.pending[0][0]
                self.pending.pop(0)[1].wait()
        else:
            host = vec.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM)
            self.last = host

    def fence(self, sync):
        """Everything issued so far is done on every rank: drain, synchronise, barrier, synchronise."""
        while self.pending:
            self.last = self.pending[0][0]
            self.pending.pop(0)[1].wait()
        sync()
        if self.world > 1:
            self.dist.barrier()
        sync()


def count_ranks(world, backend, dev):
    """How many ranks take part, by an all-reduce of ones."""
    if world <= 1:
        return 1
    import torch.distributed as dist
    ones = torch.ones(1, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    return int(ones.item())


def timed_blocks(step, exchange, sync, steps, warmup, repeats, world, backend, dev):
    """`warmup` untimed steps, then `repeats` blocks of exactly `steps` steps, each block between
    two fences; returns the block times in seconds, per block the slowest rank."""
    for _ in range(warmup):
        step()
    exchange.fence(sync)
    blocks = []
    for _ in range(max(repeats, 1)):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        exchange.fence(sync)
        blocks.append(time.perf_counter() - t0)
    blocks = torch.tensor(blocks, dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist
        t = blocks.to(dev) if backend == "nccl" else blocks
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        blocks = t.cpu()
    return blocks.numpy()


def require_ranks(ranks_seen, gpus, devices, backend):
    """A scaling run must be what it says: one rank per GPU, all of them there."""
    if ranks_seen != gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but the all-reduce saw {ranks_seen} ranks")
    if gpus > 1 and backend == "nccl" and devices < gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but this node shows {devices} GPU(s): ranks would "
                         "share a card (RH_BENCH_BACKEND=gloo rehearses that on purpose)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=30, help="timed blocks of --steps steps each")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--post-streams", type=int, default=3,
                    help="batches in flight (one HIP stream each) in the pipelined decode+NMS leg")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="every step runs its own IoU assignment in line (no side branch)")
    ap.add_argument("--eager", action="store_true", help="time the eager Python path, no hipGraph")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks "
                         f"(WORLD_SIZE={world})")
    import torch.distributed as dist
    # one rank per GPU; RH_BENCH_BACKEND=gloo lets several ranks share one card to rehearse the
    # multi-rank path on a one-GPU box (RCCL refuses two ranks on the same device)
    backend = os.environ.get("RH_BENCH_BACKEND", "nccl")
    devices = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and devices < world:
        raise SystemExit(f"bench.py: --gpus {world} but this node shows {devices} GPU(s): ranks would "
                         "share a card (RH_BENCH_BACKEND=gloo rehearses that on purpose)")
    dev_index = local_rank % max(devices, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import retinahead as rh
    from retinahead import _lib, functional as F
    rh.load_library()
    verbose = bool(os.environ.get("RH_BENCH_VERBOSE"))

    def note(msg):
        if verbose:
            print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)

    img = torch.zeros(1, 3, SIZE, SIZE, device=dev)
    anchors = rh.Anchors()(img)
    n = anchors.shape[1]
    rng = np.random.default_rng(2000 + rank)
    cls_np, reg_np, ann_np = synth_train(rng, BATCH, n, CLASSES, SIZE, MAX_GT)
    ann = torch.from_numpy(ann_np).to(dev)
    # Two sets of predictions that take turns (a training loop never streams the same 355 MB
    # twice in a row): the second set is the first with its images rotated by one, so that the
    # numbers of the CPU leg below, which are for set 0, stay what they were.
    cls_sets = [torch.from_numpy(cls_np).to(dev)]
    reg_sets = [torch.from_numpy(reg_np).to(dev)]
    cls_sets.append(torch.roll(cls_sets[0], 1, 0).contiguous())
    reg_sets.append(torch.roll(reg_sets[0], 1, 0).contiguous())
    for t in cls_sets + reg_sets:
        t.requires_grad_(True)
    cls, reg = cls_sets[0], reg_sets[0]

    prefetch = not args.no_prefetch
    sync = lambda: torch.cuda.synchronize(dev)

    def make_runner(n_sets, eager=False, announce=prefetch):
        """A step() that runs the module's forward + backward on input set (step index % n_sets).

        The step is a handful of kernels (~130 us of GPU work), which eager PyTorch cannot enqueue
        fast enough (autograd + dispatcher cost ~160 us of host time per step): forward + backward
        are captured into hipGraphs and replayed.  Two graphs take turns: with the assignment
        running ahead there are two workspaces in turn (A: step on workspace X, next assignment
        into Y; B: the reverse), and graph k reads input set k % n_sets."""
        # Each rank's losses and gradients are scaled by 1/B_global inside the kernels
        # (batch_divisor); the only exchange is the all-reduce of a 4-float vector (loss sums,
        # positive anchors, images).
        mod = rh.FocalLoss(batch_divisor=BATCH * world)

        def local_step(k):
            c, r = cls_sets[k % n_sets], reg_sets[k % n_sets]
            c.grad = None
            r.grad = None
            if announce:
                # the NEXT step's IoU assignment (it needs the annotations only, which a loader has
                # ahead of time; here every batch carries the same ones) rides on this step's
                # [B,N,C] stream launch; this step uses the assignment the previous step carried.
                # One assignment per step either way.
                mod.prefetch_assignment(anchors, ann, CLASSES)
            out = mod(c, r, anchors, ann)
            loss = sum(l for l in out.values())      # train.py:185
            loss.backward()                          # train.py:190
            # what the ranks exchange: loss sums, positive anchors, images (one 16-byte
            # all-reduce); the finalize stage leaves the four floats side by side
            out = dict(out)
            out["exchange"] = mod.last["exchange"]
            return out

        side = torch.cuda.Stream(device=dev)
        graphs, outs = [], []
        if not eager:
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for k in range(4):
                    local_step(k)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            for k in range(2):
                g = torch.cuda.CUDAGraph()
                # thread_local: calls made meanwhile by other threads of the process (the
                # collective library's watchdog polls events) cannot invalidate the capture
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    outs.append(local_step(k))
                graphs.append(g)
        turn = [0]

        def step():
            k = turn[0]
            turn[0] += 1
            if graphs:
                graphs[k % 2].replay()
                out = outs[k % 2]
            else:
                out = local_step(k)
            exchange.submit(out["exchange"])
            return out
        step.local_step, step.side, step.graphs = local_step, side, graphs
        return step

    exchange = RankExchange(world, backend)
    launch = "eager"
    try:
        step = make_runner(2, eager=args.eager)
        if step.graphs:
            launch = ("hipGraph replay of FocalLoss forward + backward"
                      + (", next step's assignment inside the stream launch" if prefetch else "")
                      + "; two graphs in turn, each on its own set of predictions")
    except Exception as e:                       # pragma: no cover - depends on the runtime
        if world > 1:
            raise
        print(f"[bench] graph capture failed ({e!r}); timing the eager path", file=sys.stderr)
        torch.cuda.synchronize()
        step = make_runner(2, eager=True)

    ranks_seen = count_ranks(world, backend, dev)
    require_ranks(ranks_seen, args.gpus, devices, backend)

    note(f"launch mode: {launch}")
    blocks = timed_blocks(step, exchange, sync, args.steps, args.warmup, args.repeats, world, backend, dev)
    note(f"timed blocks done: median {np.median(blocks):.5f} s")
    # The same steps issued eagerly (plain Python calls of the module and of backward, no graph),
    # by the same protocol: between two graph launches the GPU idles for a few microseconds, between
    # two eager launches of one stream it does not, and since the host side of an eager step costs
    # less than the step takes on the GPU the eager loop can be the faster one.  Both are whole
    # executions of the step; the headline is the faster, and both are in the line.
    modes = {"graph" if step.graphs else "eager": blocks}
    if step.graphs:
        eager_step = make_runner(2, eager=True)
        with torch.cuda.stream(step.side):
            modes["eager"] = timed_blocks(eager_step, exchange, sync, args.steps, args.warmup, args.repeats,
                                          world, backend, dev)
        if float(np.median(modes["eager"])) < float(np.median(blocks)):
            blocks = modes["eager"]
            launch = ("eager calls of FocalLoss forward + backward on one stream"
                      + (", next step's assignment inside the stream launch" if prefetch else "")
                      + "; two sets of predictions in turn")
    elapsed = float(np.median(blocks))
    block_p10, block_p90 = (float(np.percentile(blocks, q)) for q in (10, 90))
    anchors_per_s = world * BATCH * n * args.steps / elapsed

    def replay_ms(fn, per_call=1):
        """Median milliseconds per step of fn() replayed back to back (rank-local, short)."""
        times = []
        for _ in range(12):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(max(args.steps // per_call, 1)):
                fn()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t1) / (per_call * max(args.steps // per_call, 1)) * 1e3)
        return float(np.median(times[2:]))

    # the same with ONE set of predictions streamed every step (what this benchmark measured
    # before): beside the headline so that any help from the caches shows
    same_input_ms = None
    if step.graphs and world == 1:
        try:
            one = make_runner(1)
            same_input_ms = replay_ms(one)
        except Exception as e:                   # pragma: no cover
            print(f"[bench] single-input graphs failed ({e!r})", file=sys.stderr)

    # eager path beside it (rank-local, short), for the record
    eager_ms = None
    if step.graphs:
        with torch.cuda.stream(step.side):        # the stream the tensors' autograd nodes were made on
            for k in range(5):
                step.local_step(k)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(50):
                step.local_step(k)
            torch.cuda.synchronize()
            eager_ms = (time.perf_counter() - t1) / 50 * 1e3

    # the same step with every assignment in a launch of its own (no announcement of the next
    # batch's annotations): what a loop that cannot look ahead gets
    plain_ms, plain_by_mode = None, None
    if step.graphs and prefetch and world == 1:
        try:
            plain_by_mode = {"graph": replay_ms(make_runner(2, announce=False))}
            plain_eager = make_runner(2, eager=True, announce=False)
            with torch.cuda.stream(step.side):
                plain_by_mode["eager"] = replay_ms(plain_eager)
            plain_ms = min(plain_by_mode.values())
        except Exception as e:                   # pragma: no cover - depends on the runtime
            print(f"[bench] plain-step graph failed ({e!r})", file=sys.stderr)

    # two consecutive steps (A then B) captured as ONE graph: between two graph launches the GPU
    # idles about 8 us (the marker that ends a launch), inside a graph the next kernel starts when
    # the previous one ends -- what the path costs inside a training iteration captured whole
    pair_ms = None
    if step.graphs and prefetch and world == 1:
        try:
            for t in cls_sets + reg_sets:
                t.grad = None
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=step.side, capture_error_mode="thread_local"):
                step.local_step(0)
                step.local_step(1)
            pair_ms = replay_ms(g2.replay, per_call=2)
        except Exception as e:                   # pragma: no cover - depends on the runtime
            print(f"[bench] two-step graph failed ({e!r})", file=sys.stderr)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    anc_np = anchors.cpu().numpy()
    parity = None     # set by the CPU-baseline leg below: the port's results on these inputs

    # ---- roofline of the dominant kernel: the N x C stream -----------------------------------
    with torch.no_grad():
        det = [(c.detach(), r.detach()) for c, r in zip(cls_sets, reg_sets)]
        bufs = F.focal_loss_fwd_bwd(det[0][0], det[0][1], anchors, ann)   # fills the workspace
        flip = [0]

        def stream_fn(**kw):
            # consecutive launches take the two sets of predictions in turn (same annotations, so
            # the workspace's assignment serves both)
            c, r = det[flip[0] % 2]
            flip[0] += 1
            return F.focal_loss_fwd_bwd(c, r, anchors, ann, stages=_lib.STAGE_STREAM, out=bufs, **kw)
        for _ in range(10):
            stream_fn()
        k1_mean, k1_min = event_times(stream_fn, 100)
        plain_mean, plain_min = event_train(stream_fn, 20, 10)
        carried_fn = stream_fn
        if prefetch:
            # the launch as the timed steps issue it: with the next step's assignment workgroups
            # at the end of its grid
            next_ws = F.focal_workspace(BATCH, n, CLASSES, ann.shape[1], dev)
            next_out = F.new_small_outputs(BATCH, dev)
            carried_fn = lambda: stream_fn(ahead=(ann, next_ws, next_out))
            for _ in range(10):
                carried_fn()
        k_mean, k_min = event_train(carried_fn, 20, 10)
        k_p10, k_p50, k_p90 = (float(np.percentile(event_train.last, q)) for q in (10, 50, 90))
        assign_fn = lambda: F.focal_loss_fwd_bwd(det[0][0], det[0][1], anchors, ann,
                                                 stages=_lib.STAGE_ASSIGN, out=bufs)
        a_mean, _ = event_times(assign_fn, 50)
        all_fn = lambda: F.focal_loss_fwd_bwd(det[0][0], det[0][1], anchors, ann, out=bufs)
        f_mean, _ = event_times(all_fn, 100)
    algo_bytes = BATCH * n * (8 * CLASSES + 48)
    achieved = algo_bytes / (k_mean * 1e-3) / 1e9
    # HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of this same
    # configuration; they cannot be read inside this process, so the committed summary is used --
    # but only while it names the kernel sources it was measured on.
    traffic, traffic_note = None, "no PMC summary"
    tfile = os.path.join(ROOT, "profiles", "traffic_focal_stream.json")
    variant = ", 1>" if prefetch else ", 0>"      # focal_stream_kernel<U, logits, form of the assignment it carries>
    if os.path.exists(tfile):
        rec = json.load(open(tfile))
        if rec.get("source_sha256") == kernel_source_hash():
            for name, k in rec.get("kernels", {}).items():
                if name.endswith(variant) and ", false, " in name:
                    traffic, traffic_note = k["hbm_bytes_per_launch"], rec.get("how", "rocprofv3 PMC")
        else:
            traffic_note = "PMC summary is of other kernel sources: not reported"

    # ---- the loss on logits and on the heads' own layout (same bytes, same timing method) ----
    with torch.no_grad():
        shapes = [((SIZE + 2 ** l - 1) // 2 ** l,) * 2 for l in (3, 4, 5, 6, 7)]
        logits = torch.logit(cls.detach().double()).float().contiguous()

        def to_levels(flat, last):
            res, at = [], 0
            for (h, w) in shapes:
                k = h * w * 9
                res.append(flat[:, at:at + k].reshape(BATCH, h, w, 9 * last).permute(0, 3, 1, 2).contiguous())
                at += k
            return res
        lg_bufs = F.focal_loss_fwd_bwd(logits, reg.detach(), anchors, ann, from_logits=True)
        lg_mean, _ = event_train(lambda: F.focal_loss_fwd_bwd(
            logits, reg.detach(), anchors, ann, stages=_lib.STAGE_STREAM, out=lg_bufs, from_logits=True), 20, 5)
        cls_lv, reg_lv = to_levels(logits, CLASSES), to_levels(reg.detach(), 4)
        lv_bufs = F.focal_loss_levels_fwd_bwd(cls_lv, reg_lv, anchors, ann, from_logits=True)
        lv_mean, _ = event_train(lambda: F.focal_loss_levels_fwd_bwd(
            cls_lv, reg_lv, anchors, ann, stages=_lib.STAGE_STREAM, out=lv_bufs, from_logits=True), 20, 5)
        del lg_bufs, lv_bufs, cls_lv, reg_lv, logits

    # ---- decode + NMS (second half of the metric) ------------------------------------------
    # Input sets: every batch in flight has predictions of its own, and the single-stream loop
    # takes two sets in turn -- a serving loop never reads the same 355 MB twice, and with one
    # shared tensor a trailing scan could be fed from the caches by the one ahead of it.  The
    # old figures on ONE shared tensor are measured beside them.
    n_streams = args.post_streams
    icls_np, ireg_np = synth_infer(np.random.default_rng(3000), BATCH, n, CLASSES)
    n_sets = max(n_streams + 1, 2)
    icls_sets = [torch.from_numpy(icls_np).to(dev)]
    ireg_sets = [torch.from_numpy(ireg_np).to(dev)]
    for k in range(1, n_sets):          # the same images in another order: same work per batch
        icls_sets.append(torch.roll(icls_sets[0], k, 0).contiguous())
        ireg_sets.append(torch.roll(ireg_sets[0], k, 0).contiguous())
    icls, ireg = icls_sets[0], ireg_sets[0]

    def post_on(k):
        c, r = icls_sets[k % n_sets], ireg_sets[k % n_sets]
        return lambda out=None: F.postprocess_raw(c, r, anchors, SIZE, SIZE, cand_cap=4096, out=out)
    post = post_on(0)
    res = post()
    counts = res["counts"].cpu().numpy()
    for _ in range(5):
        post(res)
    p_mean, p_min = event_times(lambda: post(res), 200)
    event_times.last_post = list(event_times.last)

    def single_stream(fns, batches=300):
        """ms per batch, batches issued back to back on ONE stream by the host clock (an event pair
        around every call adds a few microseconds of its own to these short kernels), the calls
        taking the given input sets in turn; median of 5 runs and the runs."""
        runs = []
        for _ in range(5):
            for i in range(50):
                fns[i % len(fns)](res)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(batches):
                fns[i % len(fns)](res)
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t1) / batches * 1e3)
        return float(np.median(runs)), runs
    single_ms, singles = single_stream([post_on(0), post_on(1)])
    single_same_ms, _ = single_stream([post])
    d_mean, _ = event_times(lambda: F.postprocess_raw(icls, ireg, anchors, SIZE, SIZE, cand_cap=4096,
                                                      stages=_lib.POST_DECODE, out=res), 50)
    decode_bytes = BATCH * n * (4 * CLASSES + 32)

    # One image, one stream: the call shape of the reference's inference scripts
    # (visualize.py:95, evaluate.py:95, csv_eval.py:89, coco_eval.py:27).  Latency of the whole
    # post-processing call as the GPU sees it (calls issued back to back on one stream: each
    # call's kernels depend on one another, so time per call = the chain's latency), and as a host
    # sees it (call + synchronize).  Two images take turns.
    one_fns = [(lambda c, r: (lambda out=None: F.postprocess_raw(c, r, anchors, SIZE, SIZE, cand_cap=4096,
                                                                 out=out)))(icls_sets[k][:1].contiguous(),
                                                                            ireg_sets[k][:1].contiguous())
               for k in range(2)]
    res1 = one_fns[0]()
    b1_runs = []
    for _ in range(6):
        for i in range(20):
            one_fns[i % 2](res1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(200):
            one_fns[i % 2](res1)
        torch.cuda.synchronize()
        b1_runs.append((time.perf_counter() - t1) / 200 * 1e6)
    b1_latency_us = float(np.median(b1_runs[1:]))
    b1_sync = []
    for i in range(220):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        one_fns[i % 2](res1)
        torch.cuda.synchronize()
        b1_sync.append((time.perf_counter() - t1) * 1e6)
    b1_sync_us = float(np.median(b1_sync[20:]))
    b1_kept = int(res1["counts"][0, 0])

    # Batches in flight on separate streams (each with its own inputs, workspace and outputs): the
    # score scan is bound by HBM, the rank sort, the pair search and the resolve rounds by latency,
    # so consecutive batches overlap well.
    # HIP multiplexes streams onto a few hardware queues (4 per process here), and which queue a
    # stream gets is not visible.  Three streams and the default stream have a queue each.  Four
    # streams are faster when the queues fall well and slower when two of them share one, at random
    # from one trial to the next (tools/stream_sets.py, tools/scan_chain.py), so three is the
    # default, and three trials with one stream more are listed beside it.
    # Five fresh sets of streams are measured in full; the reported rate is their median, and every
    # trial is listed.  (One more set runs first and is not counted: the first pipelined run of a
    # process is 10-15 us per batch slower whatever its streams, and is listed on its own.)
    stream_sets = [[torch.cuda.Stream(device=dev) for _ in range(n_streams)] for _ in range(6)]
    streams = stream_sets[0]

    def piped(fn_of_slot, rounds=100):
        fns = [fn_of_slot(i) for i in range(len(streams))]
        outs = []
        for st, fn in zip(streams, fns):
            st.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(st):
                outs.append(fn())
                fn(outs[-1])
        torch.cuda.synchronize()

        def run(k):
            for _ in range(k):
                for st, fn, o in zip(streams, fns, outs):
                    with torch.cuda.stream(st):
                        fn(o)
            torch.cuda.synchronize()
        # untimed rounds first: the first ~200 batches after a pause on the host run about 20 %
        # slower than the steady state (87 against 72 us per batch, whatever the kernels)
        run(rounds)
        t1 = time.perf_counter()
        run(rounds)
        return (time.perf_counter() - t1) / (rounds * len(streams)) * 1e3

    p_med = float(np.median(event_times.last_post))
    streams = stream_sets[0]
    warmup_trial = piped(post_on)
    trial, trial_same = [], []
    for cand in stream_sets[1:]:
        streams = cand
        trial.append(piped(post_on))
        trial_same.append(piped(lambda i: post))
    piped_ms = float(np.median(trial))
    piped_same_ms = float(np.median(trial_same))
    trial_p10, trial_p90 = (float(np.percentile(trial, q)) for q in (10, 90))
    wider = []
    for _ in range(3):
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams + 1)]
        wider.append(piped(post_on))
    streams = stream_sets[1 + int(np.argsort(trial)[len(trial) // 2])]
    # the same through the public helper (its host read of the counts per batch included); three
    # fresh pipelines (each draws its own streams), the median is reported and all are listed
    pipe_trials = []
    for _ in range(3):
        pipe = rh.PostprocessPipeline(depth=2 * n_streams, streams=n_streams)

        def pipe_run(k):
            for i in range(k):
                if len(pipe) == 2 * n_streams:
                    pipe.results()
                pipe.submit(icls_sets[i % n_sets], ireg_sets[i % n_sets], anchors, (SIZE, SIZE))
            while len(pipe):
                pipe.results()
        pipe_run(100)
        t1 = time.perf_counter()
        pipe_run(400)
        pipe_trials.append((time.perf_counter() - t1) / 400 * 1e3)
    pipe_ms = float(np.median(pipe_trials))
    # the same with only equal labels suppressing each other (batched per-class NMS)
    per_class_ms = piped(lambda i: (lambda out=None: F.postprocess_raw(
        icls_sets[i % n_sets], ireg_sets[i % n_sets], anchors, SIZE, SIZE, per_class=True,
        cand_cap=4096, out=out)), rounds=50)

    line = {
        "metric": "focal_loss_fwd_bwd_anchors_per_sec",
        "value": anchors_per_s,
        "unit": "anchors/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_p10_p90": [block_p10 / args.steps * 1e3, block_p90 / args.steps * 1e3],
        "repeats": len(blocks),
        "timing": f"median of {len(blocks)} timed blocks of {args.steps} steps, each between two "
                  "barrier + synchronize fences, per block the slowest rank",
        "ranks_seen": ranks_seen,
        "backend": ("RCCL (torch.distributed nccl)" if backend == "nccl" else backend) if world > 1 else None,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "COCO-80 608px: B=16 images/GPU, N=69354 anchors, C=80, <=20 GT/image; "
                               "FocalLoss module forward + backward (fused kernels)",
                   "global_batch": BATCH * world, "parallelism": f"dp{world}", "launch": launch,
                   "ms_per_step_by_launch_mode": {k: float(np.median(v)) / args.steps * 1e3 for k, v in modes.items()},
                   "eager_ms_per_step": eager_ms,
                   "ms_per_step_without_announcing_the_next_batch": plain_ms,
                   "ms_per_step_without_announcing_by_launch_mode": plain_by_mode, "ms_per_step_two_steps_per_graph": pair_ms,
                   "inputs": "two sets of predictions take turns (no step streams the tensor the step before it streamed)",
                   "ms_per_step_one_input_set_every_step": same_input_ms},
        "parity_ok": parity,
        "roofline": {"bound": "hbm",
                     "kernel": "focal_stream_kernel<5, false, %s>" % ("1" if prefetch else "0")
                               + (" (the [B,N,C] stream; its grid ends with the next step's assignment"
                                  " workgroups, whose bytes are not counted)" if prefetch else ""),
                     "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_note,
                     "algorithmic_bytes": algo_bytes,
                     "kernel_ms_mean": k_mean, "kernel_ms_min": k_min,
                     "kernel_ms_p10_p50_p90": [k_p10, k_p50, k_p90],
                     "timing": "HIP events around trains of 20 back-to-back launches (two sets of predictions in turn), 10 trains",
                     "stream_alone_kernel_ms_mean": plain_mean,
                     "stream_alone_frac": algo_bytes / (plain_mean * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "single_launch_event_pair_ms_mean": k1_mean,
                     "single_launch_event_pair_ms_min": k1_min},
        "kernels_ms": {"assign_stage": a_mean, "stream_stage": plain_mean, "stream_stage_carrying_next_assignment": k_mean if prefetch else None, "fused_call": f_mean,
                       "stream_stage_from_logits": lg_mean,
                       "stream_stage_head_layout_from_logits": lv_mean},
        "decode_nms": {"images_per_s": BATCH / (piped_ms * 1e-3), "ms_per_batch": piped_ms,
                       "launch": f"{n_streams} batches in flight on {n_streams} HIP streams, each with its own "
                                 f"input tensors; median of {len(trial)} stream sets, each measured over "
                                 f"{100 * n_streams} batches",
                       "ms_per_batch_all_streams_reading_one_tensor": piped_same_ms,
                       "stream_set_trials_ms_per_batch": trial,
                       "uncounted_first_trial_ms_per_batch": warmup_trial,
                       "one_stream_more_trials_ms_per_batch": wider,
                       "ms_per_batch_p10_p90": [trial_p10, trial_p90],
                       "frac_of_hbm_peak": decode_bytes / (piped_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "single_stream_images_per_s": BATCH / (single_ms * 1e-3),
                       "single_stream_ms_per_batch": single_ms,
                       "single_stream_trials_ms_per_batch": singles,
                       "single_stream_frac_of_hbm_peak": decode_bytes / (single_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "single_stream_timing": "host clock over 300 batches issued back to back on one stream, two "
                                               "input sets in turn, median of 5",
                       "single_stream_ms_per_batch_one_tensor_every_batch": single_same_ms,
                       "b1_latency_us": b1_latency_us,
                       "b1_latency_runs_us": b1_runs[1:],
                       "b1_latency_timing": "one 608 px image per call, 200 calls back to back on one stream (two "
                                            "images in turn), host clock over the run / 200; median of 5 runs",
                       "b1_call_and_synchronize_us": b1_sync_us,
                       "b1_kept": b1_kept,
                       "single_stream_event_pair_per_call_ms": {"median": p_med, "mean": p_mean, "min": p_min},
                       "scan_stage_ms": d_mean,
                       "scan_stage_frac_of_hbm_peak": decode_bytes / (d_mean * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "candidates_per_image": float(counts[1].mean()),
                       "kept_per_image": float(counts[0].mean()), "mode": "agnostic, thr 0.05, IoU 0.5",
                       "per_class_images_per_s": BATCH / (per_class_ms * 1e-3),
                       "pipeline_class_ms_per_batch": pipe_ms,
                       "pipeline_class_trials_ms_per_batch": pipe_trials,
                       "pipeline_class_images_per_s": BATCH / (pipe_ms * 1e-3)},
    }
    line["cpu_baseline"] = None
    if not args.no_cpu_baseline and world == 1:      # the host-side baseline is timed at N=1 only
        threads = min(16, os.cpu_count() or 1)
        sample = BATCH                # about 4 s of CPU work on the whole batch
        v, secs = cpu_baseline(cls_np, reg_np, anc_np, ann_np, sample, threads, 4.0)
        # parity of the measured path with the port on the benchmark's own batch (1e-5 relative;
        # regression gradients with the float32 floor of the quadratic zone, 2e-6 of the largest)
        ref = cpu_baseline.last
        chk = F.focal_loss_fwd_bwd(cls.detach(), reg.detach(), anchors, ann)
        got_c, got_r = chk["d_classifications"].cpu().numpy(), chk["d_regressions"].cpu().numpy()
        line["parity_ok"] = bool(
            np.allclose(chk["losses"].cpu().numpy(), [ref["cls_loss"], ref["reg_loss"]], rtol=1e-5)
            and np.allclose(got_c, ref["d_cls"], rtol=1e-5, atol=0)
            and np.allclose(got_r, ref["d_reg"], rtol=1e-5, atol=2e-6 * float(np.abs(ref["d_reg"]).max())))
        line["parity_checked_against"] = "torch-CPU port, whole batch: losses and both gradients"
        line["cpu_baseline"] = {"value": v, "unit": "anchors/s", "cores": threads, "kind": "port",
                                "sample": f"the whole batch ({sample} images, N={n}, C={CLASSES}), "
                                          f"forward+backward, best of {ref['runs']} runs of {secs:.2f} s"}
        pv, psecs = cpu_baseline_post(icls_np, ireg_np, anc_np, SIZE, 4, threads, 1.0)
        line["decode_nms"]["cpu_port_images_per_s"] = pv
        line["decode_nms"]["cpu_port_sample"] = f"4 images, {threads} threads, best of {cpu_baseline_post.runs} runs of {psecs:.2f} s"
    print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
