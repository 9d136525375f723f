"""Patch-batch data parallelism for the IQT trainer: one process per GPU, RCCL (backend "nccl" on ROCm) over
xGMI, gradients all-reduced in contiguous buckets of ONE flat fp32 arena, launched from autograd hooks in
reverse-layer order so the collectives overlap the rest of backward.

Replaces what the reference gets implicitly from ``accelerate.Accelerator`` -> torch DDP -> NCCL
(trainer.py:296-301, 476-497, 1118-1123): ``split_batches=True`` sharding,
``find_unused_parameters=True`` semantics (``mid_block`` / ``norm_cond`` never receive gradients),
``no_sync`` on non-boundary micro-steps, rank-0-only EMA / checkpoints.

xGMI is point-to-point (7 links per GPU), so few, large buckets are used: the 54 MB of config-2
gradients go out as ~25 MB buckets (first bucket 1 MB so the first collective starts early).
"""
import os
import sys
import time
from typing import List

import torch
import torch.distributed as dist


_TRACE = os.environ.get("DIQT_DDP_TRACE") == "1"      # diagnostic: every collective of the gradient reducer on stderr


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(device_type=None):
    """Initialises torch.distributed from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* when WORLD_SIZE > 1.
    Returns (world_size, rank, device)."""
    world, rank, local = env_world()
    use_cuda = torch.cuda.is_available() if device_type is None else device_type == "cuda"
    # rehearsal on a one-GPU box (bench.py --rehearse): DIQT_SHARE_DEVICE=1 maps every rank onto the cards that exist and
    # DIQT_DIST_BACKEND=gloo carries the collectives (RCCL refuses two ranks on one device).  Never set in production.
    share = os.environ.get("DIQT_SHARE_DEVICE") == "1"
    backend = os.environ.get("DIQT_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
    device = torch.device("cuda", local % torch.cuda.device_count() if share else local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if use_cuda and backend == "nccl":
            kwargs["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return world, rank, device


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def barrier():
    if world_size() > 1:
        dist.barrier()


class FlatArena:
    """All parameters of a module as views into one flat fp32 buffer (+ a flat gradient buffer whose slices are
    installed as ``p.grad``), so the optimiser is one kernel launch and a gradient bucket is a contiguous range."""

    def __init__(self, params: List[torch.nn.Parameter], with_grad=True):
        params = [p for p in params]
        assert len(params) > 0
        dev = params[0].device
        self.params = params
        self.offsets, off = [], 0
        for p in params:
            assert p.device == dev and p.dtype == torch.float32
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.numel = off
        self._collected = set()
        self.touched = set()                 # indices of parameters that have ever handed over a gradient (Adam's lazy state)
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev) if with_grad else None
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                self.flat[o:o + p.numel()].view_as(p).copy_(p.data)
                p.data = self.flat[o:o + p.numel()].view_as(p)
                if with_grad and p.requires_grad:
                    p.grad = self.grad[o:o + p.numel()].view_as(p)

    # ---- gradient accumulation without one add kernel per parameter -------------------------------------------------
    # autograd's AccumulateGrad adds into a defined ``p.grad`` with one small kernel per parameter (~290 per micro-step
    # for the C2 U-Net).  begin_backward() detaches the arena views so autograd just hands over each fresh gradient;
    # collect() adds them into the arena with ONE multi-tensor launch and re-installs the views.
    def begin_backward(self):
        self._collected = set()
        for p in self.params:
            if p.requires_grad:
                p.grad = None

    def collect(self, indices=None):
        idx = range(len(self.params)) if indices is None else indices
        srcs, offs, done = [], [], []
        for i in idx:
            if i in self._collected:
                continue
            p = self.params[i]
            if not p.requires_grad:
                continue
            g, o = p.grad, self.offsets[i]
            view = self.grad[o:o + p.numel()].view_as(p)
            if g is not None and g.data_ptr() != view.data_ptr():
                srcs.append(g.detach().contiguous().view(-1))
                offs.append(o)
                self.touched.add(i)
            p.grad = view
            done.append(i)
        self._collected.update(done)
        if not srcs:
            return
        if self.grad.is_cuda:
            from . import ops
            ops.multi_accumulate(self.grad, srcs, offs)
        else:                                   # host arenas (gloo tests): plain torch
            for t, o in zip(srcs, offs):
                self.grad[o:o + t.numel()].add_(t)

    def intact(self):
        """False if something (e.g. ``module.to()``) re-allocated a parameter and broke the views."""
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def reinstall_grads(self):
        for p, o in zip(self.params, self.offsets):
            if p.requires_grad and (p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o):
                p.grad = self.grad[o:o + p.numel()].view_as(p)


class BucketedGradReducer:
    """All-reduce(mean) of an arena's gradient buffer in contiguous buckets, overlapped with backward.

    Buckets are cut in REVERSE parameter order (the order backward produces gradients).  A bucket is launched
    (async) from the post-accumulate-grad hook of the last of its *used* parameters; the set of used parameters
    is learned in the first synchronised backward (every rank runs the same graph, so the sets agree) — the
    equivalent of DDP's ``find_unused_parameters=True``.  ``sync=False`` (gradient-accumulation micro-steps)
    skips communication like DDP's ``no_sync``.
    """

    def __init__(self, arena: FlatArena, bucket_cap_mb=25.0, first_bucket_mb=1.0, process_group=None, force=False):
        self.arena, self.pg = arena, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force: run the collectives on a one-rank group too (bring-up of the RCCL path on a one-GPU box, tests/test_gpu_rccl.py);
        # a single-rank mean is the identity, so the gradients must come out unchanged
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.sync = True
        self.used = None                    # learned set of parameter indices that receive gradients
        self._seen, self._pending, self._handles, self._order = set(), {}, [], []
        n = len(arena.params)
        ends = [arena.offsets[i + 1] if i + 1 < n else arena.numel for i in range(n)]
        self.buckets, cap, cur, cur_hi = [], int(first_bucket_mb * (1 << 18)), [], arena.numel
        for i in reversed(range(n)):
            cur.append(i)
            if cur_hi - arena.offsets[i] >= cap or i == 0:
                self.buckets.append((arena.offsets[i], cur_hi, tuple(cur)))
                cur, cur_hi, cap = [], arena.offsets[i], int(bucket_cap_mb * (1 << 18))
        self.bucket_of = {i: b for b, (_, _, idx) in enumerate(self.buckets) for i in idx}
        self._launched = set()
        self.stragglers_seen = 0
        for i, p in enumerate(arena.params):
            if p.requires_grad:
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(_param):
            if not self.active or not self.sync:
                return
            self._seen.add(i)
            if self.used is None:
                return
            b = self.bucket_of[i]
            left = self._pending.get(b)
            if left is None:
                left = set(j for j in self.buckets[b][2] if j in self.used)
                self._pending[b] = left
            left.discard(i)
            if not left and b not in self._launched:
                self._launch(b)
        return hook

    def _launch(self, b, only=None):
        lo, hi, idx = self.buckets[b]
        # this micro-step's gradients of the bucket join the accumulated ones.  Parameters outside the learned ``used`` set keep
        # ``p.grad = None``: should one of them receive a gradient after all (a straggler), autograd hands over a fresh tensor
        # instead of adding in place into a range RCCL is reducing, and finalize_backward() reduces the bucket once more.
        if only is None:
            only = idx if self.used is None else [j for j in idx if j in self.used]
        self.arena.collect(only)
        buf = self.arena.grad[lo:hi]
        native_avg = dist.get_backend(self.pg) == "nccl"        # RCCL averages in the collective; gloo sums, divided after the wait
        if _TRACE:
            print(f"[ddp rank {dist.get_rank()}] t={time.time() % 10000:.2f} launch bucket {b} [{lo}:{hi}] params {len(only)} handles {len(self._handles)}", file=sys.stderr, flush=True)
        h = dist.all_reduce(buf, op=dist.ReduceOp.AVG if native_avg else dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._launched.add(b)
        self._order.append(b)
        self._handles.append((h, buf, native_avg))

    def prepare_backward(self, sync=True):
        self.sync = sync
        self._seen, self._pending, self._handles, self._launched = set(), {}, [], set()
        self._order = []                    # bucket indices in the order this rank enqueued their collectives

    def finalize_backward(self):
        """Call after ``loss.backward()`` of a synchronised micro-step: launches whatever is left and waits."""
        if not self.active or not self.sync:
            return
        for b in range(len(self.buckets)):
            if b not in self._launched:
                self._launch(b)
        self._wait()
        if _TRACE:
            # every rank must enqueue the SAME collectives in the SAME order: compare the launch orders of this step
            orders = [None] * self.world
            dist.all_gather_object(orders, list(self._order), group=self.pg)
            print(f"[ddp rank {dist.get_rank()}] step done, launch order {self._order}", file=sys.stderr, flush=True)
            assert all(o == orders[0] for o in orders), f"ranks enqueued gradient buckets in different orders: {orders}"
        if self.used is None:
            self.used = set(self._seen)
        stragglers = sorted(self._seen - self.used)
        if stragglers:
            # a parameter outside the learned set received a gradient (the graph changed): its contribution arrived after the
            # bucket's collective.  The bucket holds avg(A) on every rank; adding the local late gradients s_r and averaging
            # again gives avg(A) + avg(s) -- exactly what one collective over everything would have produced.
            self.stragglers_seen += len(stragglers)
            for b in sorted({self.bucket_of[i] for i in stragglers}):
                self._launch(b, only=[i for i in stragglers if self.bucket_of[i] == b])
            self._wait()
            self.used |= set(stragglers)

    def _wait(self):
        for k, (h, buf, native_avg) in enumerate(self._handles):
            if _TRACE:
                print(f"[ddp rank {dist.get_rank()}] t={time.time() % 10000:.2f} wait {k} of {len(self._handles)}", file=sys.stderr, flush=True)
            h.wait()
            if _TRACE:
                print(f"[ddp rank {dist.get_rank()}] t={time.time() % 10000:.2f} done {k}", file=sys.stderr, flush=True)
            if not native_avg:
                buf.div_(self.world)
        self._handles = []


def broadcast_arena(arena: FlatArena, src=0, process_group=None):
    """Rank-0 parameters to everyone (what DDP does at construction, trainer.py:487)."""
    if world_size() > 1:
        dist.broadcast(arena.flat, src=src, group=process_group)


def broadcast_tensors(tensors, src=0, process_group=None):
    """In-place broadcast of each tensor from ``src`` (optimiser moments, step counters on resume)."""
    if world_size() > 1:
        for t in tensors:
            dist.broadcast(t, src=src, group=process_group)


def broadcast_ints(values, device, src=0, process_group=None):
    """Host integers from ``src`` to every rank; returns the list every rank agrees on."""
    if world_size() == 1:
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    dist.broadcast(t, src=src, group=process_group)
    return [int(v) for v in t.tolist()]


def shard_batch(t, world, rank_, full=None, initial=None):
    """``split_batches=True`` (trainer.py:297): every rank takes its contiguous 1/world slice of a global batch.

    The reference's loaders keep the last partial batch of an epoch (``drop_last=False``, train.py:56/67).  accelerate's
    ``BatchSamplerShard`` (``split_batches=True``, ``even_batches=True``) completes such a batch to the full batch size with
    samples of the epoch's FIRST batch (cycled if that is short too) before slicing it, so every rank still gets
    ``full / world`` samples and joins every collective.  ``full`` = the loader's batch size, ``initial`` = the first batch
    of the pass; without them the batch is completed to the next multiple of ``world`` from its own head."""
    if world == 1 or not torch.is_tensor(t):
        return t
    b = t.shape[0]
    target = full if (full is not None and full % world == 0 and b < full) else -(-b // world) * world
    if b != target:
        src = initial if (torch.is_tensor(initial) and initial.shape[0] > 0) else t
        pieces, have = [t], b
        while have < target:
            take = min(src.shape[0], target - have)
            pieces.append(src[:take].to(t.device))
            have += take
        t = torch.cat(pieces, dim=0)
    per = target // world
    return t[rank_ * per:(rank_ + 1) * per]
