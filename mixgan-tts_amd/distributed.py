"""Batch-sharded data parallelism for the training step: one process per GPU, RCCL over xGMI
(`torch.distributed` backend "nccl" is RCCL on ROCm), gloo on CPU for tests.

The reference has no multi-device path (its nn.DataParallel is pinned to one GPU, SURVEY.md
section 5).  The path shards over the batch with no data-path collective; the only exchange is
one SUM all-reduce per optimizer of the flattened gradients (G: 92 MB fp32, D: 7.5 MB), divided by
the world size, BEFORE `clip_grad_norm_` (train.py:81 clips the gradient the optimizer sees).
A single flat bucket per optimizer keeps the collective large (xGMI rings are per-link bound;
fewer, larger messages), and lets the D all-reduce overlap the start of the G phase on a side
stream when `async_op=True`.
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class GradBucket:
    """Flat fp32 buffer holding the gradients of a parameter list, so the all-reduce is one collective with no
    packing step.  `.grad` of every parameter is (re-)pointed at its slice: producers that know the layout write
    there directly (Denoiser.bind_grad_buffer: the backward's layer-major gradient arrays ARE slices of `flat`, and
    autograd adopts them as `.grad` without a copy), every other gradient is copied in by `gather()` in one
    multi-tensor launch.  `order` (optional) lists parameters in the order they should be laid out; the rest follow."""

    def __init__(self, params, order=None):
        wanted = [p for p in params if p.requires_grad]
        ids = {id(p) for p in wanted}
        first = [p for p in (order or []) if id(p) in ids]
        seen = {id(p) for p in first}
        self.params = first + [p for p in wanted if id(p) not in seen]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.views = []
        self.offsets = {}
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            self.offsets[id(p)] = off
            off += p.numel()

    def gather(self):
        """Make `flat` hold the current gradients and every `.grad` alias its slice.  A parameter without a gradient
        on this rank (conditionally unused) contributes zeros AND gets `.grad` set, so that after the all-reduce every
        rank's optimizer steps it with the same averaged gradient (ranks must not diverge)."""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
                p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)
        return self.flat

    # ------------------------------------------------------------------ exchange
    # One SUM all-reduce per optimizer step, divided by the world size -- optionally in pieces: a producer that knows
    # a slice of `flat` is final before the rest (the denoiser's backward finishes its k=3 weight gradients, a third
    # of the G bucket, 0.6 ms before its last kernel) hands it to all_reduce_chunk_async, which puts the collective
    # behind an event on a side stream so that it runs next to the remaining backward kernels; all_reduce_mean then
    # reduces whatever is left and waits for the pieces.
    _pending = ()
    _side = None
    stub = False          # measurement only (bench.py --workload train): skip the collectives, keep everything else
    always_exchange = False   # run the collectives on a world of one too (proves the RCCL path; tests, bench)

    def exchanging(self):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.always_exchange)

    def all_reduce_chunk_async(self, lo, hi, ready_event=None, group=None):
        """Start the SUM all-reduce of flat[lo:hi] now.  ready_event (torch.cuda.Event recorded on the producing
        stream) marks the point after which the slice is final; the collective waits for it instead of for everything
        queued on the current stream.  No-op (returns False) on one process or on CPU tensors."""
        if not self.exchanging() or not self.flat.is_cuda or hi <= lo:
            return False
        for (_, plo, phi) in self._pending:
            if lo < phi and plo < hi:
                raise RuntimeError("GradBucket: overlapping chunks in flight")
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.flat.device)
        cur = torch.cuda.current_stream(self.flat.device)
        if ready_event is not None:
            self._side.wait_event(ready_event)
        else:
            self._side.wait_stream(cur)
        work = None
        if not self.stub:
            with torch.cuda.stream(self._side):      # c10d orders the collective behind the CURRENT stream: the side one
                work = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True)
        self._pending = tuple(self._pending) + ((work, lo, hi),)
        return True

    def all_reduce_mean(self, group=None, async_op=False):
        """SUM all-reduce of the flat gradient (the parts not already in flight), then divide by the world size."""
        self.gather()
        if not self.exchanging():
            return None
        world = dist.get_world_size(group)
        pending, self._pending = self._pending, ()
        if pending and self.flat.is_cuda:
            # what the chunks did not cover, in as few collectives as possible
            cuts, at = [], 0
            for (_, lo, hi) in sorted(pending, key=lambda c: c[1]):
                if lo > at:
                    cuts.append((at, lo))
                at = max(at, hi)
            if at < self.flat.numel():
                cuts.append((at, self.flat.numel()))
            works = [w for (w, _, _) in pending if w is not None]
            if not self.stub:
                works += [dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True)
                          for (lo, hi) in cuts]
            for w in works:
                w.wait()                             # stream-level wait on the GPU, no host block
            self.flat.div_(world)
            return None
        if self.stub:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            return _Pending(work, self.flat, world)
        self.flat.div_(world)
        return None


class _Pending:
    def __init__(self, work, flat, world):
        self.work, self.flat, self.world = work, flat, world

    def wait(self):
        self.work.wait()
        self.flat.div_(self.world)


def shard_batch(n_items, rank=None, world=None):
    """Contiguous shard [lo, hi) of a batch of n_items for this rank (inference: no collective)."""
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    if world is None:
        world = dist.get_world_size() if is_distributed() else 1
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)
