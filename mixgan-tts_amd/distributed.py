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

    def all_reduce_mean(self, group=None, async_op=False):
        """SUM all-reduce of the flat gradient, then divide by the world size."""
        self.gather()
        if not is_distributed():
            return None
        world = dist.get_world_size(group)
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
