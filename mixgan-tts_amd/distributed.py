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
    """Flat fp32 view of the gradients of a parameter list; `.grad` of every parameter is made a
    view into one contiguous buffer so the all-reduce needs no packing copy."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def gather(self):
        """Copy (or alias) the current .grad tensors into the flat buffer."""
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
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
