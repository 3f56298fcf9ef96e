"""world_size-2 gloo test of the gradient exchange used by the training step (distributed.py):
the mean-all-reduced flat gradients of two batch shards equal the single-process gradients of the
whole batch, the flat bucket aliases .grad, and shard_batch partitions without overlap."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mixgan_tts_amd as mg
from mixgan_tts_amd.distributed import GradBucket, shard_batch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv1d(8, 16, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv1d(16, 4, 1))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(123)
        x = torch.randn(8, 8, 20)
        y = torch.randn(8, 4, 20)
        m = _model()
        lo, hi = shard_batch(8)
        loss = (m(x[lo:hi]) - y[lo:hi]).pow(2).mean()
        loss.backward()
        b = GradBucket(list(m.parameters()))
        b.all_reduce_mean()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(b.params, b.views))
        # async form gives the same numbers
        m2 = _model()
        (m2(x[lo:hi]) - y[lo:hi]).pow(2).mean().backward()
        b2 = GradBucket(list(m2.parameters()))
        b2.all_reduce_mean(async_op=True).wait()
        assert torch.equal(b.flat, b2.flat)
        if rank == 0:
            q.put(b.flat.clone())
    finally:
        dist.destroy_process_group()


def test_grad_bucket_allreduce_matches_full_batch():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    flat = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    torch.manual_seed(123)
    x = torch.randn(8, 8, 20)
    y = torch.randn(8, 4, 20)
    m = _model()
    (m(x) - y).pow(2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.allclose(flat, ref, rtol=1e-5, atol=1e-7)


def test_shard_batch_partitions():
    for n in (1, 7, 16, 64):
        for world in (1, 2, 3, 8):
            spans = [shard_batch(n, r, world) for r in range(world)]
            covered = [i for lo, hi in spans for i in range(lo, hi)]
            assert covered == list(range(n))


def test_single_process_is_noop():
    m = _model()
    m(torch.randn(2, 8, 5)).sum().backward()
    g0 = [p.grad.clone() for p in m.parameters()]
    b = GradBucket(list(m.parameters()))
    assert b.all_reduce_mean() is None
    for p, g in zip(m.parameters(), g0):
        assert torch.equal(p.grad, g)


def _unused_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(5)
        a, b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)      # `b` is only used on rank 1
        x = torch.randn(3, 4)
        y = a(x) + (b(x) if rank == 1 else 0)
        y.pow(2).mean().backward()
        params = list(a.parameters()) + list(b.parameters())
        assert (b.weight.grad is None) == (rank == 0)
        bucket = GradBucket(params, order=list(b.parameters()))  # a layout order other than the parameter order
        assert bucket.params[0] is b.weight
        bucket.all_reduce_mean()
        assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
        opt = torch.optim.Adam(params, lr=0.1)
        opt.step()
        q.put((rank, [p.detach().numpy().copy() for p in params]))     # numpy: pickled by value
    finally:
        dist.destroy_process_group()


def test_parameter_unused_on_one_rank_still_steps_identically():
    """A parameter without a gradient on one rank contributes zeros to the all-reduce and still gets `.grad` set, so
    every rank's optimizer applies the same averaged gradient and the replicas cannot drift apart."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_unused_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for w0, w1 in zip(got[0][1], got[1][1]):
        assert (w0 == w1).all()
    torch.manual_seed(5)
    b_init = [p.detach().clone() for p in (torch.nn.Linear(4, 4), torch.nn.Linear(4, 4))[1].parameters()]
    assert not (got[0][1][2] == b_init[0].numpy()).all()         # rank 0 moved `b` too
