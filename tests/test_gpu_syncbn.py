"""Cross-rank BatchNorm statistics of the aux-mode PostNet (the only batch-coupled op
on the path): two processes on the one test GPU, each with half the batch, must reproduce single-process
BatchNorm over the whole batch -- forward, input gradient and (after summing over ranks) parameter gradients.
gloo carries the two tiny all-reduces (RCCL needs one GPU per rank; the driver's multi-GPU runs use nccl)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(4, 80, 33, generator=gen) * 1.5 + 0.3
    go = torch.randn(4, 80, 33, generator=gen)
    gamma, beta = torch.randn(80, generator=gen), torch.randn(80, generator=gen)
    keep = torch.rand(4, 80, 33, generator=gen) >= 0.5
    return x, go, gamma, beta, keep


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mixgan_tts_amd as mg
        x, go, gamma, beta, keep = _data()
        lo, hi = rank * 2, rank * 2 + 2
        xg = x[lo:hi].cuda().requires_grad_()
        gg, bg = gamma.cuda().requires_grad_(), beta.cuda().requires_grad_()
        out, mean, var = mg.autograd.batchnorm_act(xg, gg, bg, keep[lo:hi].to(torch.uint8).cuda(), 2.0, "tanh", 1e-5,
                                                   dist.group.WORLD)
        (out * go[lo:hi].cuda()).sum().backward()
        q.put((rank, out.detach().cpu(), xg.grad.cpu(), gg.grad.cpu(), bg.grad.cpu(), mean.cpu(), var.cpu()))
    finally:
        dist.destroy_process_group()


def test_two_rank_batchnorm_equals_full_batch():
    x, go, gamma, beta, keep = _data()
    xr, gr, br = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ref = torch.tanh(F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)) * keep * 2.0
    (ref * go).sum().backward()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    out = torch.cat([got[0][1], got[1][1]])
    dx = torch.cat([got[0][2], got[1][2]])
    tol = lambda a, b, t: (a - b).abs().max().item() <= t * (b.abs().max().item() + 1e-30)  # noqa: E731
    assert tol(out, ref.detach(), 1e-5)
    assert tol(dx, xr.grad, 5e-5)
    assert tol(got[0][3] + got[1][3], gr.grad, 5e-5)       # parameter grads: per-rank partial sums
    assert tol(got[0][4] + got[1][4], br.grad, 5e-5)
    assert tol(got[0][5], x.mean((0, 2)), 1e-5) and tol(got[1][6], x.var((0, 2), unbiased=False), 1e-5)
