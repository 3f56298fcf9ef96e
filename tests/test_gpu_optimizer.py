"""FlatAdam (mixgan-tts_amd/optimizer.py; mg_grad_norm + mg_adam_flat) against torch.optim.Adam +
nn.utils.clip_grad_norm_, the pair train.py:81-83 runs per optimizer (utils/model.py:32-40)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd as m
    assert torch.cuda.is_available()
    m.lib()
    return m


def make_params(seed, dev):
    g = torch.Generator().manual_seed(seed)
    shapes = [(256, 80, 1), (256,), (512, 256, 3), (7,), (3, 5), (1,), (128, 33)]
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]


@pytest.mark.parametrize("clip,wd", [(1.0, 0.0), (1e9, 0.0), (0.05, 0.01)])
def test_flat_adam_matches_torch_adam(mg, clip, wd):
    ref = make_params(3, "cpu")
    ours = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref]
    bucket = mg.GradBucket(ours, order=[ours[2], ours[0]])          # a layout that is not the list order
    opt = mg.optimizer.FlatAdam(bucket, lr=2e-3, betas=(0.8, 0.99), weight_decay=wd)
    ropt = torch.optim.Adam(ref, lr=2e-3, betas=(0.8, 0.99), weight_decay=wd)
    sched, rsched = (torch.optim.lr_scheduler.ExponentialLR(o, gamma=0.5) for o in (opt, ropt))
    g = torch.Generator().manual_seed(11)
    for step in range(6):
        grads = [torch.randn(p.shape, generator=g) * (10.0 if step == 2 else 0.3) for p in ref]
        for p, q, gr in zip(ref, ours, grads):
            p.grad = gr.clone()
            q.grad = gr.cuda()
        bucket.gather()
        versions = [q._version for q in ours]
        norm = torch.nn.utils.clip_grad_norm_(ref, clip)
        ropt.step()
        out = opt.step(max_grad_norm=clip)
        assert abs(out[0].item() - norm.item()) <= 1e-5 * norm.item()
        assert all(q._version > v for q, v in zip(ours, versions)), "in-place update must bump the version counters"
        if step == 3:
            sched.step(), rsched.step()
        for p, q in zip(ref, ours):
            assert q.data_ptr() >= opt.flat_p.data_ptr()
            # atol = 0.5 % of one update (lr = 2e-3).  Where clipped gradient and weight decay nearly cancel, the
            # update lr * m / (sqrt(v) + eps) has slope lr / eps = 2e5 in the gradient: a last-bit difference in
            # g * clip + wd * p (the clip factor comes from a differently ordered norm) moves it by ~1e-6
            torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-6, atol=1e-5)
    # the state dict is torch.optim.Adam's: a stock Adam continues from it, and FlatAdam from a stock one
    sd = copy.deepcopy(opt.state_dict())
    assert sd["state"][0]["step"] == 6 and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    stock = torch.optim.Adam([torch.nn.Parameter(q.detach().clone()) for q in bucket.params], lr=1.0)
    stock.load_state_dict(sd)
    assert stock.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
    ours2 = [torch.nn.Parameter(q.detach().clone()) for q in bucket.params]
    b2 = mg.GradBucket(ours2)
    opt2 = mg.optimizer.FlatAdam(b2, lr=1.0)
    opt2.load_state_dict(stock.state_dict())
    grads = [torch.randn(p.shape, generator=g).cuda() for p in ours2]
    for q, s_, gr in zip(ours2, stock.param_groups[0]["params"], grads):
        q.grad, s_.grad = gr.clone(), gr.clone()
    b2.gather()
    opt2.step()
    stock.step()
    for q, s_ in zip(ours2, stock.param_groups[0]["params"]):
        torch.testing.assert_close(q.detach(), s_.detach(), rtol=2e-6, atol=1e-5)
    assert float(opt2.state[ours2[0]]["step"]) == 7


def test_flat_adam_adopts_a_restored_stock_adam_and_keeps_its_index_order(mg):
    """The resume path: get_model() hands back torch.optim.Adam objects with a checkpoint's state (utils/model.py:41-46);
    the trainer's FlatAdam takes that over by parameter identity and, built with the same param_order, writes state
    dicts that index parameters as the stock optimizer does."""
    params = make_params(9, "cuda")
    stock = torch.optim.Adam(params, lr=3e-4, betas=(0.5, 0.9))
    g = torch.Generator().manual_seed(2)
    for _ in range(2):
        for p in params:
            p.grad = torch.randn(p.shape, generator=g).cuda()
        stock.step()
    sched = torch.optim.lr_scheduler.ExponentialLR(stock, gamma=0.9)
    sched.step()
    twin = [torch.nn.Parameter(p.detach().clone()) for p in params]              # what the stock optimizer would do next
    stock2 = torch.optim.Adam(twin, lr=1.0)
    stock2.load_state_dict(stock.state_dict())
    bucket = mg.GradBucket(params, order=[params[4], params[2]])
    flat = mg.FlatAdam(bucket, lr=1.0, param_order=params).adopt(stock)
    assert flat.param_groups[0]["lr"] == stock.param_groups[0]["lr"] and flat.param_groups[0]["betas"] == (0.5, 0.9)
    sd, ref = flat.state_dict(), stock.state_dict()
    assert sd["param_groups"][0]["params"] == ref["param_groups"][0]["params"]
    for k in ref["state"]:
        assert float(sd["state"][k]["step"]) == float(ref["state"][k]["step"]) == 2
        assert torch.equal(sd["state"][k]["exp_avg"], ref["state"][k]["exp_avg"])           # same index -> same parameter
        assert torch.equal(sd["state"][k]["exp_avg_sq"], ref["state"][k]["exp_avg_sq"])
    grads = [torch.randn(p.shape, generator=g).cuda() for p in params]
    for p, q, gr in zip(params, twin, grads):
        p.grad, q.grad = gr.clone(), gr.clone()
    bucket.gather()
    flat.step()
    stock2.step()
    for p, q in zip(params, twin):
        torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-6, atol=1e-6)
    with pytest.raises(ValueError):
        mg.FlatAdam(bucket, lr=1.0, param_order=params[:-1])


def test_grad_norm_tail_and_arg_checks(mg):
    import ctypes
    L = mg._lib.lib()
    for n in (1, 3, 4, 5, 1023, 4 * 256 * 1024 + 2):
        g = torch.randn(n, device="cuda")
        out = mg.ops.grad_norm(g, 0.5)
        ref = torch.linalg.vector_norm(g.double()).item()
        assert abs(out[0].item() - ref) <= 1e-5 * ref
        assert abs(out[1].item() - min(1.0, 0.5 / (ref + 1e-6))) <= 1e-5
    assert L.mg_adam_flat(None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None, None) == -1
    p = torch.zeros(8, device="cuda")
    vp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    assert L.mg_adam_flat(vp(p), vp(p), vp(p), vp(p), 8, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None, None) == -2   # step >= 1
    assert L.mg_adam_flat(vp(p[1:]), vp(p), vp(p), vp(p), 4, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None, None) == -1  # alignment
