"""GPU parity of the backward path (mg_conv1d_wgrad, mg_rowsum, mg_denoiser_bwd, the autograd
wrappers) against the reference fixtures (gradients recorded from the real reference) and against
torch.autograd on the CPU oracle.  Tolerance 5e-5 (max-abs err / max-abs ref): fp32 sums over up to
B*L frames in a different order (split-K + atomics for weight gradients)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import (golden, T, seeded, assert_close, assert_digest, hot_path_configs, write_stats, load_seeded, Tape)
from oracle import refmath as R

pytestmark = pytest.mark.gpu
GT = 5e-5


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd as m
    assert torch.cuda.is_available()
    m.lib()
    return m


def dev(a):
    return (T(a) if isinstance(a, np.ndarray) else a).cuda()


@pytest.mark.parametrize("Ci,Co,K,stride,L", [
    (256, 512, 3, 1, 130), (256, 512, 1, 1, 64), (80, 256, 1, 1, 37), (256, 80, 1, 1, 200),
    (64, 128, 5, 2, 37), (128, 512, 5, 2, 300), (128, 1, 3, 1, 10), (7, 5, 3, 1, 3), (160, 64, 3, 1, 1000),
])
def test_wgrad_matches_autograd(mg, Ci, Co, K, stride, L):
    g = torch.Generator().manual_seed(Ci + Co * 7 + K)
    B = 3
    pad = (K - 1) // 2
    x = torch.randn(B, Ci, L, generator=g)
    w = (torch.randn(Co, Ci, K, generator=g) / (Ci * K) ** 0.5).requires_grad_()
    y = F.conv1d(x, w, None, stride=stride, padding=pad)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    dw = mg.ops.conv1d_wgrad(gy.cuda(), x.cuda(), K, stride, pad)
    assert_close(dw.cpu(), w.grad, 2e-5, "wgrad")
    assert_close(mg.ops.rowsum(gy.cuda()).cpu(), gy.sum((0, 2)), 2e-5, "bias grad")
    assert_close(mg.ops.rowsum(gy.cuda(), per_batch=True).cpu(), gy.sum(2), 2e-5, "per-sample sums")


def test_wgrad_xvec(mg):
    g = torch.Generator().manual_seed(4)
    B, Ci, Co, L, K = 2, 512, 128, 70, 5
    x, v = torch.randn(B, Ci, L, generator=g), torch.randn(B, Ci, generator=g)
    w = (torch.randn(Co, Ci, K, generator=g) / 50).requires_grad_()
    gy = torch.randn(B, Co, L, generator=g)
    F.conv1d(x + v[:, :, None], w, None, padding=2).backward(gy)
    dw = mg.ops.conv1d_wgrad(gy.cuda(), x.cuda(), K, 1, 2, x_vec=v.cuda())
    assert_close(dw.cpu(), w.grad, 2e-5, "wgrad with input vector")


@pytest.mark.parametrize("ms", [0, 1])
def test_denoiser_backward_golden(mg, manifest, tmp_path, ms):
    name = "denoiser_ms%d" % ms
    g = golden(name)
    _, pre, mc, _ = hot_path_configs(multi_speaker=bool(ms), stats_dir=str(tmp_path))
    den = mg.Denoiser(pre, mc)
    load_seeded(den, manifest, name, 21 + ms)
    den = den.cuda()
    x, cond = dev(g["x"]).requires_grad_(), dev(g["cond"]).requires_grad_()
    spk = dev(g["spk"]).requires_grad_() if ms else None
    out = den(x, dev(g["t"]), cond, spk)
    assert_close(out.detach().cpu(), g["out"], 2e-5, "forward (save path)")
    (out * dev(g["go"])).sum().backward()
    torch.cuda.synchronize()
    assert_close(x.grad.cpu(), g["d_x"], GT, "d_x")
    assert_close(cond.grad.cpu(), g["d_cond"], GT, "d_cond")
    if ms:
        assert_close(spk.grad.cpu(), g["d_spk"], GT, "d_spk")
    for k, p in den.named_parameters():
        assert p.grad is not None, k
        assert_digest(p.grad, g, k, 1e-4)


@pytest.mark.parametrize("model,ms", [("naive", 0), ("naive", 1), ("shallow", 0)])
def test_gaussian_diffusion_training_gradients(mg, manifest, tmp_path, model, ms):
    name = "diffusion_%s_ms%d" % (model, ms)
    g = golden(name)
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    gd = mg.GaussianDiffusion(*hot_path_configs(model, 4, multi_speaker=bool(ms), stats_dir=stats))
    load_seeded(gd, manifest, name, 31 + ms)
    gd = gd.cuda().train()
    mel, pad = dev(g["mel"]), dev(g["pad"])
    cond = dev(g["cond"]).requires_grad_()
    spk = dev(g["spk"]) if ms else None
    coarse = dev(g["coarse"]) if model == "shallow" else None
    gd.t_fn = Tape([g["t"]])
    gd.noise_fn = Tape([g["n_xt"], g["n_prev"], g["n_post"]])
    x0p, x_t, x_prev, x_pp, t = gd(mel, cond, spk, pad, coarse)
    assert_close(x0p.detach().cpu(), g["x0_pred"], 2e-5, "x0_pred")
    assert_close(x_pp.detach().cpu(), g["x_prev_pred"], 2e-5, "x_prev_pred")
    ((x0p * dev(g["w1"])).sum() + (x_pp * dev(g["w2"])).sum()).backward()
    assert_close(cond.grad.cpu(), g["d_cond"], GT, "d_cond")
    params = dict(gd.named_parameters())
    for k in [k[len("dw_sum/"):] for k in g if k.startswith("dw_sum/")]:
        assert_digest(params[k].grad, g, k, 1e-4)


def test_denoiser_backward_vs_oracle_autograd_ragged(mg, manifest, tmp_path):
    """B, L not covered by fixtures (tile boundaries, L % 4 != 0): compare with torch.autograd on the oracle."""
    _, pre, mc, _ = hot_path_configs(stats_dir=str(tmp_path))
    den = mg.Denoiser(pre, mc)
    load_seeded(den, manifest, "denoiser_ms0", 78)
    den = den.cuda()
    W, _ = seeded(manifest, "denoiser_ms0", 78, requires_grad=True)
    gen = torch.Generator().manual_seed(2)
    B, L = 3, 131
    x = torch.randn(B, 1, 80, L, generator=gen)
    cond = torch.randn(B, 256, L, generator=gen)
    t = torch.tensor([0, 999, 17])
    go = torch.randn(B, 1, 80, L, generator=gen)
    xr, cr = x.clone().requires_grad_(), cond.clone().requires_grad_()
    (R.denoiser_forward(W, "", xr, t, cr, None) * go).sum().backward()
    xg, cg = x.cuda().requires_grad_(), cond.cuda().requires_grad_()
    (den(xg, t.cuda(), cg, None) * go.cuda()).sum().backward()
    assert_close(xg.grad.cpu(), xr.grad, GT, "d_x")
    assert_close(cg.grad.cpu(), cr.grad, GT, "d_cond")
    for k, p in den.named_parameters():
        assert_close(p.grad.cpu(), W[k].grad, 1e-4, k)


@pytest.mark.parametrize("Ci,Co,K,B,L", [(256, 512, 3, 16, 1000), (256, 256, 1, 64, 1004), (128, 128, 3, 1025, 68),
                                         (256, 128, 1, 1030, 36), (256, 640, 1, 13, 1000)])
def test_wgrad_streaming_kernel(mg, monkeypatch, Ci, Co, K, B, L):
    """Shapes the streaming kernel (wgrad_stream.h) takes: against autograd, and against the split kernel
    (MG_WGRAD_STREAM=0) that the small shapes above run on."""
    g = torch.Generator().manual_seed(Ci + Co * 7 + K + L)
    x = torch.randn(B, Ci, L, generator=g)
    w = (torch.randn(Co, Ci, K, generator=g) / (Ci * K) ** 0.5).requires_grad_()
    y = F.conv1d(x, w, None, padding=(K - 1) // 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    gyc, xc = gy.cuda(), x.cuda()
    dw = mg.ops.conv1d_wgrad(gyc, xc, K, 1, (K - 1) // 2)
    assert_close(dw.cpu(), w.grad, 2e-5, "streaming wgrad vs autograd")
    assert torch.equal(dw, mg.ops.conv1d_wgrad(gyc, xc, K, 1, (K - 1) // 2)), "summation order must not change run to run"
    monkeypatch.setenv("MG_WGRAD_STREAM", "0")
    dw0 = mg.ops.conv1d_wgrad(gyc, xc, K, 1, (K - 1) // 2)
    assert not torch.equal(dw, dw0), "both calls ran the same kernel"
    assert_close(dw.cpu(), dw0.cpu(), 2e-5, "streaming vs split kernel")


def test_grouped_wgrad_streaming(mg, monkeypatch):
    """The residual stack's grouped gradients on the streaming kernel: layer slots inside wider tensors, a shared dY
    (group stride 0), alpha and accumulation; a workgroup's run of units crosses tile and group boundaries."""
    import ctypes
    G, B, Co, Ci, K, L = 3, 38, 256, 128, 3, 520
    gen = torch.Generator().manual_seed(13)
    dy_all = torch.randn(B, G * Co, L, generator=gen).cuda()
    x_all = torch.randn(G, B, Ci, L, generator=gen).cuda()
    lib = mg._lib.lib()
    cp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    scratch = torch.empty(lib.mg_conv1d_wgrad_grouped_scratch_floats(Co, Ci, K, G), device="cuda")

    def run(dy_gs, alpha, acc, out):
        mg._lib.check(lib.mg_conv1d_wgrad_grouped(cp(dy_all), G * Co * L, dy_gs, cp(x_all), Ci * L, B * Ci * L, cp(out), 0,
                                                  cp(scratch), G, B, Co, Ci, L, L, K, 1, 1, alpha, acc, None))
        return out
    dw = run(Co * L, 1.0, 0, torch.empty(G, Co, Ci, K, device="cuda"))
    dw2 = run(0, 0.5, 1, torch.ones(G, Co, Ci, K, device="cuda"))
    monkeypatch.setenv("MG_WGRAD_STREAM", "0")
    ref = run(Co * L, 1.0, 0, torch.empty(G, Co, Ci, K, device="cuda"))
    ref2 = run(0, 0.5, 1, torch.ones(G, Co, Ci, K, device="cuda"))
    assert not torch.equal(dw, ref)
    assert_close(dw.cpu(), ref.cpu(), 2e-5, "grouped streaming wgrad")
    assert_close(dw2.cpu(), ref2.cpu(), 2e-5, "grouped streaming wgrad, shared dy + accumulate")
    xr = x_all[1].cpu()
    w = torch.zeros(Co, Ci, K, requires_grad=True)
    F.conv1d(xr, w, None, padding=1).backward(dy_all[:, Co:2 * Co].cpu())
    assert_close(dw[1].cpu(), w.grad, 2e-5, "grouped streaming wgrad vs autograd")


def test_grouped_wgrad_bias_rides_along(mg, monkeypatch):
    """mg_conv1d_wgrad_grouped_bias: db[g][co] = sum_{b,l} dy_g -- from the streaming kernel's staging registers for the
    shapes it takes, from row-sum launches otherwise; shared dY (group stride 0), a strided db, alpha, accumulation."""
    import ctypes
    lib = mg._lib.lib()
    cp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    gen = torch.Generator().manual_seed(14)
    for (G, B, Co, Ci, K, L, dy_shared) in [(3, 38, 256, 128, 3, 520, False), (2, 64, 256, 256, 1, 1004, True),
                                           (2, 3, 96, 40, 3, 204, False)]:
        dy_all = torch.randn(B, G * Co, L, generator=gen).cuda()
        x_all = torch.randn(G, B, Ci, L, generator=gen).cuda()
        scratch = torch.empty(lib.mg_conv1d_wgrad_grouped_scratch_floats(Co, Ci, K, G), device="cuda")
        dy_gs = 0 if dy_shared else Co * L
        dbs = 2 * Co                                   # bias rows of a group sit inside a wider row (the [2C] output-conv bias)
        for stream in ("1", "0"):
            monkeypatch.setenv("MG_WGRAD_STREAM", stream)
            dw = torch.empty(G, Co, Ci, K, device="cuda")
            db = torch.full((G, dbs), 7.0, device="cuda")
            mg._lib.check(lib.mg_conv1d_wgrad_grouped_bias(cp(dy_all), G * Co * L, dy_gs, cp(x_all), Ci * L, B * Ci * L, cp(dw), 0,
                                                           cp(db), dbs, cp(scratch), G, B, Co, Ci, L, L, K, 1, (K - 1) // 2, 0.5, 1, None))
            for g in range(G):
                rows = dy_all[:, :Co] if dy_shared else dy_all[:, g * Co:(g + 1) * Co]
                ref = 7.0 + 0.5 * rows.double().sum((0, 2))
                assert_close(db[g, :Co].cpu(), ref.float().cpu(), 2e-5, "bias gradient, group %d, stream=%s" % (g, stream))
                assert torch.all(db[g, Co:] == 7.0), "rows outside the group's bias slice were written"


def test_grouped_wgrad_matches_per_group_calls(mg):
    """mg_conv1d_wgrad_grouped: G gradients of one shape in one launch (shared or per-group operands, strided
    slots inside wider tensors) == G separate mg_conv1d_wgrad calls."""
    import ctypes
    G, B, Co, Ci, K, L = 5, 3, 96, 40, 3, 204
    gen = torch.Generator().manual_seed(12)
    dy_all = torch.randn(B, G * Co, L, generator=gen).cuda()          # slots in the row dimension: [B][G*Co][L]
    x_all = torch.randn(G, B, Ci, L, generator=gen).cuda()            # layer-major saves: [G][B][Ci][L]
    lib = mg._lib.lib()
    cp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    scratch = torch.empty(lib.mg_conv1d_wgrad_grouped_scratch_floats(Co, Ci, K, G), device="cuda")
    dw = torch.empty(G, Co, Ci, K, device="cuda")
    mg._lib.check(lib.mg_conv1d_wgrad_grouped(cp(dy_all), G * Co * L, Co * L, cp(x_all), Ci * L, B * Ci * L, cp(dw), 0,
                                              cp(scratch), G, B, Co, Ci, L, L, K, 1, 1, 1.0, 0, None))
    for g in range(G):
        ref = mg.ops.conv1d_wgrad(dy_all[:, g * Co:(g + 1) * Co].contiguous(), x_all[g], K, 1, 1)
        assert_close(dw[g].cpu(), ref.cpu(), 1e-6, "grouped wgrad group %d" % g)
    # shared dy (group stride 0), accumulate into an existing gradient
    dw2 = torch.ones(G, Co, Ci, K, device="cuda")
    mg._lib.check(lib.mg_conv1d_wgrad_grouped(cp(dy_all), G * Co * L, 0, cp(x_all), Ci * L, B * Ci * L, cp(dw2), 0,
                                              cp(scratch), G, B, Co, Ci, L, L, K, 1, 1, 0.5, 1, None))
    for g in range(G):
        ref = 1.0 + 0.5 * mg.ops.conv1d_wgrad(dy_all[:, :Co].contiguous(), x_all[g], K, 1, 1)
        assert_close(dw2[g].cpu(), ref.cpu(), 1e-6, "grouped wgrad shared dy, group %d" % g)
