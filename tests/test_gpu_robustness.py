"""Ownership of the Denoiser's workspaces and of the gradient bucket (host-side robustness):
  * a pending backward keeps the workspace that holds its saved activations (autograd.DenoiserFn), so no-grad forwards
    at many other shapes -- which cycle the per-shape cache -- cannot take it away, and two grad-enabled forwards may
    both be pending;
  * a captured sampling graph owns the workspace whose address it baked in;
  * the denoiser's backward writes its weight gradients straight into the all-reduce bucket (no gather copy)."""
import pytest
import torch

from helpers import golden, T, assert_close, assert_digest, hot_path_configs, write_stats, load_seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd as m
    assert torch.cuda.is_available()
    return m


def _den(mg, manifest, tmp_path):
    g = golden("denoiser_ms0")
    _, pre, mc, _ = hot_path_configs(stats_dir=str(tmp_path))
    den = mg.Denoiser(pre, mc)
    load_seeded(den, manifest, "denoiser_ms0", 21)
    return den.cuda(), g


def test_backward_survives_nine_other_shapes_between_forward_and_backward(mg, manifest, tmp_path):
    den, g = _den(mg, manifest, tmp_path)
    dev = lambda k: T(g[k]).cuda()  # noqa: E731
    x, cond = dev("x").requires_grad_(), dev("cond").requires_grad_()
    out = den(x, dev("t"), cond, None)
    gen = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():
        for i in range(9):       # nine new (B, L): more than the cache holds
            B, L = 1 + i % 3, 40 + 8 * i
            den(torch.randn(B, 1, 80, L, device="cuda", generator=gen), torch.zeros(B, dtype=torch.long, device="cuda"),
                torch.randn(B, 256, L, device="cuda", generator=gen), None)
    # and a second grad-enabled forward at the SAME shape, still pending when the first one's backward runs
    x2, cond2 = (dev("x") * 0.5).requires_grad_(), dev("cond").requires_grad_()
    out2 = den(x2, dev("t"), cond2, None)
    (out * dev("go")).sum().backward()
    assert_close(x.grad.cpu(), g["d_x"], 5e-5, "d_x")
    assert_close(cond.grad.cpu(), g["d_cond"], 5e-5, "d_cond")
    for k, p in den.named_parameters():
        assert_digest(p.grad, g, k, 1e-4)
    den.zero_grad(set_to_none=True)
    (out2 * dev("go")).sum().backward()                      # its own activations are intact too
    assert torch.isfinite(x2.grad).all() and not torch.equal(x2.grad, x.grad)
    ref = den(x2.detach().clone().requires_grad_(), dev("t"), cond2.detach().clone().requires_grad_(), None)
    assert torch.equal(ref.detach(), out2.detach())


def test_captured_graph_owns_its_workspace(mg, manifest, tmp_path):
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=stats))
    load_seeded(gd, manifest, "diffusion_naive_ms0", 31)
    gd = gd.cuda().eval()
    B, L = 2, 72
    gen = torch.Generator(device="cuda").manual_seed(3)
    gd.cond, gd.spk_emb = torch.randn(B, 256, L, device="cuda", generator=gen), None
    x_T = torch.randn(B, 1, 80, L, device="cuda", generator=gen)
    gd.posterior_log_variance_clipped.fill_(-1.0e4)          # sigma = 0: deterministic
    first = gd.sampling(noise=x_T, keep_trace=False, use_graph=True)[0]
    eager = gd.sampling(noise=x_T, keep_trace=False)[0]
    assert torch.equal(first, eager)
    den = gd.denoise_fn
    with torch.no_grad():
        for i in range(10):      # cycle the denoiser's workspace cache and churn the allocator
            Bo, Lo = 1 + i % 2, 48 + 16 * i
            den(torch.randn(Bo, 1, 80, Lo, device="cuda", generator=gen), torch.zeros(Bo, dtype=torch.long, device="cuda"),
                torch.randn(Bo, 256, Lo, device="cuda", generator=gen), None)
            junk = torch.randn(1 << 20, device="cuda", generator=gen)   # lands in whatever was freed
            del junk
    again = gd.sampling(noise=x_T, keep_trace=False, use_graph=True)[0]
    assert torch.equal(again, eager)


def test_denoiser_backward_writes_into_the_gradient_bucket(mg, manifest, tmp_path):
    den, g = _den(mg, manifest, tmp_path)
    bucket = mg.distributed.GradBucket(list(den.parameters()), order=den.grad_order())
    den.bind_grad_buffer(bucket.flat, bucket.offsets)
    dev = lambda k: T(g[k]).cuda()  # noqa: E731
    x, cond = dev("x").requires_grad_(), dev("cond").requires_grad_()
    (den(x, dev("t"), cond, None) * dev("go")).sum().backward()
    lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + 4 * bucket.flat.numel()
    assert all(lo <= p.grad.data_ptr() < hi for p in den.parameters()), "a gradient was not produced in place"
    for p, v in zip(bucket.params, bucket.views):
        assert p.grad.data_ptr() == v.data_ptr()
    for k, p in den.named_parameters():
        assert_digest(p.grad, g, k, 1e-4)
    before = bucket.flat.clone()
    bucket.gather()                                            # nothing to copy, nothing changes
    assert torch.equal(before, bucket.flat)
    # a second backward before zero_grad must ACCUMULATE (fresh tensors, added by autograd into the bucket views)
    (den(x, dev("t"), cond, None) * dev("go")).sum().backward()
    assert_close(bucket.flat.cpu(), (2 * before).cpu(), 1e-5, "accumulated gradients")
