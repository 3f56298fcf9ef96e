"""BASELINE.json configs[2] (shallow, T=100 p_sample loop, B=32, hipGraph-captured) and configs[4] (long-form
L=4000, shallow T=1000) on the GPU: the T=100 / T=1000 schedules through the real kernels, not only the T=4 fixtures.
  (i)   shallow T=100 inference (diffuse_fn(coarse, T-1) -> 100 p_sample steps -> denorm * mask) against the oracle
        chain with every noise injected (model/diffusion.py:194-200,155-165);
  (ii)  the same config at the BASELINE size B=32, L=1000 through the captured graph: graph == eager bit for bit with
        the posterior noise switched off, batch independence, finite output;
  (iii) T=1000: p_sample at t in {0, 1, 999} and diffuse_fn at t in {-1, 0, 999} against the oracle, with every
        schedule buffer the kernels have no business reading poisoned with NaN (model/diffusion.py:47-83);
  (iv)  Denoiser.forward at B=1, L=4000 against the oracle; the T=1000 graph-captured chain == eager.
Tolerances: max-abs error / max-abs reference, as everywhere in tests/ (helpers.rel_err)."""
import numpy as np
import pytest
import torch

from helpers import golden, T, seeded, assert_close, hot_path_configs, write_stats, load_seeded, Tape
from oracle import refmath as R, schedule as S

pytestmark = pytest.mark.gpu

USED_BUFFERS = ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
                "posterior_log_variance_clipped")


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd as m
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return m


def _gd(mg, tmp_path, manifest, model, T_, seed_name="diffusion_shallow_ms0"):
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    gd = mg.GaussianDiffusion(*hot_path_configs(model, T_, stats_dir=stats))
    load_seeded(gd, manifest, seed_name, 31)
    W, _ = seeded(manifest, seed_name, 31)
    buf = {k: T(v) for k, v in S.diffusion_buffers(S.beta_schedule("vpsde", T_, 0.1, 40, 0.008)).items()}
    buf["spec_min"], buf["spec_max"] = T(e["spec_min"])[None, None, :], T(e["spec_max"])[None, None, :]
    return gd.cuda().eval(), W, buf


def test_cfg2_shallow_T100_inference_vs_oracle(mg, manifest, tmp_path):
    Tn, B, L = 100, 2, 64
    gd, W, buf = _gd(mg, tmp_path, manifest, "shallow", Tn)
    assert gd.num_timesteps == Tn
    gen = torch.Generator().manual_seed(100)
    cond = torch.randn(B, L, 256, generator=gen)
    coarse = torch.rand(B, L, 80, generator=gen) * 12.5 - 11.0
    pad = torch.arange(L)[None, :] >= torch.tensor([64, 51])[:, None]
    noises = [torch.randn(B, 1, 80, L, generator=gen) for _ in range(Tn + 1)]
    ref, *_ = R.diffusion_forward(W, buf, "shallow", Tn, None, cond, None, pad, coarse, R.NoiseTape(noises))
    gd.noise_fn = Tape([n.numpy() for n in noises])
    with torch.no_grad():
        out, x_t, x_prev, x_pp, t = gd(None, cond.cuda(), None, pad.cuda(), coarse.cuda())
    assert gd.noise_fn.i == Tn + 1 and x_t is None and x_prev is None and x_pp is None
    assert torch.equal(t.cpu(), torch.full((B,), Tn - 1))
    assert_close(out.cpu(), ref, 5e-5, "shallow T=100 inference mel")
    assert (out[1, 51:] == 0).all()


def test_cfg2_graph_captured_T100_full_size(mg, manifest, tmp_path):
    Tn, B, L = 100, 32, 1000
    gd, _, _ = _gd(mg, tmp_path, manifest, "shallow", Tn)
    gen = torch.Generator(device="cuda").manual_seed(7)
    cond = torch.randn(B, L, 256, device="cuda", generator=gen)
    x_T = torch.randn(B, 1, 80, L, device="cuda", generator=gen)
    pad = torch.zeros(B, L, dtype=torch.bool, device="cuda")
    gd.cond, gd.spk_emb = mg.ops.transpose_bml(cond, False), None
    gd.posterior_log_variance_clipped.fill_(-1.0e4)          # sigma = 0: eager and graph are both deterministic
    eager = gd.sampling(noise=x_T, keep_trace=False)[0]
    graphed = gd.sampling(noise=x_T, keep_trace=False, use_graph=True)[0]
    assert torch.isfinite(eager).all()
    assert torch.equal(eager, graphed)
    # batch independence through the eager loop at B=1 (sample 9 alone == sample 9 inside the 32)
    gd.cond = gd.cond[9:10].contiguous()
    one = gd.sampling(noise=x_T[9:10], keep_trace=False)[0]
    assert torch.equal(one[0], eager[9])
    # the forward() entry replays the graph when use_graph is set (on-device noise: finite, right shape)
    gd.posterior_log_variance_clipped.copy_(torch.from_numpy(
        S.diffusion_buffers(S.beta_schedule("vpsde", Tn, 0.1, 40, 0.008))["posterior_log_variance_clipped"]).float())
    gd._graph = None
    gd.use_graph = True
    coarse = torch.rand(B, L, 80, device="cuda", generator=gen) * 12.5 - 11.0
    with torch.no_grad():
        y, *_ = gd(None, cond, None, pad, coarse)
    assert tuple(y.shape) == (B, L, 80) and torch.isfinite(y).all() and getattr(gd, "_graph", None) is not None


def test_cfg4_T1000_steps_read_only_finite_buffers(mg, manifest, tmp_path):
    Tn, L = 1000, 50
    gd, W, buf = _gd(mg, tmp_path, manifest, "shallow", Tn)
    for k in USED_BUFFERS:
        assert torch.isfinite(getattr(gd, k)).all(), k
    for k in mg.schedule.BUFFER_NAMES:
        if k not in USED_BUFFERS:
            getattr(gd, k).fill_(float("nan"))         # a kernel reading any of these would poison its output
    gen = torch.Generator().manual_seed(1000)
    t = torch.tensor([0, 1, 999])
    x_t = torch.randn(3, 1, 80, L, generator=gen)
    cond = torch.randn(3, 256, L, generator=gen)
    nz = torch.randn(3, 1, 80, L, generator=gen)
    ref = R.p_sample(W, buf, x_t, t, cond, None, nz)
    gd.noise_fn = Tape([nz.numpy()])
    out = gd.p_sample(x_t.cuda(), t.cuda(), cond.cuda(), None)
    assert torch.isfinite(out).all()
    assert_close(out.cpu(), ref, 2e-5, "p_sample at t = 0, 1, 999 of T = 1000")
    mel = torch.rand(3, L, 80, generator=gen) * 13.5 - 11.5
    td = torch.tensor([-1, 0, 999])
    refd = R.diffuse_fn(buf, mel, td.clone(), nz)
    gd.noise_fn = Tape([nz.numpy()])
    outd = gd.diffuse_fn(mel.cuda(), td.clone().cuda())
    assert torch.isfinite(outd).all()
    assert_close(outd.cpu(), refd, 2e-6, "diffuse_fn at t = -1, 0, 999 of T = 1000")


def test_cfg4_long_form_denoiser_and_T1000_chain(mg, manifest, tmp_path):
    Tn = 1000
    gd, W, buf = _gd(mg, tmp_path, manifest, "shallow", Tn)
    gen = torch.Generator().manual_seed(4000)
    B, L = 1, 4000
    x = torch.randn(B, 1, 80, L, generator=gen)
    cond = torch.randn(B, 256, L, generator=gen)
    t = torch.tensor([517])
    with torch.no_grad():
        ref = R.denoiser_forward(W, "denoise_fn.", x, t, cond, None)
        out = gd.denoise_fn(x.cuda(), t.cuda(), cond.cuda(), None)
    assert_close(out.cpu(), ref, 2e-5, "Denoiser.forward B=1 L=4000")
    # the whole 1000-step chain, captured, equals the eager chain (sigma = 0), and stays finite with noise on
    B, L = 2, 96
    gd.cond = torch.randn(B, 256, L, generator=gen).cuda()
    gd.spk_emb = None
    x_T = torch.randn(B, 1, 80, L, generator=gen).cuda()
    saved = gd.posterior_log_variance_clipped.clone()
    gd.posterior_log_variance_clipped.fill_(-1.0e4)
    eager = gd.sampling(noise=x_T, keep_trace=False)[0]
    graphed = gd.sampling(noise=x_T, keep_trace=False, use_graph=True)[0]
    assert torch.equal(eager, graphed) and torch.isfinite(eager).all()
    gd.posterior_log_variance_clipped.copy_(saved)
    gd._graph = None
    noisy = gd.sampling(noise=x_T, keep_trace=False, use_graph=True)[0]
    assert torch.isfinite(noisy).all() and not torch.equal(noisy, eager)
