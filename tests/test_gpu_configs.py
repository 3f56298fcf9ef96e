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


# ------------------------------------------------------------------------------------------------------------------
# Whole workloads at the per-GPU shard size of BASELINE configs[3] and configs[4]: the
# components were covered at full size, the workloads only at small size.
# ------------------------------------------------------------------------------------------------------------------
def _cfg3_trainer(mg, manifest, stats, reduced_d=None):
    args, pre, mc, tr = hot_path_configs("naive", 4, multi_speaker=True, stats_dir=stats)
    G = mg.GaussianDiffusion(args, pre, mc, tr)
    D = mg.JCUDiscriminator(pre, mc, tr)
    load_seeded(G, manifest, "diffusion_naive_ms1", 32)
    load_seeded(D, manifest, "jcu_ms1", 42)
    with torch.no_grad():   # the fixture recipe leaves output_projection at its zero init: make the path live
        G.denoise_fn.output_projection.conv.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(3))
    trainer = mg.HotPathTrainer(G.cuda(), D.cuda(), tr, mc)
    seen = []

    def hook(name, bucket):
        seen.append((name, bucket.flat.detach().clone()))
        if name == "D" and reduced_d is not None:
            # what the all-reduce would have left here: the mean over the shards (so that, as on 8 GPUs, every shard's G
            # phase sees the same updated discriminator)
            bucket.flat.copy_(reduced_d)
    trainer.grad_hook = hook
    return trainer, seen


def test_cfg3_training_step_at_the_full_per_gpu_shard(mg, manifest, tmp_path):
    """BASELINE configs[3]: AISHELL3 multi-speaker GAN step, batch 64 over 8 GPUs = B=8, L=1000 per GPU.  Two
    HotPathTrainer.step()s at exactly that shard: finite; reproducible given pinned t / noise; and after step 1 the
    reduced-gradient buckets equal the mean of the buckets of two B=4 half shards (what the 8-GPU all-reduce relies on,
    SURVEY.md section 8e; train.py:133-184)."""
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"], n_speakers=5)
    B, L, steps = 8, 1000, 2
    gen = torch.Generator().manual_seed(83)
    mel = (torch.rand(B, L, 80, generator=gen) * 13.5 - 11.5).cuda()
    cond = torch.randn(B, L, 256, generator=gen).cuda()
    spk = torch.randn(B, 256, generator=gen).cuda()
    ts = [torch.randint(0, 4, (B,), generator=gen) for _ in range(2 * steps)]
    noises = [torch.randn(B, 1, 80, L, generator=gen) for _ in range(6 * steps)]

    def run(lo, hi, n_steps, reduced_d=None):
        trainer, seen = _cfg3_trainer(mg, manifest, stats, reduced_d)
        trainer.G.t_fn = Tape([t[lo:hi].numpy() for t in ts])
        trainer.G.noise_fn = Tape([n[lo:hi].numpy() for n in noises])
        pad = torch.zeros(hi - lo, L, dtype=torch.bool, device="cuda")
        outs = [trainer.step(mel[lo:hi].contiguous(), cond[lo:hi].contiguous(), spk[lo:hi].contiguous(), pad)
                for _ in range(n_steps)]
        vals = [trainer.log_scalars(o) for o in outs]           # also raises on a hand-off timeout
        return vals, seen, trainer

    vals, seen, trainer = run(0, B, steps)
    assert [n for n, _ in seen] == ["D", "G"] * steps
    for v in vals:
        assert all(np.isfinite(x) for x in v.values()), v
    for _, flat in seen:
        assert torch.isfinite(flat).all()
    assert all(torch.isfinite(p).all() for p in list(trainer.G.parameters()) + list(trainer.D.parameters()))
    # reproducible (a few reductions combine partial sums with atomics: to rounding, not bitwise)
    vals2, seen2, _ = run(0, B, steps)
    for a, b in zip(vals, vals2):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-5 * max(1.0, abs(a[k])), k
    for i, ((_, f1), (_, f2)) in enumerate(zip(seen, seen2)):
        tol = 1e-5 if i < 2 else 2e-3       # step 2 starts from weights that may differ in the last bits
        assert (f1 - f2).abs().max().item() <= tol * f1.abs().max().item(), i
    # the two half shards
    _, h0, _ = run(0, B // 2, 1, seen[0][1])
    _, h1, _ = run(B // 2, B, 1, seen[0][1])
    for i, name in enumerate(("D", "G")):
        full, avg = seen[i][1], 0.5 * (h0[i][1] + h1[i][1])
        err = (full - avg).abs().max().item() / full.abs().max().item()
        assert err <= 3e-4, "%s bucket: full shard vs mean of half shards %.2e" % (name, err)


class _FixedEncoder(torch.nn.Module):
    """Stands in for the linguistic encoder (upstream of the path): hands back a fixed conditioner and frame counts."""

    def __init__(self, cond, mel_lens):
        super().__init__()
        self.cond, self.mel_lens = cond, mel_lens

    def forward(self, *args, **kwargs):
        B, L, _ = self.cond.shape
        valid = torch.arange(L, device=self.cond.device)[None, :] < self.mel_lens[:, None]     # True = valid here
        z = torch.zeros(B, 3, device=self.cond.device)
        return self.cond, None, None, z, z, self.mel_lens, valid, None, None


def test_cfg4_shallow_long_form_inference_end_to_end(mg, manifest, tmp_path):
    """BASELINE configs[4]: 'shallow' inference on L=4000 frames with the f16 MFMA attention path and the T=1000
    reverse chain, end to end through MixGANTTS.forward (model/mixgantts.py:136-160, model/diffusion.py:194-200):
    Decoder (L above max_seq_len) -> mel_linear -> PostNet -> diffuse_fn(coarse, T-1) -> 1000 p_sample steps -> denorm
    * mask.  B=2: graph replay == eager loop bit for bit with sigma = 0 and the same diffuse_fn draw; each utterance
    alone gives the same mel as inside the batch; finite with the in-kernel noise on; the coarse mel against the oracle
    (fp32 attention 5e-5, f16 attention 2e-3 of the mel's range)."""
    from oracle import weights as WR
    Tn, B, L = 1000, 2, 4000
    stats = write_stats(tmp_path, np.linspace(-11.5, -9.0, 80), np.linspace(1.0, 2.0, 80))
    gen = torch.Generator().manual_seed(404)
    cond = (torch.randn(B, L, 256, generator=gen) * 0.5).cuda()
    mel_lens = torch.tensor([L, L - 517]).cuda()
    man = manifest["mixgantts_shallow_ms0"]
    w = WR.draw(man["seeded"], 61)

    def build(rows):
        m = mg.MixGANTTS(*hot_path_configs("shallow", Tn, stats_dir=stats),
                         linguistic_encoder=_FixedEncoder(cond[rows].contiguous(), mel_lens[rows].contiguous()))
        sd = m.state_dict()
        for k, a in w.items():
            if k in sd and tuple(sd[k].shape) == a.shape:       # the T=4 recipe's schedule buffers have another length
                sd[k] = torch.from_numpy(a)
        m.load_state_dict(sd)
        with torch.no_grad():
            m.diffusion.denoise_fn.output_projection.conv.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(3))
        return m.cuda().eval()

    def infer(m, seed=5):
        n = m.linguistic_encoder.cond.shape[0]
        z = torch.zeros(n, dtype=torch.long, device="cuda")
        torch.manual_seed(seed)                                   # diffuse_fn's torch.randn draw
        with torch.no_grad():
            out, _, coarse = m(z, torch.ones(n, 5, dtype=torch.long, device="cuda"), torch.full((n,), 5, device="cuda"), 5,
                               torch.ones(n, 3, dtype=torch.long, device="cuda"), torch.full((n,), 3, device="cuda"), 3)
        return out[0], coarse

    m = build(slice(0, B))
    assert m.diffusion.num_timesteps == Tn
    # ---- coarse mel vs the oracle, utterance 0 (fp32 attention, then the f16 path of this config)
    Wt = {k: T(a) for k, a in w.items()}
    pad = torch.arange(L)[None, :] >= mel_lens.cpu()[:, None]
    with torch.no_grad():
        ref = R.coarse_mel(Wt, cond[:1].cpu(), pad[:1], max_seq_len=1000)
        c32 = m.coarse_mel(cond, pad.cuda())
        m.decoder.set_attention_precision("f16")
        c16 = m.coarse_mel(cond, pad.cuda())
    assert_close(c32[:1].cpu(), ref, 5e-5, "coarse mel, L=4000 > max_seq_len, fp32 attention")
    assert_close(c16[:1].cpu(), ref, 2e-3, "coarse mel, f16 MFMA attention")
    # ---- the whole inference, sigma = 0: captured graph == eager loop, batch independence
    sched = m.diffusion.posterior_log_variance_clipped.clone()
    m.diffusion.posterior_log_variance_clipped.fill_(-1.0e4)
    eager, coarse_e = infer(m)
    m.diffusion.use_graph = True
    graphed, coarse_g = infer(m)
    assert tuple(eager.shape) == (B, L, 80) and torch.isfinite(eager).all()
    assert torch.equal(coarse_e, coarse_g) and torch.equal(eager, graphed)
    assert (eager[1, L - 517:] == 0).all() and eager[1, :L - 517].abs().max() > 0
    m.diffusion.use_graph = False
    one = build(slice(1, 2))
    one.decoder.set_attention_precision("f16")
    one.diffusion.posterior_log_variance_clipped.fill_(-1.0e4)
    # the same diffuse_fn noise for utterance 1: draw the batch's tensor and hand its row over
    torch.manual_seed(5)
    nz = torch.randn((B, 1, 80, L), device="cuda")
    one.diffusion.noise_fn = lambda shape: nz[1:2]
    alone, _ = infer(one)
    assert_close(alone[0].cpu(), eager[1].cpu(), 1e-5, "utterance 1 alone vs inside the batch")
    # ---- in-kernel noise on: finite, differs from the sigma = 0 chain, fresh per call
    m.diffusion.posterior_log_variance_clipped.copy_(sched)
    m.diffusion.use_graph = True
    m.diffusion._graph = None
    n1, _ = infer(m)
    n2, _ = infer(m)
    assert torch.isfinite(n1).all() and not torch.equal(n1, eager) and not torch.equal(n1, n2)
