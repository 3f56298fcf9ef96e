"""The x_t-independent conditioner projections of a sampling loop, hoisted out of its steps (mg_denoiser_cond_project +
mg_denoiser_psample's cproj): conditioner_projection(cond) of model/blocks.py:1150,1160 for all residual layers at once,
and p_sample reading it instead of projecting inside the kernel -- the same x_{t-1} bit for bit, on every tile width."""
import pytest
import torch

from helpers import golden, T, seeded, assert_close, hot_path_configs, write_stats, load_seeded, Tape
from oracle import refmath as R, schedule as S

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd as m
    assert torch.cuda.is_available()
    return m


def _diffusion(mg, manifest, tmp_path, ms=False):
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, multi_speaker=ms, stats_dir=stats))
    name = "diffusion_naive_ms%d" % int(ms)
    load_seeded(gd, manifest, name, 31)
    W, _ = seeded(manifest, name, 31)
    return gd.cuda().eval(), W


def _pin_width(monkeypatch, nt):
    monkeypatch.setenv("MG_PERSIST_NT", str(16 if nt in (116, 216) else 32 if nt == 232 else nt))
    if nt == 232:      # 32-frame tiles, the two-workgroups-per-CU build also where one tile per CU would get the other one
        monkeypatch.setenv("MG_PERSIST_SOLO", "0")
    else:
        monkeypatch.delenv("MG_PERSIST_SOLO", raising=False)
    if nt == 116:
        monkeypatch.setenv("MG_PERSIST_TEAM", "0")
    elif nt == 216:
        monkeypatch.setenv("MG_PERSIST_TEAM", "2")
    else:
        monkeypatch.delenv("MG_PERSIST_TEAM", raising=False)


@pytest.mark.parametrize("B,L", [(2, 200), (1, 77), (3, 1000), (1, 1)])
def test_cond_projection_matches_every_layers_conv(mg, manifest, tmp_path, B, L):
    gd, _ = _diffusion(mg, manifest, tmp_path)
    den = gd.denoise_fn
    cond = torch.randn(B, 256, L, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    got = den.cond_projection(cond)
    C = den._dims.channels
    assert got.shape == (B, den._dims.n_layers * C, L)
    for l, layer in enumerate(den.residual_layers):
        conv = layer.conditioner_projection.conv
        ref = torch.nn.functional.conv1d(cond.double(), conv.weight.double(), conv.bias.double())
        assert_close(got[:, l * C:(l + 1) * C].cpu(), ref.float().cpu(), TOL, "layer %d" % l)


@pytest.mark.parametrize("nt", [16, 216, 116, 32, 232, 64, 328, 864])
@pytest.mark.parametrize("ms", [False, True])
def test_p_sample_reading_the_projection_is_bitwise_the_same(mg, manifest, tmp_path, monkeypatch, ms, nt):
    _pin_width(monkeypatch, nt)
    gd, W = _diffusion(mg, manifest, tmp_path, ms)
    den = gd.denoise_fn
    buf = {k: T(v) for k, v in S.diffusion_buffers(S.beta_schedule("vpsde", 4, 0.1, 40, 0.008)).items()}
    gen = torch.Generator().manual_seed(17)
    for B, L in [(1, 200), (2, 129), (3, 333), (1, 1000), (4, 64)]:
        x = torch.randn(B, 80, L, generator=gen)
        cond = torch.randn(B, 256, L, generator=gen)
        spk = torch.randn(B, 256, generator=gen) if ms else None
        nz = torch.randn(B, 80, L, generator=gen)
        t = torch.tensor([3, 0, 2, 1][:B])
        xd, cd, nd, td = x.cuda(), cond.cuda(), nz.cuda(), t.cuda()
        sd = None if spk is None else spk.cuda()
        a = gd._p_sample_bml(xd, td, cd, sd, nd)
        # a step that leaves its projections behind: the same x_{t-1} bit for bit, and so is a step that reads them
        left = torch.full((B, 20 * 256, L), float("nan"), device="cuda")
        c = gd._p_sample_bml(xd, td, cd, sd, nd, cproj_out=left)
        assert torch.equal(a, c), "B=%d L=%d: %g" % (B, L, (a - c).abs().max().item())
        b = gd._p_sample_bml(xd, td, cd, sd, nd, cproj=left)
        assert torch.equal(a, b), "B=%d L=%d: %g" % (B, L, (a - b).abs().max().item())
        # the stand-alone GEMM adds the bias behind the sum, the kernels start their accumulators at it: the same
        # projections to the last bit or two, and a step reading them stays within the parity tolerance
        cproj = den.cond_projection(cd)
        assert_close(left.cpu(), cproj.cpu(), 2e-6, "in-kernel projections vs the GEMM B=%d L=%d" % (B, L))
        assert_close(gd._p_sample_bml(xd, td, cd, sd, nd, cproj=cproj).cpu(), a.cpu(), TOL, "reading the GEMM's")
        # ... and the step-dependent vectors computed for three steps at once (this step in the middle)
        ts = torch.stack([(td + 1) % 4, td, (td + 2) % 4])
        vecs = den.step_vectors(ts, sd)
        e = gd._p_sample_bml(xd, td, cd, sd, nd, cproj=left, step_vectors=(vecs, 1, 3))
        assert torch.equal(a, e), "B=%d L=%d step vectors" % (B, L)
        if B > 1:     # another step's slice gives another result: the slice is really read
            other = gd._p_sample_bml(xd, td, cd, sd, nd, step_vectors=(vecs, 0, 3))
            assert not torch.equal(a, other)
        if L <= 333:
            ref = R.p_sample(W, buf, x[:, None], t, cond, spk, nz[:, None], clip=True)
            assert_close(b.cpu()[:, None], ref, TOL, "p_sample with projection B=%d L=%d" % (B, L))
    den.check(sync=True)


def test_sampling_loop_same_with_and_without_hoisting(mg, manifest, tmp_path, monkeypatch):
    gd, _ = _diffusion(mg, manifest, tmp_path)
    B, L = 2, 300
    gen = torch.Generator().manual_seed(23)
    gd.cond = torch.randn(B, 256, L, generator=gen).cuda()
    gd.spk_emb = None
    start = torch.randn(B, 1, 80, L, generator=gen)
    draws = [torch.randn(B, 1, 80, L, generator=gen).numpy() for _ in range(4)]
    outs = []
    for hoist in ("1", "0"):
        monkeypatch.setenv("MG_COND_PREPROJECT", hoist)
        gd.noise_fn = Tape(list(draws))
        trace = gd.sampling(noise=start.cuda())
        assert len(trace) == 5
        outs.append(trace)
    gd.noise_fn = None
    assert gd._cproj_buf is not None and gd._cproj_buf[1].shape == (B, 20 * 256, L)
    assert_close(gd._cproj_buf[1].cpu(), gd.denoise_fn.cond_projection(gd.cond).cpu(), 2e-6, "the loop's projections")
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("ms", [False, True])
def test_full_size_loop_hoisted_equals_step_by_step(mg, manifest, tmp_path, monkeypatch, ms):
    """BASELINE configs[1] size (B=16, L=1000, T=4; 256 workgroups, every CU busy, every tile exchanging halos with both
    neighbours while the first step streams 328 MB of projections out and the others stream them in): the whole trace of
    the loop that shares its projections and step vectors equals the step-by-step loop's bit for bit, twice in a row (the
    second loop overwrites the first one's buffer)."""
    gd, _ = _diffusion(mg, manifest, tmp_path, ms)
    B, L = 16, 1000
    gen = torch.Generator(device="cuda").manual_seed(41)
    gd.cond = torch.randn(B, 256, L, device="cuda", generator=gen)
    gd.spk_emb = torch.randn(B, 256, device="cuda", generator=gen) if ms else None
    start = torch.randn(B, 1, 80, L, device="cuda", generator=gen)
    draws = [torch.randn(B, 1, 80, L, device="cuda", generator=gen).cpu().numpy() for _ in range(4)]
    traces = {}
    for hoist in ("1", "1", "0"):
        monkeypatch.setenv("MG_COND_PREPROJECT", hoist)
        gd.noise_fn = Tape(list(draws))
        traces.setdefault(hoist, []).append(gd.sampling(noise=start.clone()))
    gd.noise_fn = None
    ref = traces["0"][0]
    assert len(ref) == 5 and all(torch.isfinite(a).all() for a in ref)
    for tr in traces["1"]:
        for k, (a, b) in enumerate(zip(tr, ref)):
            assert torch.equal(a, b), "step %d: %g" % (k, (a - b).abs().max().item())
    gd.denoise_fn.check(sync=True)


def test_long_loop_takes_its_step_vectors_in_chunks(mg, manifest, tmp_path, monkeypatch):
    """T = 100 steps at B = 64: 1024 // 64 = 16 steps per chunk; the trace equals the loop that computes every step's
    vectors by itself."""
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    gd = mg.GaussianDiffusion(*hot_path_configs("naive", 100, multi_speaker=True, stats_dir=stats))
    load_seeded(gd.denoise_fn, manifest, "denoiser_ms1", 22)
    gd = gd.cuda().eval()
    B, L = 64, 40
    gen = torch.Generator().manual_seed(37)
    gd.cond = torch.randn(B, 256, L, generator=gen).cuda()
    gd.spk_emb = torch.randn(B, 256, generator=gen).cuda()
    start = torch.randn(B, 1, 80, L, generator=gen).cuda()
    zeros = torch.zeros(B, 1, 80, L, device="cuda")
    gd.noise_fn = lambda shape: zeros            # the posterior means: deterministic
    calls = []
    orig = gd.denoise_fn.step_vectors
    monkeypatch.setattr(gd.denoise_fn, "step_vectors", lambda ts, spk, packed=None: calls.append(ts.shape[0]) or orig(ts, spk, packed))
    a = gd.sampling(noise=start.clone(), keep_trace=False)[-1]
    assert calls == [16] * 6 + [4]
    monkeypatch.setenv("MG_COND_PREPROJECT", "0")
    b = gd.sampling(noise=start.clone(), keep_trace=False)[-1]
    assert len(calls) == 7 and torch.isfinite(a).all() and torch.equal(a, b)


def test_launch_per_layer_path_fills_the_projections_too(mg, manifest, tmp_path, monkeypatch):
    """MG_DENOISER_PERSIST=0 (and every shape the single-launch kernels do not take): the step projects per layer as
    before and the buffer a later step may read is filled by the GEMM."""
    gd, _ = _diffusion(mg, manifest, tmp_path)
    den = gd.denoise_fn
    gen = torch.Generator().manual_seed(31)
    B, L = 2, 150
    x, cond, nz = (torch.randn(B, c, L, generator=gen).cuda() for c in (80, 256, 80))
    t = torch.tensor([2, 1]).cuda()
    a = gd._p_sample_bml(x, t, cond, None, nz)
    monkeypatch.setenv("MG_DENOISER_PERSIST", "0")
    left = torch.full((B, 5120, L), float("nan"), device="cuda")
    b = gd._p_sample_bml(x, t, cond, None, nz, cproj_out=left)
    assert torch.equal(left, den.cond_projection(cond))
    vecs = den.step_vectors(torch.stack([t, t]), None)
    c = gd._p_sample_bml(x, t, cond, None, nz, cproj=left, step_vectors=(vecs, 1, 2))   # ignored by the per-layer kernels
    assert torch.equal(b, c)
    assert_close(b.cpu(), a.cpu(), TOL, "per-layer path")


def test_graphed_sampling_loop_projects_inside_the_graph(mg, manifest, tmp_path):
    """The captured loop recomputes the projections from the static conditioner buffer on every replay: a second replay
    with another conditioner must follow it (zero posterior noise is not available in-graph, so compare t = 0 ...
    instead: replay twice with the same inputs and different ones, and check against the eager loop's mean path)."""
    gd, _ = _diffusion(mg, manifest, tmp_path)
    B, L = 2, 160
    gen = torch.Generator().manual_seed(29)
    start = torch.randn(B, 1, 80, L, generator=gen).cuda()
    c1 = torch.randn(B, 256, L, generator=gen).cuda()
    c2 = torch.randn(B, 256, L, generator=gen).cuda()
    gd.spk_emb = None
    # the last step (t = 0) adds no noise, and the steps before it only perturb x_1: with the clamp to [-1, 1] the final
    # mel is a function of (x_1, cond); a replay that kept the FIRST conditioner's projections would not follow c2
    gd.cond = c1
    a1 = gd.sampling(noise=start.clone(), keep_trace=False, use_graph=True)[-1]
    assert gd._graph["cproj"] is not None
    p1 = gd._graph["cproj"].clone()
    gd.cond = c2
    a2 = gd.sampling(noise=start.clone(), keep_trace=False, use_graph=True)[-1]
    p2 = gd._graph["cproj"]
    assert torch.isfinite(a1).all() and torch.isfinite(a2).all()
    assert_close(p1.cpu(), gd.denoise_fn.cond_projection(c1).cpu(), 2e-6, "first replay's projections")
    assert_close(p2.cpu(), gd.denoise_fn.cond_projection(c2).cpu(), 2e-6, "second replay's projections")
    assert p2.data_ptr() != gd._loop_cond_buffer(c2, gd.denoise_fn.packed_weights()).data_ptr()   # the graph owns its buffer
    assert not torch.equal(a1, a2)


def test_projection_needs_the_inference_packs(mg, manifest, tmp_path):
    gd, _ = _diffusion(mg, manifest, tmp_path)
    den = gd.denoise_fn
    packed = den.packed_weights(with_backward=True)
    assert not den.has_cond_projection(packed)
    with pytest.raises(mg._lib.MixganHipError):
        den.cond_projection(torch.randn(1, 256, 64, device="cuda"), packed=packed)
    with pytest.raises(mg._lib.MixganHipError):          # the library refuses cproj without MG_FWD_P16 packs as well
        x = torch.randn(1, 80, 64, device="cuda")
        gd._p_sample_bml(x, torch.zeros(1, dtype=torch.long, device="cuda"), torch.randn(1, 256, 64, device="cuda"), None,
                         torch.zeros_like(x), packed=packed, cproj=torch.zeros(1, 5120, 64, device="cuda"))
