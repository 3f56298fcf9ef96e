"""The single-launch Denoiser.forward / p_sample (csrc/denoiser_persist.h: 32-frame tiles resident for all layers, two
workgroups per CU, halo columns of h handed between neighbouring workgroups inside the launch) against the oracle, against
the launch-per-layer kernels, and against itself under conditions that expose a stale or torn hand-off: full chip
occupancy (B=16, L=1000 = 512 workgroups), repeated launches, a second stream hammering HBM, single samples vs the
batch (bitwise)."""
import numpy as np
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


def _den(mg, manifest, tmp_path, ms=False):
    name = "denoiser_ms%d" % int(ms)
    _, pre, mc, _ = hot_path_configs(multi_speaker=ms, stats_dir=str(tmp_path))
    den = mg.Denoiser(pre, mc)
    load_seeded(den, manifest, name, 21 + int(ms))
    W, _ = seeded(manifest, name, 21 + int(ms))
    return den.cuda(), W


def _pin_width(monkeypatch, nt):
    """MG_PERSIST_NT pins a tile width; 16: teams of workgroups per tile where they fit (denoiser_team16.h: 4 members,
    else 2), 216: teams of 2 only, 116: 16-frame tiles with one workgroup per tile (denoiser_persist16.h) everywhere;
    328: 32-frame tiles, 8 waves; 64: 64-frame tiles as four waves (one per SIMD), 864: as eight waves of 32 channels."""
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


@pytest.mark.parametrize("nt", [16, 216, 116, 32, 232, 64, 328, 864])
@pytest.mark.parametrize("ms", [False, True])
def test_single_launch_forward_vs_oracle_and_per_layer_path(mg, manifest, tmp_path, monkeypatch, ms, nt):
    _pin_width(monkeypatch, nt)     # every tile width, whatever the heuristic would pick
    den, W = _den(mg, manifest, tmp_path, ms)
    gen = torch.Generator().manual_seed(11)
    # one tile, a partial tile, tile boundaries, L % 4 != 0 (scalar staging) and == 0 (float4 staging)
    for B, L in [(1, 1), (2, 31), (1, 32), (3, 33), (2, 129), (1, 300), (5, 257), (2, 64), (3, 65), (1, 1000)]:
        x = torch.randn(B, 1, 80, L, generator=gen)
        cond = torch.randn(B, 256, L, generator=gen)
        spk = torch.randn(B, 256, generator=gen) if ms else None
        t = torch.randint(0, 1000, (B,), generator=gen)
        with torch.no_grad():
            ref = R.denoiser_forward(W, "", x, t, cond, spk)
            args = (x.cuda(), t.cuda(), cond.cuda(), None if spk is None else spk.cuda())
            monkeypatch.delenv("MG_DENOISER_PERSIST", raising=False)
            out = den(*args)
            monkeypatch.setenv("MG_DENOISER_PERSIST", "0")
            per_layer = den(*args)
            monkeypatch.delenv("MG_DENOISER_PERSIST", raising=False)
        st = den.persist_status(B, L)
        assert st["error"] == 0 and st["ticket"] == 0 and st["done"] == 0 and st["launches"] >= 1, st
        assert_close(out.cpu(), ref, TOL, "single launch B=%d L=%d" % (B, L))
        assert_close(out.cpu(), per_layer.cpu(), TOL, "single launch vs per-layer kernels B=%d L=%d" % (B, L))


def test_fused_p_sample_vs_oracle(mg, manifest, tmp_path):
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=stats))
    load_seeded(gd, manifest, "diffusion_naive_ms0", 31)
    W, _ = seeded(manifest, "diffusion_naive_ms0", 31)
    buf = {k: T(v) for k, v in S.diffusion_buffers(S.beta_schedule("vpsde", 4, 0.1, 40, 0.008)).items()}
    gd = gd.cuda().eval()
    gen = torch.Generator().manual_seed(5)
    for B, L in [(3, 50), (2, 96), (4, 257), (1, 16)]:       # small launches: the 16-frame width by default
        x_t = torch.randn(B, 1, 80, L, generator=gen)
        cond = torch.randn(B, 256, L, generator=gen)
        nz = torch.randn(B, 1, 80, L, generator=gen)
        t = torch.tensor([3, 0, 2, 1][:B])
        for clip in (True, False):
            ref = R.p_sample(W, buf, x_t, t, cond, None, nz, clip=clip)
            gd.noise_fn = Tape([nz.numpy()])
            out = gd.p_sample(x_t.cuda(), t.cuda(), cond.cuda(), None, clip_denoised=clip)
            assert_close(out.cpu(), ref, TOL, "fused p_sample B=%d L=%d clip=%s" % (B, L, clip))
    gd.noise_fn = None


def test_in_kernel_noise_is_standard_normal_and_fresh(mg, manifest, tmp_path):
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=stats))
    load_seeded(gd, manifest, "diffusion_naive_ms0", 31)
    gd = gd.cuda().eval()
    B, L = 4, 512
    gen = torch.Generator(device="cuda").manual_seed(2)
    x_t = torch.randn(B, 1, 80, L, device="cuda", generator=gen)
    cond = torch.randn(B, 256, L, device="cuda", generator=gen)
    t = torch.tensor([3, 2, 1, 0], device="cuda")
    zero = torch.zeros(B, 1, 80, L)
    gd.noise_fn = Tape([zero.numpy()])
    mean = gd.p_sample(x_t, t, cond, None)                  # sigma * 0: the posterior mean alone
    gd.noise_fn = None
    a = gd.p_sample(x_t, t, cond, None)
    b = gd.p_sample(x_t, t, cond, None)
    sig = torch.exp(0.5 * gd.posterior_log_variance_clipped[t]).view(B, 1, 1, 1)
    za, zb = ((a - mean) / sig)[:3], ((b - mean) / sig)[:3]  # rows with t > 0
    assert torch.equal(a[3], mean[3]) and torch.equal(b[3], mean[3])        # t == 0: no noise (model/diffusion.py:116)
    for z in (za, zb):
        n = z.numel()
        assert abs(z.mean().item()) < 5.0 / n ** 0.5 and abs(z.var().item() - 1.0) < 0.02
        assert abs((z ** 3).mean().item()) < 0.05 and abs((z ** 4).mean().item() - 3.0) < 0.1
        # (n = 122,880: one sigma of a sample correlation is 0.0029 -- 0.015 is five of them; which stream a run draws
        # depends on how many workspaces the process has made before)
        assert abs(torch.corrcoef(torch.stack([z.flatten()[:-1], z.flatten()[1:]]))[0, 1].item()) < 0.015
    assert abs(torch.corrcoef(torch.stack([za.flatten(), zb.flatten()]))[0, 1].item()) < 0.015   # a fresh stream per call


def _corr(a, b):
    n = min(a.numel(), b.numel())
    return abs(torch.corrcoef(torch.stack([a.flatten()[:n], b.flatten()[:n]]))[0, 1].item())


def test_in_kernel_noise_streams_never_repeat(mg, manifest, tmp_path):
    """The reference draws fresh torch.randn_like noise in every p_sample (model/diffusion.py:32-35,118).  The in-kernel
    generator's counter is (workspace number, launches on that workspace): the first launch on ANOTHER workspace -- a new
    shape, the same shape after the 8-entry cache evicted it, a second captured graph -- must not replay the first
    launch of the first one (round 2: every workspace restarted at offset 0 under one seed)."""
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=stats))
    load_seeded(gd, manifest, "diffusion_naive_ms0", 31)
    gd = gd.cuda().eval()
    den = gd.denoise_fn
    gen = torch.Generator(device="cuda").manual_seed(5)
    M = 80

    def problem(B, L):
        return (torch.randn(B, M, L, device="cuda", generator=gen), torch.full((B,), 3, device="cuda"),
                torch.randn(B, 256, L, device="cuda", generator=gen))

    def z_of(x, t, cond, **kw):
        """standardised noise of one p_sample: (x_{t-1} - posterior mean) / sigma"""
        B, _, L = x.shape
        mean = gd._p_sample_bml(x, t, cond, None, torch.zeros(B, M, L, device="cuda"))
        out = gd._p_sample_bml(x, t, cond, None, None, **kw)
        return (out - mean) / torch.exp(0.5 * gd.posterior_log_variance_clipped[t]).view(B, 1, 1)

    with torch.no_grad():
        pa, pb = problem(2, 256), problem(2, 320)
        za1 = z_of(*pa)                       # first launch with in-kernel noise on workspace (2, 256)
        zb1 = z_of(*pb)                       # ... on workspace (2, 320): same flat element indices for utterance 0
        n = za1.numel()
        assert abs(za1.var().item() - 1.0) < 0.03 and abs(zb1.var().item() - 1.0) < 0.03
        assert _corr(za1, zb1) < 5.0 / n ** 0.5, "two shapes drew the same noise"
        assert _corr(za1[0], zb1[0]) < 5.0 / (n / 2) ** 0.5
        for i in range(9):                    # push (2, 256) out of the 8-entry workspace cache
            z_of(*problem(1, 32 + 16 * i))
        assert not any(k[:3] == (2, 256, False) for k in den._ws)
        za2 = z_of(*pa)                       # first in-kernel-noise launch on the re-allocated workspace
        assert _corr(za1, za2) < 5.0 / n ** 0.5, "a re-allocated workspace replayed the old workspace's stream"

        # two captured graphs of the same step, each owning its workspace; replays of one graph; graph vs graph
        packed = den.packed_weights()
        x, t, cond = pa
        mean = gd._p_sample_bml(x, t, cond, None, torch.zeros_like(x))
        sig = torch.exp(0.5 * gd.posterior_log_variance_clipped[t]).view(2, 1, 1)
        outs = []
        for _ in range(2):
            ws, out = den.new_workspace(2, 256, False, x.device), torch.empty_like(x)
            gd._p_sample_bml(x, t, cond, None, None, out=out, packed=packed, ws=ws)    # warm-up outside the capture
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gd._p_sample_bml(x, t, cond, None, None, out=out, packed=packed, ws=ws)
            reps = []
            for _ in range(2):
                graph.replay()
                reps.append(((out - mean) / sig).clone())
            outs.append((reps, graph, ws))
        (g1a, g1b), (g2a, g2b) = outs[0][0], outs[1][0]
        for u, v, what in ((g1a, g1b, "two replays of one graph"), (g1a, g2a, "first replays of two graphs"),
                           (g1b, g2b, "second replays of two graphs"), (g1a, za1, "a graph and the eager call")):
            assert abs(u.var().item() - 1.0) < 0.03
            assert _corr(u, v) < 5.0 / n ** 0.5, what + " drew the same noise"
    den.check()


@pytest.mark.parametrize("nt", [16, 216, 116, 32, 64, 864])
def test_handoff_timeout_poisons_the_output_and_raises(mg, manifest, tmp_path, monkeypatch, nt):
    """A neighbour that never sends its edge column (test hook MG_PERSIST_FLAGS bit 1) with the wait bounded to a few
    polls: the kernel must drain (not hang), its output must be NaN (not a plausible mel), the failure must reach the
    host as MixganHipError, and the module must work again afterwards."""
    _pin_width(monkeypatch, nt)
    den, W = _den(mg, manifest, tmp_path)
    gen = torch.Generator(device="cuda").manual_seed(9)
    B, L = 2, 200
    x = torch.randn(B, 1, 80, L, device="cuda", generator=gen)
    cond = torch.randn(B, 256, L, device="cuda", generator=gen)
    t = torch.tensor([5, 900], device="cuda")
    with torch.no_grad():
        good = den(x, t, cond, None)
        den.check()
        ws = den._workspace(B, L, False, x.device)
        monkeypatch.setenv("MG_PERSIST_SPIN_LIMIT", "40")
        monkeypatch.setenv("MG_PERSIST_FLAGS", "3")
        bad = den(x, t, cond, None)
        monkeypatch.delenv("MG_PERSIST_SPIN_LIMIT")
        monkeypatch.delenv("MG_PERSIST_FLAGS")
        torch.cuda.synchronize()                  # returns: every workgroup exited
        assert torch.isnan(bad[:, 0, :, :16 if nt in (116, 216) else nt]).all(), "the tile that timed out wrote a result"
        assert not torch.isfinite(bad).all()
        st = den.persist_status(B, L, ws=ws)
        assert st["error"] != 0 and st["ticket"] == 0 and st["done"] == 0 and st["launches"] >= 2, st
        with pytest.raises(mg.MixganHipError, match="hand-off"):
            den.check()
        assert mg.lib().mg_persist_error(0) == 0   # reported once
        assert den._workspace(B, L, False, x.device) is not ws      # the poisoned workspace is gone
        again = den(x, t, cond, None)
        den.check()
        assert torch.equal(again, good)


def test_sampling_raises_on_a_handoff_timeout(mg, manifest, tmp_path, monkeypatch):
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    gd = mg.GaussianDiffusion(*hot_path_configs("naive", 4, stats_dir=stats))
    load_seeded(gd, manifest, "diffusion_naive_ms0", 31)
    gd = gd.cuda().eval()
    gen = torch.Generator(device="cuda").manual_seed(4)
    gd.cond, gd.spk_emb = torch.randn(2, 256, 150, device="cuda", generator=gen), None
    ok = gd.sampling(keep_trace=False)[0]
    assert torch.isfinite(ok).all()
    monkeypatch.setenv("MG_PERSIST_SPIN_LIMIT", "40")
    monkeypatch.setenv("MG_PERSIST_FLAGS", "3")
    with pytest.raises(mg.MixganHipError, match="hand-off"):
        gd.sampling(keep_trace=False)
    monkeypatch.delenv("MG_PERSIST_SPIN_LIMIT")
    monkeypatch.delenv("MG_PERSIST_FLAGS")
    assert torch.isfinite(gd.sampling(keep_trace=False)[0]).all()


@pytest.mark.parametrize("nt", [16, 32, 64, 328, 864])   # 328: 32-frame tiles, 8 waves (one workgroup per CU)
def test_full_occupancy_handoffs_are_never_stale(mg, manifest, tmp_path, monkeypatch, nt):
    """B=16, L=1000: 512 workgroups, two per CU, every tile waiting on both neighbours in every layer.  The output must be
    bit-identical run after run, with or without a second stream saturating HBM beside it, and identical to what each
    utterance gives alone (other placement, other neighbours in flight); and it must match the per-layer kernels."""
    # 512 workgroups of 32 frames (two per CU) / 256 of 64 frames.  16: one workgroup per tile also for the single
    # utterances below (the four-workgroup teams reduce in another order: their own test compares them with themselves)
    _pin_width(monkeypatch, 116 if nt == 16 else nt)
    den, W = _den(mg, manifest, tmp_path)
    gen = torch.Generator(device="cuda").manual_seed(16)
    B, L = 16, 1000
    x = torch.randn(B, 1, 80, L, device="cuda", generator=gen)
    cond = torch.randn(B, 256, L, device="cuda", generator=gen)
    t = torch.randint(0, 1000, (B,), device="cuda", generator=gen)
    with torch.no_grad():
        first = den(x, t, cond, None).clone()
        for _ in range(5):
            assert torch.equal(den(x, t, cond, None), first)
        side = torch.cuda.Stream()
        big = torch.empty(1 << 28, device="cuda")                      # 1 GiB
        with torch.cuda.stream(side):
            for _ in range(40):
                big.add_(1.0)
        for _ in range(10):
            assert torch.equal(den(x, t, cond, None), first)
        side.synchronize()
        for bi in (0, 7, 15):
            one = den(x[bi:bi + 1], t[bi:bi + 1], cond[bi:bi + 1], None)
            assert torch.equal(one[0], first[bi])
        monkeypatch.setenv("MG_DENOISER_PERSIST", "0")
        per_layer = den(x, t, cond, None)
    assert den.persist_status(B, L)["error"] == 0
    assert_close(first.cpu(), per_layer.cpu(), TOL, "single launch vs per-layer kernels at B=16 L=1000")
    with torch.no_grad():
        ref = R.denoiser_forward(W, "", x[:1].cpu(), t[:1].cpu(), cond[:1].cpu(), None)
    assert_close(first[:1].cpu(), ref, TOL, "vs oracle")


@pytest.mark.parametrize("nt", [16, 32, 64, 328, 864])   # 328: 32-frame tiles, 8 waves (one workgroup per CU)
def test_more_tiles_than_slots_and_long_utterances(mg, manifest, tmp_path, monkeypatch, nt):
    """B=40, L=1000 = 1280 (640) workgroups on 512 (256) slots (later tiles start as earlier utterances finish), and
    L=4000 (125- / 63-tile chains): finite, deterministic, equal to each sample alone."""
    _pin_width(monkeypatch, 116 if nt == 16 else nt)     # (one workgroup per tile also for the sample alone)
    den, _ = _den(mg, manifest, tmp_path)
    gen = torch.Generator(device="cuda").manual_seed(40)
    for B, L in [(40, 1000), (6, 4000 if nt > 16 else 2000)]:     # 16-frame chains: L <= 2048
        x = torch.randn(B, 1, 80, L, device="cuda", generator=gen)
        cond = torch.randn(B, 256, L, device="cuda", generator=gen)
        t = torch.randint(0, 1000, (B,), device="cuda", generator=gen)
        with torch.no_grad():
            a = den(x, t, cond, None).clone()
            b = den(x, t, cond, None)
            one = den(x[B - 1:], t[B - 1:], cond[B - 1:], None)
        assert den.persist_status(B, L)["error"] == 0
        assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(one[0], a[B - 1])


@pytest.mark.parametrize("ms", [False, True])
@pytest.mark.parametrize("Bh,L", [(2, 131), (8, 1000), (3, 64)])
def test_paired_forwards_match_separate_launches(mg, manifest, tmp_path, ms, Bh, L):
    """mg_denoiser_fwd_pair (both generator forwards of a GAN step in one grid of 64-frame tiles): outputs of both
    problems against separate launches, and the second problem's workspace through the ordinary backward -- input and
    parameter gradients against a forward + backward that ran alone."""
    from mixgan_tts_amd.autograd import DenoiserFn
    den, _ = _den(mg, manifest, tmp_path, ms)
    gen = torch.Generator(device="cuda").manual_seed(Bh * 1000 + L)
    xa = torch.randn(Bh, 80, L, device="cuda", generator=gen)
    xb = torch.randn(Bh, 80, L, device="cuda", generator=gen)
    cond = torch.randn(Bh, 256, L, device="cuda", generator=gen)
    spk = torch.randn(Bh, 256, device="cuda", generator=gen) if ms else None
    ta = torch.randint(0, 1000, (Bh,), device="cuda", generator=gen)
    tb = torch.randint(0, 1000, (Bh,), device="cuda", generator=gen)
    go = torch.randn(Bh, 80, L, device="cuda", generator=gen)
    params = [p for p in den._weight_table() if p is not None]

    def backward_of(pre):
        xg, cg = xb.clone().requires_grad_(), cond.clone().requires_grad_()
        for p in params:
            p.grad = None
        out = DenoiserFn.apply(den, xg, tb, cg, spk, pre, *params)
        (out * go).sum().backward()
        return out.detach(), xg.grad, cg.grad, [p.grad.clone() for p in params]
    with torch.no_grad():
        ref_a = den.run(xa, ta, cond, spk).clone()
    ref_b, ref_dx, ref_dc, ref_pg = backward_of(None)
    with torch.no_grad():
        both = den.run_pair(xa, ta, xb, tb, cond, spk)
    assert both is not None
    out_a, out_b, ws_b = both
    assert den.persist_status(2 * Bh, L)["error"] == 0
    assert_close(out_a.cpu(), ref_a.cpu(), TOL, "problem A (no save)")
    assert_close(out_b.cpu(), ref_b.cpu(), TOL, "problem B (saving)")
    got_b, dx, dc, pg = backward_of((out_b, ws_b))
    assert torch.equal(got_b, out_b)
    assert_close(dx.cpu(), ref_dx.cpu(), 5e-5, "d_x through the paired forward's workspace")
    assert_close(dc.cpu(), ref_dc.cpu(), 5e-5, "d_cond")
    for p, a, b in zip(params, pg, ref_pg):
        if b.abs().max() > 0:
            assert_close(a.cpu(), b.cpu(), 1e-4, "parameter gradient %s" % (tuple(p.shape),))
    with torch.no_grad():   # twice in a row: tickets and halo tags re-arm
        again = den.run_pair(xa, ta, xb, tb, cond, spk)
    assert torch.equal(again[0], out_a) and torch.equal(again[1], out_b)


@pytest.mark.parametrize("team", ["4", "2"])
@pytest.mark.parametrize("ms", [False, True])
def test_team_kernel_one_utterance(mg, manifest, tmp_path, monkeypatch, ms, team):
    """denoiser_team16.h: four workgroups per 16-frame tile, each owning 64 channels, h and g all-gathered through tagged
    granules every layer.  One 1000-frame utterance = 252 workgroups that wait for each other: the result must match the
    launch-per-layer kernels and the one-workgroup-per-tile kernel, be bit-identical launch after launch (also with a
    second stream saturating HBM next to it), and an utterance inside a small batch must equal the utterance alone."""
    monkeypatch.delenv("MG_PERSIST_NT", raising=False)
    monkeypatch.setenv("MG_PERSIST_TEAM", team)        # 4 members of 64 channels, or 2 of 128 (what 65-128 tiles use)
    den, W = _den(mg, manifest, tmp_path, ms)
    gen = torch.Generator(device="cuda").manual_seed(61)
    B, L = 1, 1000
    x = torch.randn(B, 1, 80, L, device="cuda", generator=gen)
    cond = torch.randn(B, 256, L, device="cuda", generator=gen)
    spk = torch.randn(B, 256, device="cuda", generator=gen) if ms else None
    t = torch.randint(0, 1000, (B,), device="cuda", generator=gen)
    with torch.no_grad():
        out = den(x, t, cond, spk)
        st = den.persist_status(B, L)
        assert st["error"] == 0 and st["ticket"] == 0 and st["done"] == 0 and st["launches"] == 1, st
        monkeypatch.setenv("MG_PERSIST_TEAM", "0")
        whole_tiles = den(x, t, cond, spk)
        monkeypatch.setenv("MG_DENOISER_PERSIST", "0")
        per_layer = den(x, t, cond, spk)
        monkeypatch.delenv("MG_DENOISER_PERSIST")
        monkeypatch.setenv("MG_PERSIST_TEAM", team)
        assert_close(out.cpu(), per_layer.cpu(), TOL, "team kernel vs per-layer kernels")
        assert_close(out.cpu(), whole_tiles.cpu(), TOL, "team kernel vs one workgroup per tile")
        ref = R.denoiser_forward(W, "", x.cpu(), t.cpu(), cond.cpu(), None if spk is None else spk.cpu())
        assert_close(out.cpu(), ref, TOL, "team kernel vs oracle")
        side = torch.cuda.Stream()
        junk = torch.empty(64 << 20, device="cuda")
        for i in range(6):
            if i >= 3:
                with torch.cuda.stream(side):
                    for _ in range(8):
                        junk.add_(1.0)
            again = den(x, t, cond, spk)
            assert torch.equal(again, out), "launch %d differs" % i
        torch.cuda.synchronize()
        # a batch of four 250-frame utterances (64 tiles: still teams) against each utterance alone
        B4, L4 = 4, 250
        x4 = torch.randn(B4, 1, 80, L4, device="cuda", generator=gen)
        c4 = torch.randn(B4, 256, L4, device="cuda", generator=gen)
        s4 = torch.randn(B4, 256, device="cuda", generator=gen) if ms else None
        t4 = torch.randint(0, 1000, (B4,), device="cuda", generator=gen)
        batch = den(x4, t4, c4, s4)
        for i in range(B4):
            one = den(x4[i:i + 1], t4[i:i + 1], c4[i:i + 1].contiguous(), None if s4 is None else s4[i:i + 1])
            assert torch.equal(one[0], batch[i]), "utterance %d alone differs from the batch" % i
        # ... and the teams reduce in the order of the wider kernels: the utterance inside a batch of 64-frame tiles
        xb = torch.cat([x, torch.randn(15, 1, 80, L, device="cuda", generator=gen)])
        cb = torch.cat([cond, torch.randn(15, 256, L, device="cuda", generator=gen)])
        sb = None if spk is None else torch.cat([spk, torch.randn(15, 256, device="cuda", generator=gen)])
        tb = torch.cat([t, torch.randint(0, 1000, (15,), device="cuda", generator=gen)])
        big = den(xb, tb, cb, sb)
        assert torch.equal(big[0], out[0]), "team kernel vs the same utterance in a B=16 batch (64-frame tiles)"
        # two utterances of 1000 frames (126 tiles): teams of 2 by the launcher's own choice; each equals the utterance alone
        monkeypatch.delenv("MG_PERSIST_TEAM")
        two = den(xb[:2].contiguous(), tb[:2].contiguous(), cb[:2].contiguous(), None if sb is None else sb[:2].contiguous())
        assert torch.equal(two[0], out[0]) and torch.equal(two[1], big[1])
        monkeypatch.setenv("MG_PERSIST_TEAM", team)
    den.check()


def test_two_streams_get_their_own_workspaces(mg, manifest, tmp_path):
    """Two p_sample chains of the same shape in flight on two streams (a server overlapping two requests on one module):
    the kernels keep tickets and hand-off buffers in the workspace, so each stream must get its own -- results equal to
    the chains run one after the other."""
    den, _ = _den(mg, manifest, tmp_path)
    gen = torch.Generator(device="cuda").manual_seed(77)
    B, L = 1, 640
    xs = [torch.randn(B, 1, 80, L, device="cuda", generator=gen) for _ in range(2)]
    cs = [torch.randn(B, 256, L, device="cuda", generator=gen) for _ in range(2)]
    t = torch.tensor([7], device="cuda")
    with torch.no_grad():
        ref = [den(xs[i], t, cs[i], None).clone() for i in range(2)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for s_ in streams:
            s_.wait_stream(torch.cuda.current_stream())
        outs = [None, None]
        for rep in range(6):
            for i, s_ in enumerate(streams):
                with torch.cuda.stream(s_):
                    outs[i] = den(xs[i], t, cs[i], None)
        torch.cuda.synchronize()
    assert len({k[4] for k in den._ws if k[:3] == (B, L, False)}) == 3      # the default stream's and the two side streams'
    for i in range(2):
        assert torch.equal(outs[i], ref[i])
    den.check()
