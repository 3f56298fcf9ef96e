"""GPU parity of the aux pre-training pieces: train-mode FFTBlock / Decoder / PostNet
forward AND backward against a reference run (fixture aux_train.npz, dropout masks replayed), plus the new
kernels against plain fp32 PyTorch.  Tolerance 1e-3 relative (BASELINE north_star) unless noted."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden, assert_close, assert_digest, load_seeded, hot_path_configs, T

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd
    return mixgan_tts_amd


def dev(a):
    return T(a).cuda()


class MaskTape:
    """DROPOUT_FN stand-in: the reference's [B, L, C] keep-masks, handed out channel-major in call order."""

    def __init__(self, g, prefix, channel_major=False):
        self.cm = channel_major      # PostNet drops on [B, C, L] already (transformer/Layers.py:130-134)
        ks = sorted((k for k in g if k.startswith(prefix + "/mask")), key=lambda k: int(k.rsplit("mask", 1)[1]))
        self.masks = [T(g[k]) for k in ks]
        self.i = 0

    def __call__(self, shape, p, device):
        m = self.masks[self.i] if self.cm else self.masks[self.i].transpose(1, 2).contiguous()
        self.i += 1
        assert tuple(m.shape) == tuple(shape), (m.shape, shape)
        return m.to(device)


@pytest.fixture()
def tape_hook(mg):
    def install(tape):
        mg.transformer.DROPOUT_FN = tape
    yield install
    mg.transformer.DROPOUT_FN = None


@pytest.mark.parametrize("M,N,K,batch,heads", [(128, 128, 16, 1, 1), (37, 53, 29, 2, 3), (300, 129, 128, 2, 2),
                                               (1, 1, 1, 1, 1), (130, 257, 300, 1, 2)])
@pytest.mark.parametrize("a_kc,b_kc", [(False, False), (True, False), (False, True), (True, True)])
def test_bgemm_vs_torch(mg, M, N, K, batch, heads, a_kc, b_kc):
    import ctypes
    gen = torch.Generator().manual_seed(M * 7 + N + K)
    A = torch.randn(batch, heads, M, K, generator=gen)
    B = torch.randn(batch, heads, K, N, generator=gen)
    C0 = torch.randn(batch, heads, M, N, generator=gen)
    ref = 0.5 * (A @ B) + C0
    Ad = (A if a_kc else A.transpose(2, 3)).contiguous().cuda()      # k contiguous, or m contiguous
    Bd = (B.transpose(2, 3) if b_kc else B).contiguous().cuda()      # k contiguous, or n contiguous
    Cd = C0.clone().cuda()
    L = mg._lib.lib()
    a_ms, a_ks = (K, 1) if a_kc else (1, M)
    b_ks, b_ns = (1, K) if b_kc else (N, 1)
    cp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    mg._lib.check(L.mg_bgemm(cp(Ad), cp(Bd), cp(Cd), M, N, K, batch, heads, a_ms, a_ks, heads * M * K, M * K,
                             b_ks, b_ns, heads * K * N, K * N, N, heads * M * N, M * N, 0.5, 1, None))
    assert_close(Cd.cpu(), ref, 2e-6 * max(1, K) ** 0.5, "bgemm")


@pytest.mark.parametrize("B,L,lens", [(2, 150, [150, 97]), (1, 33, [33]), (3, 64, [64, 1, 40])])
def test_attention_train_fwd_bwd_vs_torch(mg, B, L, lens):
    H, d = 2, 128
    gen = torch.Generator().manual_seed(L)
    qkv = torch.randn(B, 3 * H * d, L, generator=gen)
    go = torch.randn(B, H * d, L, generator=gen)
    pad = torch.arange(L)[None, :] >= torch.tensor(lens)[:, None]
    x = qkv.clone().requires_grad_()
    q, k, v = [t.view(B, H, d, L) for t in x.split(H * d, dim=1)]
    att = torch.einsum("bhdq,bhdk->bhqk", q, k) / d ** 0.5
    att = torch.softmax(att.masked_fill(pad[:, None, None, :], float("-inf")), dim=-1)
    ref = torch.einsum("bhqk,bhdk->bhdq", att, v).reshape(B, H * d, L)
    (ref * go).sum().backward()
    xg = qkv.clone().cuda().requires_grad_()
    out = mg.autograd.attention_train(xg, pad.to(torch.uint8).cuda(), H, d)
    assert_close(out.detach().cpu(), ref.detach(), 1e-5, "attention train fwd")
    # the streaming-softmax inference kernel computes the same function
    assert_close(mg.ops.attention(xg.detach(), pad.to(torch.uint8).cuda(), H, d).cpu(), ref.detach(), 1e-5, "vs eval kernel")
    (out * go.cuda()).sum().backward()
    assert_close(xg.grad.cpu(), x.grad, 1e-5, "attention train bwd")


@pytest.mark.parametrize("B,L,drop", [(2, 70, True), (1, 1, False), (3, 33, True)])
def test_layernorm_train_fwd_bwd_vs_torch(mg, B, L, drop):
    C = 256
    gen = torch.Generator().manual_seed(L)
    a, res, go = (torch.randn(B, C, L, generator=gen) for _ in range(3))
    gamma, beta = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
    pad = torch.arange(L)[None, :] >= torch.randint(1, L + 1, (B,), generator=gen)[:, None]
    keep = (torch.rand(B, C, L, generator=gen) >= 0.2) if drop else None
    ar, rr, gr, br = (t.clone().requires_grad_() for t in (a, res, gamma, beta))
    pre = (ar * keep / 0.8 if drop else ar) + rr
    ref = F.layer_norm(pre.transpose(1, 2), (C,), gr, br, 1e-5).transpose(1, 2).masked_fill(pad[:, None, :], 0)
    (ref * go).sum().backward()
    ag, rg, gg, bg = (t.clone().cuda().requires_grad_() for t in (a, res, gamma, beta))
    out = mg.autograd.layernorm_train(ag, rg, gg, bg, pad.to(torch.uint8).cuda(),
                                      keep.to(torch.uint8).cuda() if drop else None, 1 / 0.8 if drop else 1.0, 1e-5)
    assert_close(out.detach().cpu(), ref.detach(), 1e-5, "LN train fwd")
    (out * go.cuda()).sum().backward()
    for name, x, y in (("d_a", ag, ar), ("d_res", rg, rr), ("dgamma", gg, gr), ("dbeta", bg, br)):
        assert_close(x.grad.cpu(), y.grad, 2e-5, "LN train " + name)


@pytest.mark.parametrize("B,C,L,act,drop", [(3, 512, 45, "tanh", True), (2, 80, 130, None, True), (1, 16, 7, "tanh", False)])
def test_batchnorm_act_train_fwd_bwd_vs_torch(mg, B, C, L, act, drop):
    gen = torch.Generator().manual_seed(C + L)
    x = torch.randn(B, C, L, generator=gen) * 2 + 0.7
    go = torch.randn(B, C, L, generator=gen)
    gamma, beta = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
    keep = (torch.rand(B, C, L, generator=gen) >= 0.5) if drop else None
    xr, gr, br = (t.clone().requires_grad_() for t in (x, gamma, beta))
    y = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    y = torch.tanh(y) if act else y
    ref = y * keep / 0.5 if drop else y
    (ref * go).sum().backward()
    xg, gg, bg = (t.clone().cuda().requires_grad_() for t in (x, gamma, beta))
    out, mean, var = mg.autograd.batchnorm_act(xg, gg, bg, keep.to(torch.uint8).cuda() if drop else None,
                                               2.0 if drop else 1.0, act, 1e-5)
    assert_close(mean.cpu(), x.mean((0, 2)), 1e-5, "BN mean")
    assert_close(var.cpu(), x.var((0, 2), unbiased=False), 1e-5, "BN var")
    assert_close(out.detach().cpu(), ref.detach(), 1e-5, "BN train fwd")
    (out * go.cuda()).sum().backward()
    for name, a, b in (("dx", xg, xr), ("dgamma", gg, gr), ("dbeta", bg, br)):
        assert_close(a.grad.cpu(), b.grad, 5e-5, "BN train " + name)


def test_fftblock_train_golden(mg, manifest, tape_hook):
    g = golden("aux_train")
    blk = mg.FFTBlock(256, 2, 128, 128, 1024, 9, dropout=0.2)
    load_seeded(blk, manifest, "fftblock", 51)
    blk = blk.cuda().train()
    tape = MaskTape(g, "fft")
    tape_hook(tape)
    x = dev(g["fft/x"]).requires_grad_()
    y, _ = blk(x, mask=dev(g["fft/pad"]))
    assert tape.i == 2
    assert_close(y.detach().cpu(), g["fft/out"], TOL, "FFTBlock train")
    (y * dev(g["fft/go"])).sum().backward()
    assert_close(x.grad.cpu(), g["fft/d_x"], TOL, "FFTBlock train d_x")
    gd = {k[4:]: v for k, v in g.items() if k.startswith("fft/dw")}
    for k, p in blk.named_parameters():
        assert p.grad is not None, k
        if k.endswith("w_ks.bias"):   # softmax is invariant to a per-query shift: this gradient is exactly 0 in
            assert p.grad.abs().sum().item() < 1e-3 and gd["dw_sum/" + k][1] < 1e-3   # theory, rounding noise in both
            continue
        assert_digest(p.grad.cpu(), gd, k, TOL)


def test_decoder_train_golden(mg, manifest, tape_hook):
    g = golden("aux_train")
    _, pre, mc, _ = hot_path_configs(stats_dir=".", max_seq_len=int(g["dec/max_seq_len"]))
    dec = mg.Decoder(mc)
    load_seeded(dec, manifest, "decoder", 52)
    dec = dec.cuda().train()
    tape = MaskTape(g, "dec")
    tape_hook(tape)
    x = dev(g["dec/x"]).requires_grad_()
    y = dec(x, dev(g["dec/pad"]))
    assert tape.i == 12 and tuple(y.shape) == tuple(g["dec/out"].shape)
    assert_close(y.detach().cpu(), g["dec/out"], TOL, "Decoder train")
    (y * dev(g["dec/go"])).sum().backward()
    assert_close(x.grad.cpu(), g["dec/d_x"], TOL, "Decoder train d_x")
    for k, p in dec.named_parameters():
        if p.requires_grad and not k.endswith("w_ks.bias"):
            ref = g["dec/dw_sum/" + k]
            got = p.grad.double()
            assert abs(got.sum().item() - ref[0]) <= TOL * (abs(ref[1]) + 1e-30), k
            assert abs(got.abs().sum().item() - ref[1]) <= TOL * (abs(ref[1]) + 1e-30), k


def test_postnet_train_golden(mg, manifest, tape_hook):
    g = golden("aux_train")
    pn = mg.PostNet()
    load_seeded(pn, manifest, "postnet", 53)
    pn = pn.cuda().train()
    tape = MaskTape(g, "pn", channel_major=True)
    tape_hook(tape)
    x = dev(g["pn/x"]).requires_grad_()
    y = pn(x)
    assert tape.i == 5
    assert_close(y.detach().cpu(), g["pn/out"], TOL, "PostNet train")
    (y * dev(g["pn/go"])).sum().backward()
    assert_close(x.grad.cpu(), g["pn/d_x"], TOL, "PostNet train d_x")
    gd = {k[3:]: v for k, v in g.items() if k.startswith("pn/dw")}
    for k, p in pn.named_parameters():
        if k.endswith("conv.bias"):   # a bias in front of BatchNorm has exactly zero gradient: rounding noise in both
            assert p.grad.abs().sum().item() < 1e-2 and gd["dw_sum/" + k][1] < 1e-2
            continue
        assert_digest(p.grad.cpu(), gd, k, TOL)
    for k, b in pn.named_buffers():
        if not k.endswith("num_batches_tracked"):
            assert_close(b.cpu(), g["pn/buf/" + k], 1e-5, k)
        else:
            assert int(b) == int(g["pn/buf/" + k])
    # eval after training uses the updated running statistics (folded-BN cache must notice)
    pn.eval()
    assert torch.isfinite(pn(x.detach())).all()


def test_aux_trainer_step_vs_oracle(mg, manifest, tmp_path):
    """One `--model aux` step (train.py:97-128) on the HIP path: losses, the gradient reaching the linguistic
    encoder's output, parameter gradients and the ScheduledOptim update against the oracle chain + torch Adam."""
    from helpers import write_stats, seeded, Tape
    from oracle import refmath as R, schedule as S
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    args, pre, mc, tr = hot_path_configs("aux", 4, stats_dir=stats, max_seq_len=64)
    tr = dict(tr)
    tr["optimizer_fs2"] = {"betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0, "warm_up_step": 4000,
                           "anneal_steps": [300000], "anneal_rate": 0.3}
    tr.setdefault("optimizer", {})["grad_clip_thresh"] = 1.0
    gen = torch.Generator().manual_seed(5)
    B, L = 2, 50
    cond = torch.randn(B, L, 256, generator=gen)
    mel = torch.rand(B, L, 80, generator=gen) * 8 - 9
    pad = torch.arange(L)[None, :] >= torch.tensor([50, 36])[:, None]
    net = mg.MixGANTTS(args, pre, mc, tr)
    W = {}
    for name, mod, seed, pfx in (("decoder", net.decoder, 71, "decoder."), ("postnet", net.postnet, 72, "postnet.")):
        load_seeded(mod, manifest, name, seed)
        w, _ = seeded(manifest, name, seed, prefix=pfx, requires_grad=True)
        W.update(w)
    W["decoder.position_enc"] = R.sinusoid_table(65, 256)[None]
    with torch.no_grad():
        net.mel_linear.weight.copy_(torch.randn(80, 256, generator=gen) / 16)
        net.mel_linear.bias.copy_(torch.randn(80, generator=gen) * 0.1)
    W["mel_linear.weight"] = net.mel_linear.weight.detach().clone().requires_grad_()
    W["mel_linear.bias"] = net.mel_linear.bias.detach().clone().requires_grad_()
    net = net.cuda().train()
    # dropout masks (12 in the decoder on [B,L,256], 5 in the PostNet on [B,C,L]) and the 4 q_sample noises
    masks = [(torch.rand(B, L, 256, generator=gen) >= 0.2) for _ in range(12)]
    pmasks = [(torch.rand(B, c, L, generator=gen) >= 0.5) for c in (512, 512, 512, 512, 80)]
    noises = [torch.randn(B, 1, 80, L, generator=gen) for _ in range(4)]

    seq_o = iter([m.float() for m in masks] + [m.float() for m in pmasks])
    seq_g = iter([m.transpose(1, 2).contiguous() for m in masks] + pmasks)
    mg.transformer.DROPOUT_FN = lambda shape, p, device: next(seq_g).to(torch.uint8).to(device)
    try:
        net.diffusion.noise_fn = Tape([n.numpy() for n in noises])
        params = [p for n_, p in net.named_parameters() if n_.split(".")[0] in ("decoder", "mel_linear", "postnet")
                  and p.requires_grad]
        trainer = mg.AuxTrainer(net, tr, mc, current_step=0, params=params)
        cg = cond.cuda().requires_grad_()
        mel_loss, post_loss, coarse = trainer.acoustic_losses(cg, mel.cuda(), pad.cuda())
        (mel_loss + post_loss).backward()
    finally:
        mg.transformer.DROPOUT_FN = None
    buf = {k: T(v) for k, v in S.diffusion_buffers(S.beta_schedule("vpsde", 4, 0.1, 40, 0.008)).items()}
    buf["spec_min"], buf["spec_max"] = T(e["spec_min"])[None, None], T(e["spec_max"])[None, None]
    co = cond.clone().requires_grad_()
    ml_o, pl_o, coarse_o = R.aux_acoustic_losses(W, buf, co, mel, pad, 64, 4, R.NoiseTape(noises),
                                                 lambda shape, p: next(seq_o))
    (ml_o + pl_o).backward()
    assert_close(coarse.detach().cpu(), coarse_o.detach(), TOL, "aux coarse mel (train mode)")
    assert abs(mel_loss.item() - ml_o.item()) <= TOL * abs(ml_o.item())
    assert abs(post_loss.item() - pl_o.item()) <= TOL * abs(pl_o.item())
    assert_close(cg.grad.cpu(), co.grad, TOL, "d loss / d encoder output")
    named = dict(net.named_parameters())
    for k, w in W.items():
        if w.requires_grad and w.grad is not None and not (k.endswith("w_ks.bias") or k.endswith("conv.bias")):
            gref = w.grad.double()
            got = named[k].grad.double().cpu()
            scale = gref.abs().sum().item() + 1e-30
            assert (got - gref).abs().sum().item() <= 2 * TOL * scale, k
    # optimizer: the clipped-gradient ScheduledOptim update equals torch Adam on the oracle gradients
    before = {k: named[k].detach().clone() for k in ("mel_linear.weight", "decoder.layer_stack.0.slf_attn.fc.weight")}
    trainer.bucket.gather()                         # what AuxTrainer.step does: flat gradients, then clip + Adam on them
    lr = trainer.opt.step(max_grad_norm=1.0)
    assert lr == pytest.approx(256 ** -0.5 * 4000 ** -1.5)
    for k, b in before.items():
        assert not torch.equal(b, named[k].detach())
        # first Adam step moves every element by ~lr (sign of the gradient)
        step = (named[k].detach() - b).abs().max().item()
        assert step <= lr * 1.001 + 1e-7 * max(1.0, b.abs().max().item())   # + fp32 rounding of the weight


class _StubEncoder(torch.nn.Module):
    """Stands in for the out-of-scope LinguisticEncoder: a trainable conditioner (so gradients must reach it)."""

    def __init__(self, cond):
        super().__init__()
        self.cond = torch.nn.Parameter(cond)

    def forward(self, texts, src_lens, word_boundaries, src_masks, src_w_lens, src_w_masks, mel_masks, max_mel_len,
                attn_priors, p_targets, e_targets, d_targets, p_control, d_control):
        return self.cond, None, None, None, None, mel_masks.sum(1), mel_masks, None, None


def test_mixgantts_forward_aux_train_mode(mg, tmp_path):
    """`MixGANTTS.forward` with args.model == "aux" in train mode (model/mixgantts.py:136-146): slot 0 is the
    diffuse_trace list (T+1 normalised mels), slot 15 / the third return value the PostNet-refined coarse mel, and the
    whole chain is differentiable down to the linguistic encoder's output."""
    from helpers import write_stats
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    args, pre, mc, tr = hot_path_configs("aux", 4, stats_dir=stats, max_seq_len=1000)
    B, L = 2, 40
    gen = torch.Generator().manual_seed(4)
    enc = _StubEncoder(torch.randn(B, L, 256, generator=gen))
    net = mg.MixGANTTS(args, pre, mc, tr, linguistic_encoder=enc).cuda().train()
    mels = (torch.rand(B, L, 80, generator=gen) * 8 - 9).cuda()
    mel_lens = torch.tensor([40, 27]).cuda()
    z = torch.zeros(B, 5, dtype=torch.long).cuda()
    out, p_t, coarse = net(None, z, torch.tensor([5, 5]).cuda(), 5, None, torch.tensor([3, 3]).cuda(), 3,
                           mels=mels, mel_lens=mel_lens, max_mel_len=L)
    assert len(out) == 16 and isinstance(out[0], list) and len(out[0]) == 5      # T + 1 trace entries
    assert out[1] == (None, None, None) and out[3] is None
    assert out[15] is coarse and tuple(coarse.shape) == (B, L, 80) and coarse.requires_grad
    pad = torch.arange(L)[None, :].cuda() >= mel_lens[:, None]
    assert torch.equal(out[9], pad)
    for tr_ in out[0]:
        assert tuple(tr_.shape) == (B, L, 80) and float(tr_.detach()[1, 27:].abs().max()) == 0.0   # padded frames masked
    assert float(out[0][0].detach().abs().max()) <= 1.0                                              # clamped normalised mel
    loss = mg.losses._L1Fn.apply(coarse, mels)
    for t_ in out[0]:
        loss = loss + mg.losses.get_mel_loss(net.diffusion.denorm_spec(t_), mels, pad)
    loss.backward()
    assert enc.cond.grad is not None and torch.isfinite(enc.cond.grad).all() and float(enc.cond.grad.abs().sum()) > 0
    assert all(p.grad is not None for n_, p in net.named_parameters()
               if n_.split(".")[0] in ("decoder", "mel_linear", "postnet") and p.requires_grad)
    # eval mode + no_grad keeps the inference behaviour (no autograd graph, running BatchNorm statistics)
    net.eval()
    with torch.no_grad():
        out_e, _, coarse_e = net(None, z, torch.tensor([5, 5]).cuda(), 5, None, torch.tensor([3, 3]).cuda(), 3,
                                 mels=mels, mel_lens=mel_lens, max_mel_len=L)
    assert not coarse_e.requires_grad and len(out_e[0]) == 5
