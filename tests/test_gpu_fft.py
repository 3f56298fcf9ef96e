"""GPU parity of the FFT-block path (transformer/*, eval mode) against the reference fixtures and
the CPU oracle: FFTBlock, Decoder below/above max_seq_len, PostNet, and the streaming-softmax
attention kernel at lengths spanning several key tiles."""
import numpy as np
import pytest
import torch

from helpers import golden, T, seeded, assert_close, hot_path_configs, load_seeded
from oracle import refmath as R

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd as m
    assert torch.cuda.is_available()
    m.lib()
    return m


def dev(a):
    return T(a).cuda()


def test_fftblock_golden(mg, manifest):
    g = golden("fftblock")
    blk = mg.FFTBlock(256, 2, 128, 128, 1024, 9, dropout=0.2)
    ck = load_seeded(blk, manifest, "fftblock", 51)
    np.testing.assert_allclose(ck, g["wsum"], rtol=1e-12)
    blk = blk.cuda().eval()
    y, _ = blk(dev(g["x"]), mask=dev(g["pad"]))
    assert_close(y.cpu(), g["out"], TOL, "FFTBlock")


def test_decoder_and_postnet_golden(mg, manifest):
    g = golden("decoder")
    _, pre, mc, _ = hot_path_configs(stats_dir=".", max_seq_len=int(g["max_seq_len"]))
    dec = mg.Decoder(mc)
    load_seeded(dec, manifest, "decoder", 52)
    dec = dec.cuda().eval()
    for tag in ("short", "long"):
        y = dec(dev(g[tag + "_x"]), dev(g[tag + "_pad"]))
        assert_close(y.cpu(), g[tag + "_out"], 5e-5, "Decoder " + tag)
    g = golden("postnet")
    pn = mg.PostNet()
    load_seeded(pn, manifest, "postnet", 53)
    pn = pn.cuda().eval()
    assert_close(pn(dev(g["x"])).cpu(), g["out"], TOL, "PostNet")


# workgroup forms: 128 queries / 64 and 128 queries with the keys split between wave pairs / 256 queries (8 waves)
@pytest.mark.parametrize("ksplit,wide", [("0", "0"), ("1", "0"), ("2", "0"), ("0", "1")])
@pytest.mark.parametrize("B,L,lens", [(2, 300, [300, 171]), (3, 64, [64, 1, 33]), (1, 129, [129])])
def test_attention_and_layernorm_vs_oracle(mg, manifest, monkeypatch, B, L, lens, ksplit, wide):
    monkeypatch.setenv("MG_ATTENTION_KSPLIT", ksplit)
    monkeypatch.setenv("MG_ATTENTION_WIDE", wide)
    W, _ = seeded(manifest, "fftblock", 99)
    gen = torch.Generator().manual_seed(L)
    x = torch.randn(B, L, 256, generator=gen)
    pad = torch.arange(L)[None, :] >= torch.tensor(lens)[:, None]
    ref = R.mha_forward(W, "slf_attn.", x, pad)
    blk = mg.FFTBlock(256, 2, 128, 128, 1024, 9)
    load_seeded(blk, manifest, "fftblock", 99)
    blk = blk.cuda().eval()
    xc = mg.ops.transpose_bml(x.cuda(), False)
    y = blk.slf_attn.forward_cm(xc, pad.to(torch.uint8).cuda())
    assert_close(mg.ops.transpose_bml(y, True).cpu(), ref, TOL, "MHA + post-LN")
    full = R.fft_block(W, "", x, pad)
    out, _ = blk(x.cuda(), mask=pad.cuda())
    assert_close(out.cpu(), full, TOL, "FFT block")


class _StubEncoder(torch.nn.Module):
    """Stands in for the out-of-scope LinguisticEncoder: returns a fixed conditioner and the masks."""

    def __init__(self, cond):
        super().__init__()
        self.cond = cond

    def forward(self, texts, src_lens, word_boundaries, src_masks, src_w_lens, src_w_masks, mel_masks, max_mel_len,
                attn_priors, p_targets, e_targets, d_targets, p_control, d_control):
        lens = (mel_masks).sum(1)
        return self.cond, None, None, None, None, lens, mel_masks, None, None


@pytest.mark.parametrize("model", ["naive", "shallow"])
def test_mixgantts_forward_inference_vs_oracle(mg, manifest, tmp_path, model):
    """MixGANTTS.forward (model/mixgantts.py:55-180) with an injected conditioner: output-list layout and the
    final mel against the oracle chain (coarse mel -> shallow diffusion / naive sampling)."""
    from helpers import write_stats, Tape
    from oracle import schedule as S
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    args, pre, mc, tr = hot_path_configs(model, 4, stats_dir=stats, max_seq_len=1000)
    gen = torch.Generator().manual_seed(11)
    B, L = 2, 70
    cond = torch.randn(B, L, 256, generator=gen)
    mel_lens = torch.tensor([70, 51])
    net = mg.MixGANTTS(args, pre, mc, tr, linguistic_encoder=_StubEncoder(cond.cuda()))
    W = {}
    if model == "shallow":
        for name, mod, seed, pfx in (("decoder", net.decoder, 71, "decoder."), ("postnet", net.postnet, 72, "postnet.")):
            _, pre2, mc2, _ = hot_path_configs(stats_dir=".", max_seq_len=48)
            man = dict(manifest)
            if name == "decoder":   # fixture manifest was recorded at max_seq_len=48; only trainable shapes matter
                pass
            load_seeded(mod, man, name, seed)
            w, _ = seeded(man, name, seed, prefix=pfx)
            W.update(w)
        W["decoder.position_enc"] = R.sinusoid_table(1001, 256)[None]
        with torch.no_grad():
            net.mel_linear.weight.copy_(torch.randn(80, 256, generator=gen) / 16)
            net.mel_linear.bias.copy_(torch.randn(80, generator=gen) * 0.1)
        W["mel_linear.weight"], W["mel_linear.bias"] = net.mel_linear.weight.detach().clone(), net.mel_linear.bias.detach().clone()
    load_seeded(net.diffusion, manifest, "diffusion_naive_ms0", 73)
    Wd, _ = seeded(manifest, "diffusion_naive_ms0", 73)
    net = net.cuda().eval()
    n_noise = 5 if model == "naive" else 5
    noises = [torch.randn(B, 1, 80, L, generator=gen) for _ in range(n_noise)]
    net.diffusion.noise_fn = Tape([n.numpy() for n in noises])
    z = torch.zeros(B, 5, dtype=torch.long).cuda()
    with torch.no_grad():
        out, p_t, coarse = net(None, z, torch.tensor([5, 5]).cuda(), 5, None, torch.tensor([3, 3]).cuda(), 3,
                               mels=None, mel_lens=mel_lens.cuda(), max_mel_len=L)
    assert len(out) == 16 and out[1] == (None, None, None)
    pad = torch.arange(L)[None, :] >= mel_lens[:, None]
    assert torch.equal(out[9].cpu(), pad)                       # slot 9: mel_masks, True = pad
    buf = {k: T(v) for k, v in S.diffusion_buffers(S.beta_schedule("vpsde", 4, 0.1, 40, 0.008)).items()}
    buf["spec_min"], buf["spec_max"] = T(e["spec_min"])[None, None], T(e["spec_max"])[None, None]
    coarse_ref = R.coarse_mel(W, cond, pad, 1000) if model == "shallow" else None
    ref, *_ = R.diffusion_forward(Wd, buf, model, 4, None, cond, None, pad, coarse_ref, R.NoiseTape(noises))
    if model == "shallow":
        assert_close(coarse.cpu(), coarse_ref, 5e-5, "coarse mel")
        assert_close(out[15].cpu(), coarse_ref, 5e-5, "postnet_outputs slot")
    assert_close(out[0].cpu(), ref, 1e-4, "final mel")


def test_attention_long_form_L4000(mg, manifest):
    """cfg5 length (L=4000): the reference materialises a [2B, L, L] score tensor;
    the streaming kernel must agree with it without ever holding more than a 64-key tile."""
    W, _ = seeded(manifest, "fftblock", 123)
    gen = torch.Generator().manual_seed(4000)
    B, L = 1, 4000
    x = torch.randn(B, L, 256, generator=gen)
    pad = torch.arange(L)[None, :] >= torch.tensor([3777])[:, None]
    ref = R.mha_forward(W, "slf_attn.", x, pad)
    blk = mg.FFTBlock(256, 2, 128, 128, 1024, 9)
    load_seeded(blk, manifest, "fftblock", 123)
    blk = blk.cuda().eval()
    y = blk.slf_attn.forward_cm(mg.ops.transpose_bml(x.cuda(), False), pad.to(torch.uint8).cuda())
    assert_close(mg.ops.transpose_bml(y, True).cpu(), ref, 3e-5, "MHA L=4000")


@pytest.mark.parametrize("wide", ["0", "1"])   # 128- / 256-query workgroups
@pytest.mark.parametrize("B,L,lens", [(2, 300, [300, 171]), (3, 64, [64, 1, 33]), (1, 129, [129]), (1, 4000, [3777])])
def test_attention_f16_mfma_path(mg, manifest, monkeypatch, B, L, lens, wide):
    """BASELINE configs[4]: attention with fp16 MFMA operands (fp32 accumulate / statistics) against the exact
    fp32 kernel at the north_star tolerance, including the L = 4000 long-form length."""
    monkeypatch.setenv("MG_ATTENTION_WIDE", wide)
    H, d = 2, 128
    gen = torch.Generator().manual_seed(L + B)
    qkv = torch.randn(B, 3 * H * d, L, generator=gen).cuda()
    pad = (torch.arange(L)[None, :] >= torch.tensor(lens)[:, None]).to(torch.uint8).cuda()
    ref = mg.ops.attention(qkv, pad, H, d)
    got = mg.ops.attention(qkv, pad, H, d, precision="f16")
    assert torch.isfinite(got).all()
    assert_close(got.cpu(), ref.cpu(), 1e-3, "attention f16 vs fp32, L=%d" % L)   # north_star tolerance
    with pytest.raises(ValueError):
        mg.ops.attention(qkv, pad, H, d, precision="fp8")


def test_decoder_golden_with_f16_attention(mg, manifest):
    g = golden("decoder")
    _, pre, mc, _ = hot_path_configs(stats_dir=".", max_seq_len=int(g["max_seq_len"]))
    dec = mg.Decoder(mc)
    load_seeded(dec, manifest, "decoder", 52)
    dec = dec.cuda().eval()
    dec.set_attention_precision("f16")
    for tag in ("short", "long"):
        y = dec(dev(g[tag + "_x"]), dev(g[tag + "_pad"]))
        assert_close(y.cpu(), g[tag + "_out"], 1e-3, "Decoder (f16 attention) " + tag)
