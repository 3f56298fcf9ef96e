"""GPU parity of the HiFi-GAN generator mirror (hifigan/models.py:112-173)
against the reference-generated fixture and the oracle.  Tolerance: 1e-3 relative (BASELINE north_star)."""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden, assert_close, load_seeded, seeded, T
from oracle import refmath as R

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd
    return mixgan_tts_amd


def _h():
    return types.SimpleNamespace(**R.HIFIGAN_V1)


def _seeded_generator(mg, manifest, g):
    G = mg.vocoder.Generator(_h())
    load_seeded(G, manifest, "hifigan", 81)
    sd = G.state_dict()
    for k in g:
        if k.startswith("gain/"):
            sd[k[5:]] = torch.from_numpy(g[k])
    G.load_state_dict(sd)
    return G.cuda().eval()


def test_generator_matches_reference_fixture(mg, manifest):
    g = golden("hifigan")
    G = _seeded_generator(mg, manifest, g)
    y = G(torch.from_numpy(g["mel"]).cuda())
    assert tuple(y.shape) == tuple(g["wav"].shape)
    assert_close(y.cpu(), g["wav"], TOL, "hifigan Generator")
    # remove_weight_norm (utils/model.py:124) leaves the function unchanged and the keys as the reference's
    G.remove_weight_norm()
    assert "conv_pre.weight" in G.state_dict() and "conv_pre.weight_g" not in G.state_dict()
    y2 = G(torch.from_numpy(g["mel"]).cuda())
    assert_close(y2.cpu(), g["wav"], TOL, "hifigan Generator after remove_weight_norm")


@pytest.mark.parametrize("B,L", [(1, 1), (3, 37), (1, 200)])
def test_generator_vs_oracle(mg, manifest, B, L):
    g = golden("hifigan")
    G = _seeded_generator(mg, manifest, g)
    W, _ = seeded(manifest, "hifigan", 81)
    for k in g:
        if k.startswith("gain/"):
            W[k[5:]] = T(g[k])
    mel = torch.from_numpy(np.random.default_rng(L).uniform(-11.5, 2.0, (B, 80, L)).astype(np.float32))
    ref = R.hifigan_forward(W, mel)
    y = G(mel.cuda())
    assert tuple(y.shape) == (B, 1, 256 * L)
    assert_close(y.cpu(), ref, TOL, "hifigan B=%d L=%d" % (B, L))


@pytest.mark.parametrize("K,dil,Ci,Co,L", [(3, 1, 64, 64, 333), (3, 5, 32, 32, 257), (7, 3, 128, 128, 400),
                                           (11, 5, 32, 32, 129), (11, 1, 256, 256, 64), (7, 1, 32, 1, 515),
                                           (7, 5, 64, 64, 9)])
def test_dilated_conv_with_fused_activations(mg, K, dil, Ci, Co, L):
    """mg_conv1d_fwd_ex: leaky_relu(x, in_slope) -> dilated conv -> *alpha + bias -> leaky_relu(act_slope) -> + add."""
    gen = torch.Generator().manual_seed(K * 100 + dil)
    x = torch.randn(2, Ci, L, generator=gen)
    w = torch.randn(Co, Ci, K, generator=gen) / (Ci * K) ** 0.5
    b = torch.randn(Co, generator=gen)
    add = torch.randn(2, Co, L, generator=gen)
    pad = (K * dil - dil) // 2
    ref = F.leaky_relu(F.conv1d(F.leaky_relu(x, 0.1), w, None, 1, pad, dil) * 0.5 + b[None, :, None], 0.3) + add
    wp = mg.ops.pack_conv_weight(w.cuda(), mg.ops.PACK_PLAIN)
    y = mg.ops.conv1d_packed(x.cuda(), wp, b.cuda(), Co, K, 1, pad, act="lrelu_s", act_slope=0.3, alpha=0.5,
                             add=add.cuda(), dilation=dil, in_slope=0.1)
    assert_close(y.cpu(), ref, 1e-5, "dilated conv")


@pytest.mark.parametrize("K,u,Ci,Co,L", [(16, 8, 512, 256, 13), (16, 8, 64, 32, 40), (4, 2, 128, 64, 77),
                                         (4, 2, 64, 32, 1)])
def test_conv_transpose_by_zero_insertion(mg, K, u, Ci, Co, L):
    """ConvTranspose1d(k, stride u, padding (k-u)/2) of hifigan/models.py:121-127 == zero insertion + stride-1
    conv with the transposed, tap-flipped weight."""
    gen = torch.Generator().manual_seed(K + L)
    x = torch.randn(2, Ci, L, generator=gen)
    w = torch.randn(Ci, Co, K, generator=gen) / (Ci * K / u) ** 0.5
    b = torch.randn(Co, generator=gen)
    pad = (K - u) // 2
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, u, pad)
    wp = mg.ops.pack_conv_weight(w.cuda(), mg.ops.PACK_DGRAD)
    z = mg.ops.upsample_zero(x.cuda(), u, (L - 1) * u + 1, slope=0.1)
    y = mg.ops.conv1d_packed(z, wp, b.cuda(), Co, K, 1, K - 1 - pad, Lout=ref.shape[2])
    assert_close(y.cpu(), ref, 1e-5, "conv transpose")


@pytest.mark.parametrize("u,Ci,Co,B,L", [(8, 512, 256, 2, 13), (8, 64, 32, 3, 133), (2, 128, 64, 2, 77), (2, 64, 32, 1, 1),
                                          (4, 32, 16, 2, 300), (8, 24, 4, 1, 50), (2, 16, 2, 2, 700)])
def test_conv_transpose_polyphase(mg, u, Ci, Co, B, L):
    """mg_conv_transpose1d_fwd == alpha * F.conv_transpose1d(leaky_relu(x), w, None, u, u/2) + bias."""
    gen = torch.Generator().manual_seed(u * 1000 + L)
    x = torch.randn(B, Ci, L, generator=gen)
    w = torch.randn(Ci, Co, 2 * u, generator=gen) / (2 * Ci) ** 0.5
    b = torch.randn(Co, generator=gen)
    ref = 0.5 * F.conv_transpose1d(F.leaky_relu(x, 0.1), w, None, u, u // 2) + b[None, :, None]
    wp = mg.ops.pack_conv_transpose_weight(w.cuda())
    y = mg.ops.conv_transpose1d_packed(x.cuda(), wp, b.cuda(), Co, u, in_slope=0.1, alpha=0.5)
    assert tuple(y.shape) == tuple(ref.shape)
    assert_close(y.cpu(), ref, 1e-5, "polyphase conv transpose")


def test_generator_cpu_input_fails_loudly(mg):
    G = mg.vocoder.Generator(_h())
    with pytest.raises(mg._lib.MixganHipError):
        G(torch.zeros(1, 80, 4))


def test_generator_full_size_properties(mg, manifest):
    """BASELINE-size utterances (L = 1000 frames -> 256,000 samples): rows of a batch are independent (bitwise:
    a row alone equals the row inside a batch of 3), the output length is exactly 256 L, values stay in [-1, 1],
    and the first 25 frames' worth of samples only depend on the first frames (finite receptive field), checked
    against the oracle on a truncated input."""
    g = golden("hifigan")
    G = _seeded_generator(mg, manifest, g)
    gen = torch.Generator().manual_seed(4)
    mel = (torch.rand(3, 80, 1000, generator=gen) * 13.5 - 11.5)
    y = G(mel.cuda())
    assert tuple(y.shape) == (3, 1, 256000)
    assert torch.isfinite(y).all() and y.abs().max().item() <= 1.0
    for b in range(3):
        assert torch.equal(G(mel[b:b + 1].cuda())[0], y[b]), "row %d depends on its batch" % b
    # receptive field of HiFi-GAN V1 at the mel rate is < 40 frames on each side: compare the first 25 frames'
    # samples with the oracle run on the first 80 frames only
    W, _ = seeded(manifest, "hifigan", 81)
    for k in g:
        if k.startswith("gain/"):
            W[k[5:]] = T(g[k])
    ref = R.hifigan_forward(W, mel[:1, :, :80])
    assert_close(y[:1, :, :25 * 256].cpu(), ref[:, :, :25 * 256], TOL, "full-size head vs oracle on a truncated input")


def test_get_vocoder_and_vocoder_infer(mg, manifest, tmp_path):
    """utils/model.py:74-126: checkpoint round trip through get_vocoder (weights_only load, remove_weight_norm) and
    the int16 conversion / length cropping of vocoder_infer."""
    import json
    g = golden("hifigan")
    G = _seeded_generator(mg, manifest, g)
    ck = tmp_path / "generator_LJSpeech.pth.tar"
    torch.save({"generator": {k: v.cpu() for k, v in G.state_dict().items()}}, ck)
    cfg = tmp_path / "config.json"
    cfg.write_text(json.dumps(R.HIFIGAN_V1))
    mc = {"vocoder": {"model": "HiFi-GAN", "speaker": "LJSpeech"}}
    voc = mg.vocoder.get_vocoder(mc, "cuda", config_path=str(cfg), checkpoint_path=str(ck))
    assert "conv_pre.weight" in voc.state_dict() and not voc.training
    mel = torch.from_numpy(g["mel"]).cuda()
    pre = {"preprocessing": {"audio": {"max_wav_value": 32768.0}}}
    wavs = mg.vocoder.vocoder_infer(mel, voc, mc, pre, lengths=[3000, 1234])
    ref = (g["wav"][:, 0] * 32768.0).astype("int16")
    assert [w.dtype.name for w in wavs] == ["int16", "int16"] and [len(w) for w in wavs] == [3000, 1234]
    for w, r in zip(wavs, ref):
        assert np.abs(w.astype(np.int32) - r[:len(w)].astype(np.int32)).max() <= 1     # fp32 rounding at .0 boundaries
    with pytest.raises(NotImplementedError):
        mg.vocoder.get_vocoder({"vocoder": {"model": "MelGAN", "speaker": "LJSpeech"}}, "cuda")
