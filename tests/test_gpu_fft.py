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
    with pytest.raises(NotImplementedError):
        blk.train()(dev(g["x"]), mask=dev(g["pad"]))


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


@pytest.mark.parametrize("B,L,lens", [(2, 300, [300, 171]), (3, 64, [64, 1, 33]), (1, 129, [129])])
def test_attention_and_layernorm_vs_oracle(mg, manifest, B, L, lens):
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
