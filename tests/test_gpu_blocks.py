"""Stand-alone forwards of the denoiser's building blocks (model/blocks.py:894-913,1157-1176) on the HIP path against
reference fixtures: `ResidualBlock.forward` with gradients (resblock_ms{0,1}.npz: the same fused layer kernel the
Denoiser launches, one layer), `DiffusionEmbedding.forward` (step_embedding.npz) and `Mish.forward`."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden, T, assert_close, assert_digest, load_seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd as m
    assert torch.cuda.is_available()
    return m


@pytest.mark.parametrize("ms", [0, 1])
def test_residual_block_forward_and_gradients(mg, manifest, ms):
    name = "resblock_ms%d" % ms
    g = golden(name)
    blk = mg.blocks.ResidualBlock(256, 256, dropout=0.2, multi_speaker=bool(ms))
    ck = load_seeded(blk, manifest, name, 11 + ms)
    np.testing.assert_allclose(ck, g["wsum"], rtol=1e-12)
    blk = blk.cuda()
    leaf = lambda k: T(g[k]).cuda().requires_grad_()  # noqa: E731
    x, cond, step = leaf("x"), leaf("cond"), leaf("step")
    spk = leaf("spk") if ms else None
    nx, sk = blk(x, cond, step, spk)
    assert_close(nx.detach().cpu(), g["out_x"], 2e-5, "x out")
    assert_close(sk.detach().cpu(), g["out_skip"], 2e-5, "skip out")
    ((nx * T(g["gx"]).cuda()).sum() + (sk * T(g["gs"]).cuda()).sum()).backward()
    assert_close(x.grad.cpu(), g["d_x"], 5e-5, "d_x")
    assert_close(cond.grad.cpu(), g["d_cond"], 5e-5, "d_cond")
    assert_close(step.grad.cpu(), g["d_step"], 5e-5, "d_step")
    if ms:
        assert_close(spk.grad.cpu(), g["d_spk"], 5e-5, "d_spk")
    for k, p in blk.named_parameters():
        assert_digest(p.grad, g, k, 1e-4)
    # inference form (no saves) gives the same numbers
    with torch.no_grad():
        nx2, sk2 = blk(x.detach(), cond.detach(), step.detach(), None if spk is None else spk.detach())
    assert torch.equal(nx2, nx.detach()) and torch.equal(sk2, sk.detach())


def test_diffusion_embedding_forward(mg):
    g = golden("step_embedding")
    emb = mg.blocks.DiffusionEmbedding(256)(T(g["t"]).cuda())
    assert_close(emb.cpu()[:3], g["emb"][:3], 1e-6, "step embedding, t <= 3")
    # t = 99, 999: the frequency table is the correctly rounded one (blocks.DiffusionEmbedding.frequencies), within
    # one ulp of a frequency (6e-5 rad at t = 999) of whatever host exp the reference run used
    assert_close(emb.cpu(), g["emb"], 1e-5, "step embedding")


def test_mish_forward_backward(mg):
    gen = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(1000, generator=gen) * 4, torch.tensor([-30.0, -20.0, 0.0, 19.9, 20.1, 50.0])])
    go = torch.randn(x.shape, generator=gen)
    xr = x.clone().requires_grad_()
    ref = xr * torch.tanh(F.softplus(xr))
    (ref * go).sum().backward()
    xg = x.clone().cuda().requires_grad_()
    out = mg.blocks.Mish()(xg)
    (out * go.cuda()).sum().backward()
    assert_close(out.detach().cpu(), ref.detach(), 2e-6, "mish")
    assert_close(xg.grad.cpu(), xr.grad, 2e-6, "mish grad")
