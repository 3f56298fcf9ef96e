"""GPU parity of JCUDiscriminator forward/backward (model/mixgantts.py:186-288) against the
reference fixtures: all 10 feature maps for fake and real pairs, the LSGAN / feature-matching
scalars computed from them, and every gradient (inputs and all parameters)."""
import numpy as np
import pytest
import torch

from helpers import golden, T, assert_close, assert_digest, hot_path_configs, load_seeded
from oracle import refmath as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd as m
    assert torch.cuda.is_available()
    m.lib()
    return m


def dev(a):
    return T(a).cuda()


@pytest.mark.parametrize("ms", [0, 1])
@pytest.mark.parametrize("L", [37, 64])
def test_jcu_forward_backward_golden(mg, manifest, ms, L):
    g = golden("jcu_ms%d_L%d" % (ms, L))
    _, pre, mc, tr = hot_path_configs(multi_speaker=bool(ms), stats_dir=".")
    D = mg.JCUDiscriminator(pre, mc, tr)
    ck = load_seeded(D, manifest, "jcu_ms%d" % ms, 41 + ms)
    np.testing.assert_allclose(ck, g["wsum"], rtol=1e-12)
    D = D.cuda()
    x_ts, fake = dev(g["x_ts"]).requires_grad_(), dev(g["fake"]).requires_grad_()
    real, t = dev(g["real"]), dev(g["t"])
    s = dev(g["s"]) if ms else None
    fc, fu = D(x_ts, fake, s, t)
    rc, ru = D(x_ts, real, s, t)
    for i in range(5):
        for name, v in (("fc", fc), ("fu", fu), ("rc", rc), ("ru", ru)):
            assert_close(v[i].detach().cpu(), g["%s%d" % (name, i)], 2e-5, "%s%d" % (name, i))
    # losses of model/loss.py:12-30,221-227 evaluated on our feature maps
    r_loss, f_loss = R.d_loss(rc[-1], ru[-1], fc[-1], fu[-1])
    adv = R.g_loss(fc[-1], fu[-1])
    fm = R.fm_loss(rc, ru, fc, fu)
    for a, k in ((r_loss, "r_loss"), (f_loss, "f_loss"), (adv, "adv"), (fm, "fm")):
        assert abs(a.item() - float(g[k])) <= 2e-5 * max(1.0, abs(float(g[k]))), k
    (r_loss + f_loss + adv + 10.0 * fm).backward()
    torch.cuda.synchronize()
    assert_close(x_ts.grad.cpu(), g["d_x_ts"], 5e-5, "d_x_ts")
    assert_close(fake.grad.cpu(), g["d_fake"], 5e-5, "d_fake")
    for k, p in D.named_parameters():
        assert p.grad is not None, k
        assert_digest(p.grad, g, k, 1e-4)


def test_jcu_full_size_shapes_and_finite(mg):
    """BASELINE-size smoke (B=16, L=1000): feature-map shapes of SURVEY.md section 3.4 and a finite backward."""
    _, pre, mc, tr = hot_path_configs(stats_dir=".")
    D = mg.JCUDiscriminator(pre, mc, tr).cuda()
    B, L = 16, 1000
    a = torch.randn(B, L, 80, device="cuda")
    b = torch.randn(B, L, 80, device="cuda", requires_grad=True)
    t = torch.randint(0, 4, (B,), device="cuda")
    c, u = D(a, b, None, t)
    assert [tuple(v.shape) for v in c] == [(B, 64, 1000), (B, 128, 500), (B, 512, 250), (B, 128, 250), (B, 1, 250)]
    assert [tuple(v.shape) for v in u] == [tuple(v.shape) for v in c]
    (c[-1].square().mean() + u[-1].square().mean()).backward()
    assert torch.isfinite(b.grad).all()
    for p in D.parameters():
        assert torch.isfinite(p.grad).all()
