"""GPU parity of the device-side linguistic-encoder index ops against the
reference fixtures (bit-exact: integer / byte work and in-order fp32 sums) and, at a batch the size of
BASELINE configs[1], against the oracle's loops."""
import pytest
import torch

from helpers import golden, T, assert_close
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


def test_lingops_golden_bit_exact(mg):
    g = golden("lingops")
    L = mg.lingops
    src = dev(g["src_seq"]).requires_grad_()
    src_len, wb, swl = dev(g["src_len"]), dev(g["wb"]), dev(g["src_w_len"])
    for red in ("sum", "mean"):
        o = L.word_level_pooling(src, src_len, wb, swl, reduce=red)
        assert torch.equal(o.detach().cpu(), T(g["pool_" + red])), red
        src.grad = None
        (o * dev(g["pool_%s_gw" % red])).sum().backward()
        assert torch.equal(src.grad.cpu(), T(g["pool_%s_dsrc" % red])), red
    xw, dur = dev(g["xw"]).requires_grad_(), dev(g["dur_w"])
    lr = L.LengthRegulator()
    for tag, ml in (("auto", None), ("max20", 20), ("crop12", 12)):
        o, lens = lr(xw, dur, ml)
        assert torch.equal(o.detach().cpu(), T(g["lr_" + tag])) and torch.equal(lens.cpu(), T(g["lr_%s_len" % tag])), tag
        xw.grad = None
        (o * dev(g["lr_%s_gw" % tag])).sum().backward()
        assert_close(xw.grad.cpu(), g["lr_%s_dx" % tag], 1e-6, "LR grad " + tag)
    mm = g["mapping_mask"]
    q = torch.zeros(mm.shape[0], mm.shape[1], 4, device="cuda")
    kv = torch.zeros(mm.shape[0], mm.shape[2], 4, device="cuda")
    assert torch.equal(L.get_mapping_mask(q, kv, dur, wb, swl).cpu(), T(mm))
    assert torch.equal(L.get_rel_coef(dur, swl, dev(g["mel_mask"])).cpu(), T(g["rel_coef_q"]))
    assert torch.equal(L.get_rel_coef(wb, swl, dev(g["src_mask"])).cpu(), T(g["rel_coef_kv"]))


def test_lingops_vs_oracle_cfg2_shape(mg):
    """B=16 utterances of ~1000 frames: ragged word / phoneme counts, zero and long durations."""
    gen = torch.Generator().manual_seed(17)
    B, Tw, H = 16, 64, 256
    swl = torch.randint(20, Tw + 1, (B,), generator=gen)
    wb = torch.randint(1, 5, (B, Tw), generator=gen)
    dur = torch.randint(0, 33, (B, Tw), generator=gen)
    for b in range(B):
        wb[b, swl[b]:] = 0
        dur[b, swl[b]:] = 0
    src_len = wb.sum(1)
    Tp = int(src_len.max())
    src = torch.randn(B, Tp, H, generator=gen)
    xw = torch.randn(B, Tw, H, generator=gen)
    L = mg.lingops
    for red in ("sum", "mean"):
        ref = R.word_level_pooling(src, src_len, wb, swl, red)
        got = L.word_level_pooling(src.cuda(), src_len.cuda(), wb.cuda(), swl.cuda(), reduce=red, max_words=int(swl.max()))
        assert torch.equal(got.cpu(), ref), red
    ref_o, ref_len = R.length_regulate(xw, dur, None)
    got_o, got_len = L.LengthRegulator()(xw.cuda(), dur.cuda(), None)
    assert torch.equal(got_o.cpu(), ref_o) and torch.equal(got_len.cpu(), ref_len)
    Lq = ref_o.shape[1]
    mm = R.mapping_mask(Lq, Tp, dur, wb, swl)
    got = L.get_mapping_mask(torch.zeros(B, Lq, 1, device="cuda"), torch.zeros(B, Tp, 1, device="cuda"), dur.cuda(),
                             wb.cuda(), swl.cuda())
    assert torch.equal(got.cpu(), mm)
    mel_mask = torch.arange(Lq)[None, :] < ref_len[:, None]
    assert torch.equal(L.get_rel_coef(dur.cuda(), swl.cuda(), mel_mask.cuda()).cpu(), R.rel_coef(dur, swl, mel_mask))
    src_mask = torch.arange(Tp)[None, :] < src_len[:, None]
    assert torch.equal(L.get_rel_coef(wb.cuda(), swl.cuda(), src_mask.cuda()).cpu(), R.rel_coef(wb, swl, src_mask))
