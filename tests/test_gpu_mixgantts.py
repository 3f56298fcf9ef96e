"""(a16) MixGANTTS.forward on the HIP path against a run of the REAL reference (tests/golden/mixgantts_*.npz, made by
tests/golden/make_golden.py with the reference's own LinguisticEncoder on CPU): the encoder's nine recorded outputs
are replayed through a stand-in module, and every one of the 16 slots + p_targets + coarse_mels is compared -- value,
None-ness and requires_grad (model/mixgantts.py:55-183, `_detach` :182) -- for naive / shallow / aux, training and
inference, single and multi speaker; in training also the gradients reaching the encoder output and the decoder /
PostNet / denoiser weights (shallow: slot 15 is NOT detached, so postnet_loss trains the decoder)."""
import numpy as np
import pytest
import torch
from torch import nn

from helpers import (golden, T, assert_close, assert_digest, hot_path_configs, write_stats, MIXGANTTS_CASES,
                     mixgantts_case_name, mixgantts_encoder_outputs, mixgantts_leaves, assert_mixgantts_slots,
                     mixgantts_tapes, Tape)
from oracle import weights as WR

pytestmark = pytest.mark.gpu


class ReplayEncoder(nn.Module):
    """Stands in for model.linguistic_encoder.LinguisticEncoder (upstream of the path): returns what the reference's
    encoder returned for this batch."""

    def __init__(self, outputs):
        super().__init__()
        self.outputs = outputs

    def forward(self, *args, **kwargs):
        return self.outputs


class DropReplay:
    """transformer.DROPOUT_FN stand-in: the reference's keep-masks in call order; the decoder's were recorded
    [B, L, C] (we drop channel-major), PostNet's are [B, C, L] already."""

    def __init__(self, masks):
        self.masks, self.i = [T(m) for m in masks], 0

    def __call__(self, shape, p, device):
        m = self.masks[self.i]
        self.i += 1
        if tuple(m.shape) != tuple(shape):
            m = m.transpose(1, 2).contiguous()
        assert tuple(m.shape) == tuple(shape), (m.shape, shape)
        return m.to(device)


@pytest.mark.parametrize("model,ms,train", MIXGANTTS_CASES)
def test_mixgantts_forward_matches_reference(manifest, tmp_path, model, ms, train):
    import mixgan_tts_amd as mg
    g = golden(mixgantts_case_name(model, ms, train))
    stats = write_stats(tmp_path, np.linspace(-11.5, -9.0, 80), np.linspace(1.0, 2.0, 80), n_speakers=5)
    enc = mixgantts_encoder_outputs(g, train, "cuda")
    m = mg.MixGANTTS(*hot_path_configs(model, 4, multi_speaker=bool(ms), stats_dir=stats), linguistic_encoder=ReplayEncoder(enc))
    # weights: the fixture recipe over the reference's state_dict manifest (keys must exist with the same shapes)
    man = manifest["mixgantts_%s_ms%d" % (model, ms)]
    sd = m.state_dict()
    missing = [k for k in man["state_dict_order"] if k not in sd]
    extra = [k for k in sd if k not in man["state_dict"]]
    assert not missing and not extra, (missing, extra)
    w = WR.draw(man["seeded"], 61 + ms)
    np.testing.assert_allclose(WR.checksum(w), g["wsum"], rtol=1e-12)
    for k, a in w.items():
        assert tuple(sd[k].shape) == a.shape, k
        sd[k] = torch.from_numpy(a)
    m.load_state_dict(sd)
    m = m.cuda()
    rng, masks = mixgantts_tapes(g)
    drop = DropReplay(masks)
    mg.transformer.DROPOUT_FN = drop
    dev = lambda k: T(g[k]).cuda()  # noqa: E731
    try:
        if train:
            m.train()
            if model == "aux":
                m.diffusion.noise_fn = Tape(rng)
            else:
                m.diffusion.t_fn, m.diffusion.noise_fn = Tape(rng[:1]), Tape(rng[1:])
            mels = dev("mels")
            out, p_t, coarse = m(dev("speakers"), dev("texts"), dev("src_lens"), int(g["src_lens"].max()), dev("wb"),
                                 dev("src_w_lens"), 3, None, None, mels, dev("mel_lens"), mels.shape[1], dev("pitch"),
                                 dev("energy"), dev("dur"))
        else:
            m.eval()
            m.diffusion.noise_fn = Tape(rng)
            with torch.no_grad():
                out, p_t, coarse = m(dev("speakers"), dev("texts"), dev("src_lens"), int(g["src_lens"].max()), dev("wb"),
                                     dev("src_w_lens"), 3, d_control=4.0)
    finally:
        mg.transformer.DROPOUT_FN = None
    assert drop.i == len(masks)
    assert m.diffusion.noise_fn.i == len(m.diffusion.noise_fn.items)
    leaves = mixgantts_leaves(out, p_t, coarse)
    assert_mixgantts_slots(leaves, g, 5e-5, check_flags=train)
    if not train:
        return
    total = 0
    for k in sorted(k for k in g if k.startswith("w/")):
        total = total + (leaves[k[2:]] * dev(k)).sum()
    total.backward()
    assert_close(enc[0].grad, g["d_enc_out"], 1e-4, "d_enc_out")
    params = dict(m.named_parameters())
    for k in [k[len("has_grad/"):] for k in g if k.startswith("has_grad/")]:
        assert (params[k].grad is not None) == bool(g["has_grad/" + k]), k
        if params[k].grad is None:
            continue
        if k.endswith("w_ks.bias") or (k.startswith("postnet.") and k.endswith("conv.bias")):
            # exactly zero in theory (softmax is invariant to a shift of every key; batch-statistics BatchNorm
            # removes a conv bias): rounding noise on both sides, compared by magnitude only
            assert params[k].grad.abs().sum().item() < 1e-2 and g["dw_sum/" + k][1] < 1e-2, k
            continue
        assert_digest(params[k].grad, g, k, 2e-4)
    bufs = dict(m.postnet.named_buffers()) if model != "naive" else {}
    for k in [k[len("pn_buf/"):] for k in g if k.startswith("pn_buf/")]:
        assert_close(bufs[k].float(), g["pn_buf/" + k].astype(np.float32), 1e-5, k)
