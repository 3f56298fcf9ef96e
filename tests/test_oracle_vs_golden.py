"""Pins oracle/ (the CPU restatement) to outputs of the REAL reference (tests/golden/*.npz,
made by tests/golden/make_golden.py from /root/reference).  CPU only; tolerance 2e-6 relative for
fp32 forwards (same ATen ops, different association in a few places), 2e-5 for gradients.
"""
import numpy as np
import pytest
import torch

from helpers import (golden, T, seeded, assert_close, assert_digest, MIXGANTTS_CASES, mixgantts_case_name,
                     mixgantts_encoder_outputs, mixgantts_leaves, assert_mixgantts_slots, mixgantts_tapes)
from oracle import schedule as S, refmath as R

FWD = 2e-6
BWD = 2e-5


def _buf(mode="vpsde", Tn=4, g=None):
    b = {k: T(v) for k, v in S.diffusion_buffers(S.beta_schedule(mode, Tn, 0.1, 40, 0.008)).items()}
    if g is not None:
        b["spec_min"] = T(g["spec_min"])[None, None, :]
        b["spec_max"] = T(g["spec_max"])[None, None, :]
    return b


def test_schedule_buffers_match_reference():
    g = golden("schedule")
    for mode, Tn in [("vpsde", 1), ("vpsde", 4), ("vpsde", 100), ("vpsde", 1000), ("linear", 4), ("cosine", 4)]:
        with np.errstate(all="ignore"):
            b = S.diffusion_buffers(S.beta_schedule(mode, Tn, 0.1, 40, 0.008))
        for k in S.BUFFER_NAMES:
            ref = g["%s_%d/%s" % (mode, Tn, k)]
            np.testing.assert_array_equal(b[k], ref, err_msg="%s T=%d %s" % (mode, Tn, k))  # bit-exact, NaN==NaN


def test_schedule_known_values():
    # SURVEY.md section 8 a2 probe values for T=4 vpsde
    b = S.diffusion_buffers(S.beta_schedule("vpsde", 4, 0.1, 40, 0.008))
    np.testing.assert_allclose(b["alphas_cumprod"], [0.2803, 6.49e-3, 1.24e-5, 1.96e-9], rtol=2e-3)
    assert abs(b["posterior_log_variance_clipped"][0] - (-46.0517)) < 1e-3


def test_step_embedding():
    g = golden("step_embedding")
    emb = R.step_embedding(T(g["t"]))
    assert_close(emb[:3], g["emb"][:3], 1e-6, "step_embedding, t <= 3 (the T=4 configs)")
    # t = 99, 999: within the 1-ulp-of-a-frequency ambiguity of the reference's own host exp (see R.step_embedding)
    assert_close(emb, g["emb"], 1e-5, "step_embedding")


def test_elementwise():
    g = golden("elementwise")
    b = _buf(g=g)
    mel, noise, x0, xt = T(g["mel"]), T(g["noise"]), T(g["x0"]), T(g["xt"])
    assert_close(R.norm_spec(mel, b["spec_min"], b["spec_max"]), g["norm"], 1e-6, "norm")
    assert_close(R.denorm_spec(x0[:, 0].transpose(1, 2), b["spec_min"], b["spec_max"]), g["denorm"], 1e-6, "denorm")
    assert_close(R.q_sample(b, x0, T(g["tq"]), noise), g["q_sample"], 1e-6, "q_sample")
    assert_close(R.diffuse_fn(b, mel, T(g["td"]), noise), g["diffuse"], 1e-6, "diffuse_fn")
    assert_close(R.q_posterior_sample(b, x0, xt, T(g["tp"]), T(g["post_noise"])), g["post"], 1e-6, "posterior")


@pytest.mark.parametrize("ms", [0, 1])
def test_resblock(manifest, ms):
    name = "resblock_ms%d" % ms
    g = golden(name)
    W, ck = seeded(manifest, name, 11 + ms, requires_grad=True)
    np.testing.assert_allclose(ck, g["wsum"], rtol=1e-12)
    x, cond, step = (T(g[k]).requires_grad_() for k in ("x", "cond", "step"))
    spk = T(g["spk"]).requires_grad_() if ms else None
    nx, sk = R.resblock_forward(W, "", x, cond, step, spk)
    assert_close(nx, g["out_x"], FWD, "x")
    assert_close(sk, g["out_skip"], FWD, "skip")
    ((nx * T(g["gx"])).sum() + (sk * T(g["gs"])).sum()).backward()
    assert_close(x.grad, g["d_x"], BWD, "d_x")
    assert_close(cond.grad, g["d_cond"], BWD, "d_cond")
    assert_close(step.grad, g["d_step"], BWD, "d_step")
    if ms:
        assert_close(spk.grad, g["d_spk"], BWD, "d_spk")
    for k, w in W.items():
        assert_digest(w.grad, g, k, BWD)


@pytest.mark.parametrize("ms", [0, 1])
def test_denoiser(manifest, ms):
    name = "denoiser_ms%d" % ms
    g = golden(name)
    W, ck = seeded(manifest, name, 21 + ms, requires_grad=True)
    np.testing.assert_allclose(ck, g["wsum"], rtol=1e-12)
    x, cond = T(g["x"]).requires_grad_(), T(g["cond"]).requires_grad_()
    spk = T(g["spk"]).requires_grad_() if ms else None
    out = R.denoiser_forward(W, "", x, T(g["t"]), cond, spk)
    assert_close(out, g["out"], FWD, "denoiser out")
    # the running-sum skip reduction is the same function up to fp32 association
    out2 = R.denoiser_forward(W, "", x, T(g["t"]), cond, spk, stack_skips=False)
    assert_close(out2, g["out"], 1e-5, "denoiser out (running skip sum)")
    (out * T(g["go"])).sum().backward()
    assert_close(x.grad, g["d_x"], BWD, "d_x")
    assert_close(cond.grad, g["d_cond"], BWD, "d_cond")
    if ms:
        assert_close(spk.grad, g["d_spk"], BWD, "d_spk")
    for k, w in W.items():
        assert_digest(w.grad, g, k, 5e-5)


@pytest.mark.parametrize("model,ms", [("naive", 0), ("naive", 1), ("shallow", 0)])
def test_gaussian_diffusion(manifest, model, ms):
    name = "diffusion_%s_ms%d" % (model, ms)
    g = golden(name)
    W, ck = seeded(manifest, name, 31 + ms, requires_grad=True)
    np.testing.assert_allclose(ck, g["wsum"], rtol=1e-12)
    e = golden("elementwise")
    b = _buf(g=e)
    mel, pad = T(g["mel"]), T(g["pad"])
    cond = T(g["cond"]).requires_grad_()
    spk = T(g["spk"]) if ms else None
    coarse = T(g["coarse"]) if model == "shallow" else None
    tape = R.NoiseTape([T(g["t"]), T(g["n_xt"]), T(g["n_prev"]), T(g["n_post"])])
    x0p, x_t, x_prev, x_pp, t = R.diffusion_forward(W, b, model, 4, mel, cond, spk, pad, coarse, tape)
    assert_close(x_t, g["x_t"], 1e-6, "x_t")
    assert_close(x_prev, g["x_prev"], 1e-6, "x_prev")
    assert_close(x0p, g["x0_pred"], FWD, "x0_pred")
    assert_close(x_pp, g["x_prev_pred"], FWD, "x_prev_pred")
    ((x0p * T(g["w1"])).sum() + (x_pp * T(g["w2"])).sum()).backward()
    assert_close(cond.grad, g["d_cond"], BWD, "d_cond")
    for k in [k[len("dw_sum/"):] for k in g if k.startswith("dw_sum/")]:
        assert_digest(W[k].grad, g, k, 5e-5)
    # inference branch
    n_inf = sorted(k for k in g if k.startswith("infer_noise"))
    tape = R.NoiseTape([T(g[k]) for k in n_inf])
    Wd = {k: v.detach() for k, v in W.items()}
    y, *_ = R.diffusion_forward(Wd, b, model, 4, None, cond.detach(), spk, pad, coarse, tape)
    assert_close(y, g["infer_out"], 1e-5, "inference mel")
    # sampling() list (T+1 entries)
    n_s = sorted(k for k in g if k.startswith("sampling_noise"))
    tape = R.NoiseTape([T(g[k]) for k in n_s])
    ys = R.sampling(Wd, b, cond.detach().transpose(1, 2), spk, 4, tape)
    assert_close(torch.stack(ys), g["sampling_list"], 1e-5, "sampling list")
    if model == "shallow":
        tape = R.NoiseTape([T(g[k]) for k in sorted(k for k in g if k.startswith("trace_noise"))])
        tr = R.diffuse_trace(b, coarse, pad, 4, tape)
        assert_close(torch.stack(tr), g["trace"], 1e-6, "diffuse_trace")


@pytest.mark.parametrize("ms", [0, 1])
@pytest.mark.parametrize("L", [37, 64])
def test_jcu_and_losses(manifest, ms, L):
    g = golden("jcu_ms%d_L%d" % (ms, L))
    W, ck = seeded(manifest, "jcu_ms%d" % ms, 41 + ms, requires_grad=True)
    np.testing.assert_allclose(ck, g["wsum"], rtol=1e-12)
    x_ts, fake = T(g["x_ts"]).requires_grad_(), T(g["fake"]).requires_grad_()
    real, t = T(g["real"]), T(g["t"])
    s = T(g["s"]) if ms else None
    fc, fu = R.jcu_forward(W, x_ts, fake, s, t)
    rc, ru = R.jcu_forward(W, x_ts, real, s, t)
    for i in range(5):
        assert_close(fc[i], g["fc%d" % i], 5e-6, "fc%d" % i)
        assert_close(fu[i], g["fu%d" % i], 5e-6, "fu%d" % i)
        assert_close(rc[i], g["rc%d" % i], 5e-6, "rc%d" % i)
        assert_close(ru[i], g["ru%d" % i], 5e-6, "ru%d" % i)
    r_loss, f_loss = R.d_loss(rc[-1], ru[-1], fc[-1], fu[-1])
    adv = R.g_loss(fc[-1], fu[-1])
    fm = R.fm_loss(rc, ru, fc, fu)
    for a, k in ((r_loss, "r_loss"), (f_loss, "f_loss"), (adv, "adv"), (fm, "fm")):
        assert abs(a.item() - float(g[k])) <= 5e-6 * max(1.0, abs(float(g[k]))), k
    (r_loss + f_loss + adv + 10.0 * fm).backward()
    assert_close(x_ts.grad, g["d_x_ts"], BWD, "d_x_ts")
    assert_close(fake.grad, g["d_fake"], BWD, "d_fake")
    for k, w in W.items():
        assert_digest(w.grad, g, k, 5e-5)


def test_mel_l1():
    g = golden("mel_loss")
    v = R.mel_l1(T(g["pred"]), T(g["targ"]), T(g["pad"]))
    assert abs(v.item() - float(g["loss"])) < 1e-6


def test_fftblock(manifest):
    g = golden("fftblock")
    W, ck = seeded(manifest, "fftblock", 51, requires_grad=True)
    np.testing.assert_allclose(ck, g["wsum"], rtol=1e-12)
    x = T(g["x"]).requires_grad_()
    y = R.fft_block(W, "", x, T(g["pad"]))
    assert_close(y, g["out"], 5e-6, "fft out")
    (y * T(g["go"])).sum().backward()
    assert_close(x.grad, g["d_x"], BWD, "fft d_x")
    for k, w in W.items():
        assert_digest(w.grad, g, k, 5e-5)


def test_decoder_and_postnet(manifest):
    g = golden("decoder")
    W, ck = seeded(manifest, "decoder", 52)
    np.testing.assert_allclose(ck, g["wsum"], rtol=1e-12)
    msl = int(g["max_seq_len"])
    W["position_enc"] = R.sinusoid_table(msl + 1, 256)[None]
    for tag in ("short", "long"):
        y = R.decoder_forward(W, "", T(g[tag + "_x"]), T(g[tag + "_pad"]), msl)
        assert_close(y, g[tag + "_out"], 1e-5, "decoder " + tag)
    g = golden("postnet")
    W, ck = seeded(manifest, "postnet", 53)
    np.testing.assert_allclose(ck, g["wsum"], rtol=1e-12)
    assert_close(R.postnet_forward(W, "", T(g["x"])), g["out"], 5e-6, "postnet")


def test_lingops_index_functions():
    """word_level_pooling, LengthRegulator, get_mapping_mask, get_rel_coef."""
    g = golden("lingops")
    src = T(g["src_seq"]).requires_grad_()
    src_len, wb, swl = T(g["src_len"]), T(g["wb"]), T(g["src_w_len"])
    for red in ("sum", "mean"):
        o = R.word_level_pooling(src, src_len, wb, swl, red)
        assert torch.equal(o.detach(), T(g["pool_" + red])), red
        src.grad = None
        (o * T(g["pool_%s_gw" % red])).sum().backward()
        assert_close(src.grad, g["pool_%s_dsrc" % red], 1e-6, "pool grad " + red)
    xw, dur = T(g["xw"]).requires_grad_(), T(g["dur_w"])
    for tag, ml in (("auto", None), ("max20", 20), ("crop12", 12)):
        o, lens = R.length_regulate(xw, dur, ml)
        assert torch.equal(o.detach(), T(g["lr_" + tag])) and torch.equal(lens, T(g["lr_%s_len" % tag])), tag
        xw.grad = None
        (o * T(g["lr_%s_gw" % tag])).sum().backward()
        assert_close(xw.grad, g["lr_%s_dx" % tag], 1e-6, "LR grad " + tag)
    mm = R.mapping_mask(g["mapping_mask"].shape[1], g["mapping_mask"].shape[2], dur, wb, swl)
    assert torch.equal(mm, T(g["mapping_mask"]))
    assert torch.equal(R.rel_coef(dur, swl, T(g["mel_mask"])), T(g["rel_coef_q"]))
    assert torch.equal(R.rel_coef(wb, swl, T(g["src_mask"])), T(g["rel_coef_kv"]))


def test_hifigan_generator(manifest):
    """hifigan/models.py:112-173 against a reference run (weight-norm gains from
    the fixture, all other parameters from the seeded recipe)."""
    g = golden("hifigan")
    W, ck = seeded(manifest, "hifigan", 81)
    np.testing.assert_allclose(ck, g["wsum"], rtol=1e-12)
    for k in g:
        if k.startswith("gain/"):
            W[k[5:]] = T(g[k])
    y = R.hifigan_forward(W, T(g["mel"]))
    assert tuple(y.shape) == (2, 1, 13 * 256)
    assert_close(y, g["wav"], 1e-5, "hifigan")


class _MaskTape:
    """drop(shape, p) stand-in replaying the reference's dropout keep-masks in call order."""

    def __init__(self, g, prefix):
        self.masks = [T(g[k]) for k in sorted((k for k in g if k.startswith(prefix + "/mask")),
                                              key=lambda k: int(k.rsplit("mask", 1)[1]))]
        self.i = 0

    def __call__(self, shape, p):
        m = self.masks[self.i]
        self.i += 1
        assert tuple(m.shape) == tuple(shape)
        return m


def test_aux_train_mode_blocks(manifest):
    """train-mode FFTBlock / Decoder / PostNet of the oracle (dropout masks replayed,
    BatchNorm batch statistics) against a reference run, forward and backward."""
    g = golden("aux_train")
    # FFTBlock
    W, _ = seeded(manifest, "fftblock", 51, requires_grad=True)
    x = T(g["fft/x"]).requires_grad_()
    tape = _MaskTape(g, "fft")
    y = R.fft_block(W, "", x, T(g["fft/pad"]), drop=tape)
    assert tape.i == len(tape.masks) == 2
    assert_close(y, g["fft/out"], 2e-6, "fft train out")
    (y * T(g["fft/go"])).sum().backward()
    assert_close(x.grad, g["fft/d_x"], 2e-5, "fft train d_x")
    for k in W:
        assert_digest(W[k].grad, {kk[4:]: v for kk, v in g.items() if kk.startswith("fft/dw")}, k, 2e-5)
    # Decoder (input longer than max_seq_len: clipped in train mode)
    W, _ = seeded(manifest, "decoder", 52, requires_grad=True)
    msl = int(g["dec/max_seq_len"])
    W["position_enc"] = R.sinusoid_table(msl + 1, 256)[None]
    x = T(g["dec/x"]).requires_grad_()
    tape = _MaskTape(g, "dec")
    y = R.decoder_forward(W, "", x, T(g["dec/pad"]), msl, training=True, drop=tape)
    assert tape.i == 12 and tuple(y.shape) == (2, msl, 256)
    assert_close(y, g["dec/out"], 1e-5, "decoder train out")
    (y * T(g["dec/go"])).sum().backward()
    assert_close(x.grad, g["dec/d_x"], 5e-5, "decoder train d_x")
    # PostNet
    W, _ = seeded(manifest, "postnet", 53, requires_grad=True)
    x = T(g["pn/x"]).requires_grad_()
    tape = _MaskTape(g, "pn")
    y = R.postnet_forward(W, "", x, training=True, drop=tape)
    assert tape.i == 5
    assert_close(y, g["pn/out"], 5e-6, "postnet train out")
    (y * T(g["pn/go"])).sum().backward()
    assert_close(x.grad, g["pn/d_x"], 5e-5, "postnet train d_x")
    for k in W:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert_close(W[k], g["pn/buf/" + k], 1e-6, k)
        elif not k.endswith("num_batches_tracked"):
            assert_digest(W[k].grad, {kk[3:]: v for kk, v in g.items() if kk.startswith("pn/dw")}, k, 5e-5)


@pytest.mark.parametrize("model,ms,train", MIXGANTTS_CASES)
def test_mixgantts_forward(manifest, model, ms, train):
    """(a16) model/mixgantts.py:55-183 downstream of the (recorded) linguistic encoder: every output slot, its
    None-ness and requires_grad, and in training the gradients reaching the encoder output and a few weights."""
    name = mixgantts_case_name(model, ms, train)
    g = golden(name)
    W, ck = seeded(manifest, "mixgantts_%s_ms%d" % (model, ms), 61 + ms, requires_grad=train)
    np.testing.assert_allclose(ck, g["wsum"], rtol=1e-12)
    W["decoder.position_enc"] = R.sinusoid_table(1001, 256)[None]          # max_seq_len 1000 (config/LJSpeech/model.yaml)
    b = _buf(g=golden("elementwise"))
    enc = mixgantts_encoder_outputs(g, train)
    src_lens, src_w_lens = T(g["src_lens"]), T(g["src_w_lens"])
    valid = lambda lens: torch.arange(int(lens.max()))[None, :] < lens[:, None]  # noqa: E731
    spk = W["speaker_emb.weight"][T(g["speakers"])] if ms else None
    rng, masks = mixgantts_tapes(g)
    tape = R.NoiseTape([T(a) for a in rng])
    mt = _MaskTape({"m/mask%d" % i: a for i, a in enumerate(masks)}, "m")
    with torch.set_grad_enabled(train):
        out, p_t, coarse = R.mixgantts_forward(
            W, b, model, 4, enc, valid(src_lens), valid(src_w_lens), src_lens, spk, T(g["mels"]) if train else None,
            train, tape, mt if train else None, p_targets=T(g["pitch"]) if train else None)
    assert tape.i == len(rng) and mt.i == len(masks)
    leaves = mixgantts_leaves(out, p_t, coarse)
    assert_mixgantts_slots(leaves, g, 2e-5, check_flags=train)
    if not train:
        return
    total = 0
    for k in sorted(k for k in g if k.startswith("w/")):
        total = total + (leaves[k[2:]] * T(g[k])).sum()
    total.backward()
    assert_close(enc[0].grad, g["d_enc_out"], 5e-5, "d_enc_out")
    for k in [k[len("has_grad/"):] for k in g if k.startswith("has_grad/")]:
        assert (W[k].grad is not None) == bool(g["has_grad/" + k]), k
        if W[k].grad is not None:
            assert_digest(W[k].grad, g, k, 1e-4)
    for k in [k[len("pn_buf/"):] for k in g if k.startswith("pn_buf/") and not k.endswith("num_batches_tracked")]:
        assert_close(W["postnet." + k], g["pn_buf/" + k], 1e-6, k)


def test_scheduled_optim_matches_reference():
    """model/optimizer.py:5-56: lr sequence across warm-up and two anneal boundaries, and the weights after 12
    Adam steps (CPU)."""
    import mixgan_tts_amd as mg
    g = golden("aux_train")
    tcfg = {"optimizer_fs2": {"betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0, "warm_up_step": 5,
                              "anneal_steps": [8, 11], "anneal_rate": 0.3}}
    lin = torch.nn.Linear(3, 2)
    with torch.no_grad():
        lin.weight.copy_(T(g["so/w0"]))
        lin.bias.copy_(T(g["so/b0"]))
    so = mg.ScheduledOptim(lin, tcfg, {"transformer": {"encoder_hidden": 256}}, 2)
    assert so.init_lr == float(g["so/init_lr"]) == so.get_last_lr()
    lrs = []
    for _ in range(12):
        so.zero_grad()
        lin(T(g["so/x"])).pow(2).sum().backward()
        lrs.append(so.step())
    np.testing.assert_allclose(np.array(lrs), g["so/lrs"], rtol=1e-15)
    assert so.get_last_lr() == lrs[-1]
    assert_close(lin.weight, g["so/w_end"], 1e-6, "ScheduledOptim weights")
    assert_close(lin.bias, g["so/b_end"], 1e-6, "ScheduledOptim bias")
