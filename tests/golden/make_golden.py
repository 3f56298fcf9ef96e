#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on CPU.

Run in the build container only:   python tests/golden/make_golden.py
The reference source never leaves this container; only inputs/outputs (data) are written.
Weights are not stored: they are re-drawn on both sides by oracle/weights.py from the
state_dict manifest kept in tests/golden/manifest.json.

Cases follow SURVEY.md section 8c (i)-(ix).
"""
import contextlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_harness as H  # noqa: E402

H.install_stubs()
import torch  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)

from model.diffusion import GaussianDiffusion  # noqa: E402
from model.modules import Denoiser  # noqa: E402
from model.blocks import ResidualBlock, DiffusionEmbedding  # noqa: E402
from model.mixgantts import JCUDiscriminator  # noqa: E402
from model import loss as ref_loss  # noqa: E402
from transformer import Decoder, PostNet  # noqa: E402
from transformer.Layers import FFTBlock  # noqa: E402

from oracle import weights as WR  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
MANIFEST = {}


def manifest_of(mod):
    """Trainable parameters + BatchNorm running stats, i.e. everything the recipe draws."""
    m = {}
    for k, p in mod.named_parameters():
        if p.requires_grad:
            m[k] = list(p.shape)
    for k, b in mod.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            m[k] = list(b.shape)
    return m


def seed_module(mod, seed, name, exclude=None):
    man = manifest_of(mod)
    if exclude is not None:          # e.g. the injected linguistic encoder of MixGANTTS: its outputs are recorded instead
        man = {k: v for k, v in man.items() if not k.startswith(exclude)}
    keep = lambda k: exclude is None or not k.startswith(exclude)  # noqa: E731
    MANIFEST[name] = {"seeded": man,
                      "state_dict": {k: list(v.shape) for k, v in mod.state_dict().items() if keep(k)},
                      "state_dict_order": [k for k in mod.state_dict().keys() if keep(k)],
                      "param_order": [k for k, _ in mod.named_parameters() if keep(k)]}
    w = WR.draw(man, seed)
    sd = mod.state_dict()
    with torch.no_grad():
        for k, a in w.items():
            sd[k].copy_(torch.from_numpy(a))
    return WR.checksum(w)


class Tape:
    """Deterministic stand-in for torch.randn / randn_like / randint inside the reference."""

    def __init__(self, rng, T):
        self.rng = rng
        self.T = T
        self.log = []
        self.forced_t = None

    def randn(self, *shape, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        a = torch.from_numpy(self.rng.standard_normal(shape).astype(np.float32))
        self.log.append(a.numpy().copy())
        return a

    def randn_like(self, x, **kw):
        return self.randn(tuple(x.shape))

    def randint(self, lo, hi, shape, **kw):
        if self.forced_t is not None:
            a = torch.as_tensor(self.forced_t, dtype=torch.long).clone()
        else:
            a = torch.from_numpy(self.rng.integers(lo, hi, size=tuple(shape)).astype(np.int64))
        self.log.append(a.numpy().copy())
        return a


@contextlib.contextmanager
def patched_rng(tape):
    saved = (torch.randn, torch.randn_like, torch.randint)
    torch.randn, torch.randn_like, torch.randint = tape.randn, tape.randn_like, tape.randint
    try:
        yield tape
    finally:
        torch.randn, torch.randn_like, torch.randint = saved


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    clean = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        clean[k] = np.asarray(v)
    np.savez_compressed(path, **clean)
    sz = os.path.getsize(path)
    print("  wrote %-40s %8.1f KB" % (os.path.basename(path), sz / 1024))


def configs(model="naive", T=4, mode="vpsde", multi_speaker=False, stats_dir=None, max_seq_len=None):
    pre, mc, tr = H.load_configs("LJSpeech")
    mc["denoiser"]["noise_schedule_naive"] = mode
    mc["denoiser"]["timesteps"] = T
    mc["denoiser"]["shallow_timesteps"] = T
    mc["multi_speaker"] = multi_speaker
    if max_seq_len is not None:
        mc["max_seq_len"] = max_seq_len
    pre["path"]["preprocessed_path"] = stats_dir
    return types.SimpleNamespace(model=model), pre, mc, tr


def grad_digest(g):
    """Small summary of a big gradient: fp64 sum, abs-sum and a fixed corner slice."""
    g = g.detach().double()
    corner = g[tuple(slice(0, min(4, s)) for s in g.shape)].float().numpy()
    return np.array([g.sum().item(), g.abs().sum().item()]), corner


# --------------------------------------------------------------------------------------------
def mixgantts_cases(stats, M):
    """(a16) MixGANTTS.forward (model/mixgantts.py:55-183) with the REAL LinguisticEncoder on CPU.  The encoder is
    upstream of the path: its nine outputs are recorded so the product test replays them through a stand-in encoder;
    every one of the 16 output slots + p_targets + coarse_mels is stored with its None-ness and requires_grad flag,
    and for the training cases the gradients of a seeded linear functional of the differentiable slots w.r.t. the
    encoder output and a few weights.  Own RNG streams: adding this case leaves every other fixture unchanged."""
    import torch.nn.functional as _F
    from model.mixgantts import MixGANTTS
    rng = np.random.default_rng(20241101)

    class DropTape:
        def __init__(self):
            self.masks = []

        def __call__(self, x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            keep = torch.from_numpy((rng.random(tuple(x.shape)) >= p).astype(np.float32))
            self.masks.append(keep.numpy().astype(np.uint8))
            return x * keep / (1.0 - p)

    def flat_slots(out, p_targets, coarse):
        """name -> tensor|None for every leaf of the return value."""
        d = {}
        for i, o in enumerate(out):
            if isinstance(o, (list, tuple)):
                for j, oo in enumerate(o):
                    d["slot%02d/%d" % (i, j)] = oo
            else:
                d["slot%02d" % i] = o
        d["p_targets"], d["coarse_mels"] = p_targets, coarse
        return d

    B = 2
    wb = torch.tensor([[2, 1, 3], [4, 2, 0]])
    src_w_lens = torch.tensor([3, 2])
    src_lens = wb.sum(1)
    Tp = int(src_lens.max())
    for model, ms, train in (("naive", False, True), ("naive", True, True), ("naive", False, False),
                             ("shallow", False, True), ("shallow", False, False), ("aux", False, True)):
        name = "mixgantts_%s_ms%d_%s" % (model, int(ms), "train" if train else "infer")
        print(" ", name)
        torch.manual_seed(977)                         # the encoder keeps its own (torch-default) initialisation
        T = 4
        a, pre, mc, tr = configs(model, T, multi_speaker=ms, stats_dir=stats)
        pre["preprocessing"]["speaker_embedder"] = "none"      # config/AISHELL3/preprocess.yaml (absent from LJSpeech's)
        m = MixGANTTS(a, pre, mc, tr)
        ck = seed_module(m, 61 + int(ms), "mixgantts_%s_ms%d" % (model, int(ms)), exclude="linguistic_encoder.")
        texts = torch.from_numpy(rng.integers(1, 50, (B, Tp)))
        dur = torch.from_numpy(rng.integers(1, 6, (B, Tp)))
        for b in range(B):
            texts[b, src_lens[b]:] = 0
            dur[b, src_lens[b]:] = 0
        speakers = torch.tensor([1, 3])
        arrs = dict(wsum=ck, texts=texts, src_lens=src_lens, wb=wb, src_w_lens=src_w_lens, speakers=speakers)
        enc_rec = {}

        def hook(mod, inp, out_):
            out_[0].retain_grad() if out_[0].requires_grad else None
            enc_rec["out"] = out_
        h = m.linguistic_encoder.register_forward_hook(hook)
        tape = Tape(rng, T)
        drop = DropTape()
        saved_dropout = _F.dropout
        if train:
            mel_lens = dur.sum(1)
            Lm = int(mel_lens.max())
            mels = torch.from_numpy(rng.uniform(-11.5, 2.0, (B, Lm, M)).astype(np.float32))
            for b in range(B):
                mels[b, mel_lens[b]:] = 0
            pitch = torch.from_numpy(rng.standard_normal((B, Tp)).astype(np.float32))
            energy = torch.from_numpy(rng.standard_normal((B, Tp)).astype(np.float32))
            attn = torch.from_numpy(rng.uniform(0, 1, (B, Tp, Lm)).astype(np.float32))
            arrs.update(mels=mels, mel_lens=mel_lens, pitch=pitch, energy=energy, dur=dur)
            m.train()
            m.linguistic_encoder.eval()                # its dropout is not replayed (its outputs are recorded)
            _F.dropout = drop
            try:
                with patched_rng(tape):
                    out, p_targets, coarse = m(speakers, texts, src_lens, Tp, wb, src_w_lens, 3, None, attn, mels, mel_lens,
                                               Lm, pitch, energy, dur)
            finally:
                _F.dropout = saved_dropout
        else:
            m.eval()
            with patched_rng(tape), torch.no_grad():
                out, p_targets, coarse = m(speakers, texts, src_lens, Tp, wb, src_w_lens, 3, d_control=4.0)
        h.remove()
        eo = enc_rec["out"]
        assert int(eo[5].min()) > 0, "an utterance came out with zero frames: change d_control"
        for i, o in enumerate(eo):
            if isinstance(o, (list, tuple)):
                for j, oo in enumerate(o):
                    arrs["enc/%d/%d" % (i, j)] = oo
            elif o is not None:
                arrs["enc/%d" % i] = o
        slots = flat_slots(out, p_targets, coarse)
        flags = {}
        for k, v in slots.items():
            flags[k] = -1 if v is None else int(bool(v.requires_grad))
            if v is not None:
                arrs[k] = v
        arrs["flag_names"] = np.array(sorted(flags))
        arrs["flags"] = np.array([flags[k] for k in sorted(flags)], dtype=np.int64)
        for i, a_ in enumerate(tape.log):
            arrs["rng%d" % i] = a_
        for i, mk in enumerate(drop.masks):
            arrs["mask%d" % i] = mk
        if train:
            # a seeded linear functional of every differentiable slot that is ours to produce
            total = 0
            for k in ("slot00", "slot01/2", "slot15"):
                v = slots.get(k)
                if k == "slot00" and model == "aux":
                    for j, tr_ in enumerate(out[0]):
                        w = torch.from_numpy(rng.standard_normal(tuple(tr_.shape)).astype(np.float32))
                        arrs["w/slot00/%d" % j] = w
                        total = total + (tr_ * w).sum()
                    continue
                if v is not None and v.requires_grad:
                    w = torch.from_numpy(rng.standard_normal(tuple(v.shape)).astype(np.float32))
                    arrs["w/" + k] = w
                    total = total + (v * w).sum()
            total.backward()
            arrs["d_enc_out"] = eo[0].grad
            picks = ("diffusion.denoise_fn.residual_layers.7.", "diffusion.denoise_fn.output_projection",
                     "decoder.layer_stack.0.", "decoder.layer_stack.5.pos_ffn", "mel_linear", "postnet.convolutions.0.",
                     "postnet.convolutions.4.", "speaker_emb")
            for k, p in m.named_parameters():
                if k.startswith(picks):
                    arrs["has_grad/" + k] = np.array(int(p.grad is not None))
                    if p.grad is not None:
                        dg, corner = grad_digest(p.grad)
                        arrs["dw_sum/" + k] = dg
                        arrs["dw_corner/" + k] = corner
            if model != "naive":
                for k, b_ in m.postnet.named_buffers():
                    arrs["pn_buf/" + k] = b_.detach().numpy().copy()
        save(name, **arrs)


def main():
    if "--only-mixgantts" in sys.argv:                 # add the (a16) case without re-running the others
        with open(os.path.join(OUT, "manifest.json")) as f:
            MANIFEST.update(json.load(f))
        M = 80
        stats = H.make_stats_dir(np.linspace(-11.5, -9.0, M), np.linspace(1.0, 2.0, M), n_speakers=5)
        mixgantts_cases(stats, M)
        with open(os.path.join(OUT, "manifest.json"), "w") as f:
            json.dump(MANIFEST, f, indent=0, sort_keys=True)
        print("manifest written")
        return
    rng = np.random.default_rng(20240607)
    M = 80
    spec_min = np.linspace(-11.5, -9.0, M)
    spec_max = np.linspace(1.0, 2.0, M)
    stats = H.make_stats_dir(spec_min, spec_max, n_speakers=5)

    # (i) schedules ---------------------------------------------------------------------------
    print("schedules")
    sched = {}
    for mode, T in [("vpsde", 1), ("vpsde", 4), ("vpsde", 100), ("vpsde", 1000), ("linear", 4), ("cosine", 4)]:
        gd = GaussianDiffusion(*configs("naive", T, mode, stats_dir=stats))
        for k, v in gd.named_buffers():
            if "denoise_fn" in k:
                continue
            sched["%s_%d/%s" % (mode, T, k)] = v.numpy().copy()
    save("schedule", **sched)

    # (ii) step embedding ---------------------------------------------------------------------
    t = torch.tensor([0, 1, 3, 99, 999], dtype=torch.long)
    save("step_embedding", t=t, emb=DiffusionEmbedding(256)(t))

    # (iii) elementwise diffusion algebra ------------------------------------------------------
    print("elementwise")
    gd = GaussianDiffusion(*configs("naive", 4, stats_dir=stats))
    B, L = 3, 37
    mel = torch.from_numpy(rng.uniform(-11.5, 2.0, (B, L, M)).astype(np.float32))
    noise = torch.from_numpy(rng.standard_normal((B, 1, M, L)).astype(np.float32))
    x0 = torch.from_numpy(rng.uniform(-1, 1, (B, 1, M, L)).astype(np.float32))
    xt = torch.from_numpy(rng.standard_normal((B, 1, M, L)).astype(np.float32))
    tq = torch.tensor([3, 0, 2], dtype=torch.long)
    td = torch.tensor([1, -1, 3], dtype=torch.long)
    tp = torch.tensor([0, 2, 3], dtype=torch.long)
    tape = Tape(rng, 4)
    with patched_rng(tape):
        post = gd.q_posterior_sample(x0, xt, tp)
    save("elementwise", mel=mel, noise=noise, x0=x0, xt=xt, tq=tq, td=td, tp=tp,
         norm=gd.norm_spec(mel), denorm=gd.denorm_spec(x0[:, 0].transpose(1, 2)),
         q_sample=gd.q_sample(x0, tq, noise), diffuse=gd.diffuse_fn(mel, td.clone(), noise),
         post_noise=tape.log[0], post=post, spec_min=spec_min.astype(np.float32), spec_max=spec_max.astype(np.float32))

    # (iv) one residual block -----------------------------------------------------------------
    print("resblock")
    for ms in (False, True):
        blk = ResidualBlock(256, 256, dropout=0.2, multi_speaker=ms)
        name = "resblock_ms%d" % int(ms)
        ck = seed_module(blk, 11 + int(ms), name)
        B, L = 2, 37
        x = torch.from_numpy(rng.standard_normal((B, 256, L)).astype(np.float32)).requires_grad_()
        cond = torch.from_numpy(rng.standard_normal((B, 256, L)).astype(np.float32)).requires_grad_()
        step = torch.from_numpy(rng.standard_normal((B, 256)).astype(np.float32)).requires_grad_()
        spk = torch.from_numpy(rng.standard_normal((B, 256)).astype(np.float32)).requires_grad_() if ms else None
        gx = torch.from_numpy(rng.standard_normal((B, 256, L)).astype(np.float32))
        gs = torch.from_numpy(rng.standard_normal((B, 256, L)).astype(np.float32))
        nx, sk = blk(x, cond, step, spk)
        ((nx * gx).sum() + (sk * gs).sum()).backward()
        arrs = dict(x=x, cond=cond, step=step, gx=gx, gs=gs, out_x=nx, out_skip=sk,
                    d_x=x.grad, d_cond=cond.grad, d_step=step.grad, wsum=ck)
        if ms:
            arrs.update(spk=spk, d_spk=spk.grad)
        for k, p in blk.named_parameters():
            dg, corner = grad_digest(p.grad)
            arrs["dw_sum/" + k] = dg
            arrs["dw_corner/" + k] = corner
        save(name, **arrs)

    # (v) full denoiser -----------------------------------------------------------------------
    print("denoiser")
    for ms in (False, True):
        _, pre, mc, _ = configs("naive", 4, multi_speaker=ms, stats_dir=stats)
        den = Denoiser(pre, mc)
        name = "denoiser_ms%d" % int(ms)
        ck = seed_module(den, 21 + int(ms), name)
        B, L = 2, 64
        x = torch.from_numpy(rng.standard_normal((B, 1, M, L)).astype(np.float32)).requires_grad_()
        cond = torch.from_numpy(rng.standard_normal((B, 256, L)).astype(np.float32)).requires_grad_()
        spk = torch.from_numpy(rng.standard_normal((B, 256)).astype(np.float32)).requires_grad_() if ms else None
        t = torch.tensor([3, 0], dtype=torch.long)
        go = torch.from_numpy(rng.standard_normal((B, 1, M, L)).astype(np.float32))
        out = den(x, t, cond, spk)
        (out * go).sum().backward()
        arrs = dict(x=x, cond=cond, t=t, go=go, out=out, d_x=x.grad, d_cond=cond.grad, wsum=ck)
        if ms:
            arrs.update(spk=spk, d_spk=spk.grad)
        for k, p in den.named_parameters():
            dg, corner = grad_digest(p.grad)
            arrs["dw_sum/" + k] = dg
            if k.startswith(("residual_layers.0.", "residual_layers.19.", "mlp.", "input_projection", "skip_projection",
                             "output_projection")):
                arrs["dw_corner/" + k] = corner
        save(name, **arrs)

    # (vi) GaussianDiffusion.forward / sampling --------------------------------------------------
    print("diffusion forward")
    for model in ("naive", "shallow"):
        for ms in (False, True):
            if model == "shallow" and ms:
                continue
            T = 4
            gd = GaussianDiffusion(*configs(model, T, multi_speaker=ms, stats_dir=stats))
            name = "diffusion_%s_ms%d" % (model, int(ms))
            ck = seed_module(gd, 31 + int(ms), name)
            B, L = 3, 48
            lens = torch.tensor([48, 33, 40])
            pad = torch.arange(L)[None, :] >= lens[:, None]          # True = pad, as MixGANTTS passes it
            mel = torch.from_numpy(rng.uniform(-11.5, 2.0, (B, L, M)).astype(np.float32))
            mel = mel.masked_fill(pad.unsqueeze(-1), 0.0)
            cond = torch.from_numpy(rng.standard_normal((B, L, 256)).astype(np.float32)).requires_grad_()
            spk = torch.from_numpy(rng.standard_normal((B, 256)).astype(np.float32)) if ms else None
            coarse = torch.from_numpy(rng.uniform(-11.0, 1.5, (B, L, M)).astype(np.float32)) if model == "shallow" else None
            tape = Tape(rng, T)
            tape.forced_t = [2, 0, 3]
            gd.train()
            with patched_rng(tape):
                x0p, x_t, x_prev, x_prev_pred, t = gd(mel, cond, spk, pad, coarse)
            w1 = torch.from_numpy(rng.standard_normal(tuple(x0p.shape)).astype(np.float32))
            w2 = torch.from_numpy(rng.standard_normal(tuple(x0p.shape)).astype(np.float32))
            ((x0p * w1).sum() + (x_prev_pred * w2).sum()).backward()
            arrs = dict(mel=mel, cond=cond, pad=pad, t=t, n_xt=tape.log[1], n_prev=tape.log[2], n_post=tape.log[3],
                        x0_pred=x0p, x_t=x_t, x_prev=x_prev, x_prev_pred=x_prev_pred, w1=w1, w2=w2,
                        d_cond=cond.grad, wsum=ck)
            for k, p in gd.named_parameters():
                if p.grad is not None and k.startswith(("denoise_fn.residual_layers.7.", "denoise_fn.output_projection")):
                    dg, corner = grad_digest(p.grad)
                    arrs["dw_sum/" + k] = dg
                    arrs["dw_corner/" + k] = corner
            if ms:
                arrs["spk"] = spk
            if coarse is not None:
                arrs["coarse"] = coarse
            # inference branch (sampling): uses the cond/spk stashed by the call above too
            gd.eval()
            tape2 = Tape(rng, T)
            with patched_rng(tape2), torch.no_grad():
                y, *_ = gd(None, cond.detach(), spk, pad, coarse)
            arrs["infer_out"] = y
            for i, a in enumerate(tape2.log):
                arrs["infer_noise%d" % i] = a
            # sampling() with no args from the stash, full list
            tape3 = Tape(rng, T)
            with patched_rng(tape3), torch.no_grad():
                ys = gd.sampling()
            arrs["sampling_list"] = torch.stack(ys)
            for i, a in enumerate(tape3.log):
                arrs["sampling_noise%d" % i] = a
            if model == "shallow":
                tape4 = Tape(rng, T)
                with patched_rng(tape4), torch.no_grad():
                    tr = gd.diffuse_trace(coarse, pad)
                arrs["trace"] = torch.stack(tr)
                for i, a in enumerate(tape4.log):
                    arrs["trace_noise%d" % i] = a
            save(name, **arrs)

    # (vii) JCU discriminator + (viii) losses ----------------------------------------------------
    print("jcu")
    for ms in (False, True):
        for L in (37, 64):
            _, pre, mc, tr = configs("naive", 4, multi_speaker=ms, stats_dir=stats)
            D = JCUDiscriminator(pre, mc, tr)
            name = "jcu_ms%d_L%d" % (int(ms), L)
            ck = seed_module(D, 41 + int(ms), "jcu_ms%d" % int(ms))
            B = 2
            x_ts = torch.from_numpy(rng.standard_normal((B, L, M)).astype(np.float32)).requires_grad_()
            fake = torch.from_numpy(rng.standard_normal((B, L, M)).astype(np.float32)).requires_grad_()
            real = torch.from_numpy(rng.standard_normal((B, L, M)).astype(np.float32))
            s = torch.from_numpy(rng.standard_normal((B, 256)).astype(np.float32)) if ms else None
            t = torch.tensor([1, 3], dtype=torch.long)
            fc, fu = D(x_ts, fake, s, t)
            rc, ru = D(x_ts, real, s, t)
            d_fn, g_fn = ref_loss.get_adversarial_losses_fn("lsgan")
            r_loss, f_loss = d_fn(rc[-1], ru[-1], fc[-1], fu[-1])
            adv = g_fn(fc[-1], fu[-1])
            holder = types.SimpleNamespace(n_layers=mc["discriminator"]["n_layer"] + mc["discriminator"]["n_cond_layer"])
            fm = ref_loss.MixGANTTSLoss.get_fm_loss(holder, rc, ru, fc, fu)
            total = r_loss + f_loss + adv + 10.0 * fm
            total.backward()
            arrs = dict(x_ts=x_ts, fake=fake, real=real, t=t, r_loss=r_loss, f_loss=f_loss, adv=adv, fm=fm,
                        d_x_ts=x_ts.grad, d_fake=fake.grad, wsum=ck)
            if ms:
                arrs["s"] = s
            for i in range(5):
                arrs["fc%d" % i] = fc[i]
                arrs["fu%d" % i] = fu[i]
                arrs["rc%d" % i] = rc[i]
                arrs["ru%d" % i] = ru[i]
            for k, p in D.named_parameters():
                dg, corner = grad_digest(p.grad)
                arrs["dw_sum/" + k] = dg
                arrs["dw_corner/" + k] = corner
            save(name, **arrs)

    # mel L1 (model/loss.py:229-242)
    print("mel loss")
    B, L = 3, 21
    lens = torch.tensor([21, 13, 17])
    pad = torch.arange(L)[None, :] >= lens[:, None]
    pred = torch.from_numpy(rng.standard_normal((B, L, M)).astype(np.float32))
    targ = torch.from_numpy(rng.standard_normal((B, L, M)).astype(np.float32))
    holder = types.SimpleNamespace(mel_masks_fill=pad)
    holder.l1_loss = types.MethodType(ref_loss.MixGANTTSLoss.l1_loss, holder)
    holder.weights_nonzero_speech = types.MethodType(ref_loss.MixGANTTSLoss.weights_nonzero_speech, holder)
    ml = ref_loss.MixGANTTSLoss.get_mel_loss(holder, pred, targ)
    save("mel_loss", pred=pred, targ=targ, pad=pad, loss=ml)

    # (ix) FFT block / Decoder / PostNet ---------------------------------------------------------
    print("fft")
    blk = FFTBlock(256, 2, 128, 128, 1024, 9, dropout=0.2).eval()
    ck = seed_module(blk, 51, "fftblock")
    B, L = 2, 40
    lens = torch.tensor([40, 29])
    pad = torch.arange(L)[None, :] >= lens[:, None]
    x = torch.from_numpy(rng.standard_normal((B, L, 256)).astype(np.float32)).requires_grad_()
    go = torch.from_numpy(rng.standard_normal((B, L, 256)).astype(np.float32))
    y, _ = blk(x, mask=pad, slf_attn_mask=pad.unsqueeze(1).expand(-1, L, -1))
    (y * go).sum().backward()
    arrs = dict(x=x, pad=pad, go=go, out=y, d_x=x.grad, wsum=ck)
    for k, p in blk.named_parameters():
        dg, corner = grad_digest(p.grad)
        arrs["dw_sum/" + k] = dg
        arrs["dw_corner/" + k] = corner
    save("fftblock", **arrs)

    _, pre, mc, _ = configs("shallow", 4, stats_dir=stats, max_seq_len=48)
    dec = Decoder(mc).eval()
    ck = seed_module(dec, 52, "decoder")
    arrs = dict(wsum=ck, max_seq_len=48)
    for tag, L, lens in (("short", 40, [40, 31]), ("long", 60, [60, 47])):
        lens = torch.tensor(lens)
        pad = torch.arange(L)[None, :] >= lens[:, None]
        x = torch.from_numpy(rng.standard_normal((2, L, 256)).astype(np.float32))
        with torch.no_grad():
            y = dec(x, pad)
        arrs.update({tag + "_x": x, tag + "_pad": pad, tag + "_out": y})
    save("decoder", **arrs)

    pn = PostNet().eval()
    ck = seed_module(pn, 53, "postnet")
    x = torch.from_numpy(rng.standard_normal((2, 40, M)).astype(np.float32))
    with torch.no_grad():
        y = pn(x)
    save("postnet", x=x, out=y, wsum=ck)

    # (f1) host-loop index ops of the linguistic encoder ---------------------------------------------
    print("lingops")
    from model.linguistic_encoder import LinguisticEncoder, LengthRegulator
    from utils.tools import word_level_pooling, get_mask_from_lengths
    B, Hd = 3, 256
    src_w_len = torch.tensor([5, 3, 4])
    wb = torch.tensor([[2, 1, 3, 1, 2], [4, 2, 1, 0, 0], [1, 1, 5, 2, 0]])          # phones per word
    src_len = wb.sum(1)
    Tp = int(src_len.max())
    src_seq = torch.from_numpy(rng.standard_normal((B, Tp, Hd)).astype(np.float32)).requires_grad_()
    arrs = dict(src_seq=src_seq, src_len=src_len, wb=wb, src_w_len=src_w_len)
    for red in ("sum", "mean"):
        o = word_level_pooling(src_seq, src_len, wb, src_w_len, reduce=red)
        gw = torch.from_numpy(rng.standard_normal(tuple(o.shape)).astype(np.float32))
        src_seq.grad = None
        (o * gw).sum().backward()
        arrs.update({"pool_" + red: o, "pool_" + red + "_gw": gw, "pool_" + red + "_dsrc": src_seq.grad.clone()})
    dur_w = torch.tensor([[3, 0, 7, 2, 4], [6, 1, 2, 0, 0], [2, 5, 1, 3, 0]])       # frames per word
    xw = torch.from_numpy(rng.standard_normal((B, 5, Hd)).astype(np.float32)).requires_grad_()
    lr = LengthRegulator()
    for tag, ml in (("auto", None), ("max20", 20), ("crop12", 12)):
        o, ml_out = lr(xw, dur_w, ml)
        gw = torch.from_numpy(rng.standard_normal(tuple(o.shape)).astype(np.float32))
        xw.grad = None
        (o * gw).sum().backward()
        arrs.update({"lr_%s" % tag: o, "lr_%s_len" % tag: ml_out, "lr_%s_gw" % tag: gw, "lr_%s_dx" % tag: xw.grad.clone()})
    arrs.update(dur_w=dur_w, xw=xw)
    mel_len = dur_w.sum(1)
    Lq = int(mel_len.max())
    q = torch.zeros(B, Lq, 4)
    kv = torch.zeros(B, Tp, 4)
    arrs["mapping_mask"] = LinguisticEncoder.get_mapping_mask(None, q, kv, dur_w, wb, src_w_len)
    mel_mask = get_mask_from_lengths(mel_len)                  # True = valid (utils/tools.py:144-153)
    src_mask = get_mask_from_lengths(src_len)
    arrs["rel_coef_q"] = LinguisticEncoder.get_rel_coef(None, dur_w, src_w_len, mel_mask)
    arrs["rel_coef_kv"] = LinguisticEncoder.get_rel_coef(None, wb, src_w_len, src_mask)
    arrs.update(mel_mask=mel_mask, src_mask=src_mask)
    save("lingops", **arrs)

    # (f3) HiFi-GAN V1 generator (vocoder) -----------------------------------------------------------
    print("hifigan")
    import json as _json
    from hifigan.models import Generator as HifiGenerator
    with open(os.path.join(H.REF, "hifigan", "config.json")) as f:
        hcfg = _json.load(f)
    hobj = types.SimpleNamespace(**hcfg)
    voc = HifiGenerator(hobj).eval()
    ck = seed_module(voc, 81, "hifigan")
    with torch.no_grad():   # keep the weight-norm gains near 1 so the 40-conv chain stays O(1)
        for k_, p_ in voc.named_parameters():
            if k_.endswith("weight_g"):
                p_.copy_(1.0 + 0.1 * torch.from_numpy(rng.standard_normal(tuple(p_.shape)).astype(np.float32)))
    gains = {k_: p_.detach().numpy().copy() for k_, p_ in voc.named_parameters() if k_.endswith("weight_g")}
    melv = torch.from_numpy(rng.uniform(-11.5, 2.0, (2, 80, 13)).astype(np.float32))
    with torch.no_grad():
        wav = voc(melv)
    arrs = dict(mel=melv, wav=wav, wsum=ck)
    arrs.update({"gain/" + k_: v_ for k_, v_ in gains.items()})
    save("hifigan", **arrs)

    # (f2) data path: dataset.py Dataset / TextDataset collation + utils/tools.py to_device ---------------
    print("dataset")
    import tempfile as _tf
    from dataset import Dataset as RefDataset, TextDataset as RefTextDataset
    from text import text_to_sequence as ref_t2s
    from utils.tools import to_device as ref_to_device
    drng = np.random.default_rng(424242)
    ddir = _tf.mkdtemp(prefix="mg_data_")
    spks = ["spkA", "spkB", "spkC"]
    with open(os.path.join(ddir, "speakers.json"), "w") as f:
        json.dump({s_: i_ for i_, s_ in enumerate(spks)}, f)
    with open(os.path.join(ddir, "stats.json"), "w") as f:
        json.dump({"pitch": [-2.0, 8.0, 0.0, 1.0], "energy": [-1.5, 7.0, 0.0, 1.0]}, f)
    for k_ in ("mel", "pitch", "energy", "duration", "phones_per_word", "attn_prior", "spker_embed"):
        os.makedirs(os.path.join(ddir, k_))
    phs = "HH AH0 L OW1 sp W ER1 D AA1 R K S IY1 T N Z spn M EY1 B".split()
    arrs, lines = {}, []
    for s_ in spks:
        e_ = drng.standard_normal((1, 256)).astype(np.float32)
        np.save(os.path.join(ddir, "spker_embed", "%s-spker_embed.npy" % s_), e_)
        arrs["spk/" + s_] = e_
    N_ITEMS = 11
    for i_ in range(N_ITEMS):
        base, s_ = "utt%02d" % i_, spks[i_ % 3]
        nw = int(drng.integers(2, 6))
        ppw = drng.integers(1, 4, nw).astype(np.int64)
        n_ph = int(ppw.sum())
        dur = drng.integers(1, 7, n_ph).astype(np.int64)
        Lm = int(dur.sum())
        item = {"mel": drng.uniform(-11.5, 2.0, (Lm, 80)).astype(np.float32),
                "pitch": drng.standard_normal(Lm).astype(np.float64),       # np.save of a python-float list is float64
                "energy": drng.standard_normal(Lm).astype(np.float32),
                "duration": dur, "phones_per_word": ppw,
                "attn_prior": drng.uniform(0, 1, (n_ph, Lm)).astype(np.float32)}
        for k_, a_ in item.items():
            np.save(os.path.join(ddir, k_, "%s-%s-%s.npy" % (s_, k_, base)), a_)
            arrs["item%02d/%s" % (i_, k_)] = a_
        text = "{" + " ".join(phs[int(j_)] for j_ in drng.integers(0, len(phs), n_ph)) + "}"
        lines.append("%s|%s|%s|raw text %d" % (base, s_, text, i_))
        arrs["item%02d/phone_ids" % i_] = np.array(ref_t2s(text, ["english_cleaners"]), dtype=np.int64)
    with open(os.path.join(ddir, "train.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")
    arrs["meta_lines"] = np.array(lines)

    def _dump(prefix, batchs):
        for b_, tup in enumerate(batchs):
            for j_, x_ in enumerate(tup):
                key = "%s/b%d/s%02d" % (prefix, b_, j_)
                if x_ is None:
                    continue
                if torch.is_tensor(x_):
                    arrs[key] = x_.numpy()
                    arrs[key + "_torch_dtype"] = np.array(str(x_.dtype))
                elif isinstance(x_, list):
                    arrs[key] = np.array(x_)
                else:
                    arrs[key] = np.asarray(x_)
        arrs[prefix + "/n_batches"] = np.array(len(batchs))

    dpre = {"dataset": "Synth", "path": {"preprocessed_path": ddir},
            "preprocessing": {"text": {"text_cleaners": ["english_cleaners"]}, "speaker_embedder": "none"}}
    dtrain = {"optimizer": {"batch_size": 4, "batch_size_shallow": 3}}
    # (1) train.py:31-33 configuration: sort=True, drop_last=True, naive (batch 4): 11 items -> 2 sub-batches
    ds = RefDataset("train.txt", types.SimpleNamespace(model="naive"), dpre, {"multi_speaker": False}, dtrain,
                    sort=True, drop_last=True)
    _dump("sorted_drop", ds.collate_fn([ds[i_] for i_ in range(N_ITEMS)]))
    # (2) evaluate.py configuration: sort=False, drop_last=False, shallow (batch 3), speaker embeddings loaded
    dpre2 = json.loads(json.dumps(dpre))
    dpre2["preprocessing"]["speaker_embedder"] = "DeepSpeaker"
    ds2 = RefDataset("train.txt", types.SimpleNamespace(model="shallow"), dpre2, {"multi_speaker": True}, dtrain,
                     sort=False, drop_last=False)
    order2 = [7, 2, 9, 0, 5, 10, 3]
    b2 = ds2.collate_fn([ds2[i_] for i_ in order2])
    _dump("plain_keep", b2)
    arrs["plain_keep/order"] = np.array(order2)
    _dump("plain_keep_dev", [ref_to_device(t_, torch.device("cpu")) for t_ in b2])
    # (3) TextDataset (synthesize.py:258)
    tds = RefTextDataset(os.path.join(ddir, "train.txt"), dpre2, {"multi_speaker": True})
    order3 = [4, 1, 8]
    b3 = tds.collate_fn([tds[i_] for i_ in order3])
    _dump("text", [b3])
    arrs["text/order"] = np.array(order3)
    _dump("text_dev", [ref_to_device(b3, torch.device("cpu"))])
    save("dataset", **arrs)

    # (f4) aux pre-training: train-mode FFTBlock / Decoder / PostNet (dropout masks taped) + ScheduledOptim -----
    print("aux_train")
    import torch.nn.functional as _F
    from model.optimizer import ScheduledOptim as RefScheduledOptim
    arng = np.random.default_rng(777001)

    class DropTape:
        def __init__(self):
            self.masks = []

        def __call__(self, x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            keep = torch.from_numpy((arng.random(tuple(x.shape)) >= p).astype(np.float32))
            self.masks.append(keep.numpy().astype(np.uint8))
            return x * keep / (1.0 - p)

    def taped(fn):
        tape, saved = DropTape(), _F.dropout
        _F.dropout = tape
        try:
            out = fn()
        finally:
            _F.dropout = saved
        return out, tape.masks

    def draw(shape):
        return torch.from_numpy(arng.standard_normal(shape).astype(np.float32))

    arrs = {}
    # FFTBlock, train mode
    blk = FFTBlock(256, 2, 128, 128, 1024, 9, dropout=0.2).train()
    seed_module(blk, 51, "fftblock")
    B, L = 2, 37
    pad = torch.arange(L)[None, :] >= torch.tensor([37, 22])[:, None]
    x = draw((B, L, 256)).requires_grad_()
    go = draw((B, L, 256))
    (y, _), masks = taped(lambda: blk(x, mask=pad, slf_attn_mask=pad.unsqueeze(1).expand(-1, L, -1)))
    (y * go).sum().backward()
    arrs.update({"fft/x": x, "fft/pad": pad, "fft/go": go, "fft/out": y, "fft/d_x": x.grad})
    for i_, m_ in enumerate(masks):
        arrs["fft/mask%d" % i_] = m_
    for k, p in blk.named_parameters():
        dg, corner = grad_digest(p.grad)
        arrs["fft/dw_sum/" + k] = dg
        arrs["fft/dw_corner/" + k] = corner
    # Decoder, train mode, input longer than max_seq_len (Models.py:153-162 clips it)
    _, pre, mc, _ = configs("shallow", 4, stats_dir=stats, max_seq_len=48)
    dec = Decoder(mc).train()
    seed_module(dec, 52, "decoder")
    Ld = 53
    padd = torch.arange(Ld)[None, :] >= torch.tensor([53, 40])[:, None]
    xd = draw((2, Ld, 256)).requires_grad_()
    yd, masks = taped(lambda: dec(xd, padd))
    god = draw(tuple(yd.shape))
    (yd * god).sum().backward()
    arrs.update({"dec/x": xd, "dec/pad": padd, "dec/go": god, "dec/out": yd, "dec/d_x": xd.grad,
                 "dec/max_seq_len": 48})
    for i_, m_ in enumerate(masks):
        arrs["dec/mask%d" % i_] = m_
    for k, p in dec.named_parameters():
        if p.grad is not None:
            arrs["dec/dw_sum/" + k] = grad_digest(p.grad)[0]
    # PostNet, train mode (BatchNorm batch statistics, dropout 0.5)
    pn = PostNet().train()
    seed_module(pn, 53, "postnet")
    xp = draw((3, 45, M)).requires_grad_()
    yp, masks = taped(lambda: pn(xp))
    gop = draw(tuple(yp.shape))
    (yp * gop).sum().backward()
    arrs.update({"pn/x": xp, "pn/go": gop, "pn/out": yp, "pn/d_x": xp.grad})
    for i_, m_ in enumerate(masks):
        arrs["pn/mask%d" % i_] = m_
    for k, p in pn.named_parameters():
        dg, corner = grad_digest(p.grad)
        arrs["pn/dw_sum/" + k] = dg
        arrs["pn/dw_corner/" + k] = corner
    for k, b_ in pn.named_buffers():
        arrs["pn/buf/" + k] = b_.detach().numpy().copy()
    # ScheduledOptim (model/optimizer.py:5-56): lr after each of the first steps and across the anneal boundary
    tcfg = {"optimizer_fs2": {"betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0, "warm_up_step": 5,
                              "anneal_steps": [8, 11], "anneal_rate": 0.3}}
    lin = torch.nn.Linear(3, 2)
    with torch.no_grad():
        lin.weight.copy_(draw((2, 3)))
        lin.bias.copy_(draw((2,)))
    arrs["so/w0"], arrs["so/b0"] = lin.weight.detach().numpy().copy(), lin.bias.detach().numpy().copy()
    so = RefScheduledOptim(lin, tcfg, {"transformer": {"encoder_hidden": 256}}, 2)
    xin = draw((4, 3))
    arrs["so/x"] = xin
    lrs = []
    for _ in range(12):
        so.zero_grad()
        lin(xin).pow(2).sum().backward()
        lrs.append(so.step())
    arrs["so/lrs"] = np.array(lrs, dtype=np.float64)
    arrs["so/w_end"], arrs["so/b_end"] = lin.weight.detach().numpy().copy(), lin.bias.detach().numpy().copy()
    arrs["so/init_lr"] = np.array(so.init_lr)
    save("aux_train", **arrs)

    print("mixgantts")
    mixgantts_cases(stats, M)

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(MANIFEST, f, indent=0, sort_keys=True)
    print("manifest written")


if __name__ == "__main__":
    main()
