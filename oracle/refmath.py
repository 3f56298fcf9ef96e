"""PyTorch-CPU restatement of the reference hot path as pure functions over a flat weight dict.

Every function takes `W`, a `{reference state_dict key: tensor}` mapping (optionally with a
key prefix `p`), so the checkpoint key names of SURVEY.md section 5 are spelled out where
they are used.  Autograd works through all of it (used for the gradient fixtures).

Test infrastructure (see oracle/__init__.py); fp32 throughout like the reference.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- primitives
def step_embedding(t, dim=256):
    """model/blocks.py:906-913 -- [sin | cos] of t * exp(-i ln(1e4)/(dim/2-1))."""
    half = dim // 2
    rate = math.log(10000) / (half - 1)
    # The reference takes torch.exp of the fp32 products.  A host's vectorised fp32 exp is only good to ~1 ulp and
    # differs between CPU models (3 of the 128 entries on the fixture machine are not the correctly rounded value,
    # other entries on the GPU box's host), and one ulp of a frequency is 6e-5 rad at t = 999.  Restated with the
    # correctly rounded table (exp in fp64 of the same fp32 products), which is what every host approximates:
    # 3.4e-6 from the fixture at t = 999, identical on every machine.
    freq = torch.exp((torch.arange(half, device=t.device) * -rate).double()).float()
    ang = (t[:, None] * freq[None, :]).double()
    return torch.cat((ang.sin(), ang.cos()), dim=-1).float()


def mish(x):
    """model/blocks.py:894-896."""
    return x * torch.tanh(F.softplus(x))


def _conv(W, name, x, stride=1, padding=0):
    """ConvNorm (model/blocks.py:326-371): nn.Conv1d with bias; keys `<name>.conv.{weight,bias}`."""
    return F.conv1d(x, W[name + ".conv.weight"], W[name + ".conv.bias"], stride=stride, padding=padding)


def _lin(W, name, x):
    """LinearNorm (model/blocks.py:278-291): bias-free nn.Linear; key `<name>.linear.weight`."""
    return F.linear(x, W[name + ".linear.weight"])


def extract(a, t, ndim):
    """model/diffusion.py:26-29."""
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))


# ----------------------------------------------------------------------------- diffusion algebra
def norm_spec(x, spec_min, spec_max):
    """model/diffusion.py:228-229."""
    return (x - spec_min) / (spec_max - spec_min) * 2 - 1


def denorm_spec(x, spec_min, spec_max):
    """model/diffusion.py:231-232."""
    return (x + 1) / 2 * (spec_max - spec_min) + spec_min


def q_sample(buf, x_start, t, noise):
    """model/diffusion.py:147-153."""
    return (extract(buf["sqrt_alphas_cumprod"], t, x_start.dim()) * x_start
            + extract(buf["sqrt_one_minus_alphas_cumprod"], t, x_start.dim()) * noise)


def diffuse_fn(buf, mel, t, noise):
    """model/diffusion.py:177-185.  mel [B,L,M] -> [B,1,M,L]; rows with t<0 return the clean x0.

    (The reference mutates `t` in place; callers pass temporaries, so a clone is equivalent.)
    """
    x0 = norm_spec(mel, buf["spec_min"], buf["spec_max"]).transpose(1, 2)[:, None, :, :]
    neg = t < 0
    tt = t.clone()
    tt[neg] = 0
    out = q_sample(buf, x0, tt, noise)
    out[neg] = x0[neg]
    return out


def q_posterior_sample(buf, x_start, x_t, t, noise):
    """model/diffusion.py:104-119."""
    nd = x_t.dim()
    mean = (extract(buf["posterior_mean_coef1"], t, nd) * x_start
            + extract(buf["posterior_mean_coef2"], t, nd) * x_t)
    logvar = extract(buf["posterior_log_variance_clipped"], t, nd)
    gate = (1 - (t == 0).float()).reshape(x_start.shape[0], *((1,) * (nd - 1)))
    return mean + gate * (0.5 * logvar).exp() * noise


# ----------------------------------------------------------------------------- denoiser
def resblock_forward(W, p, x, cond, step, spk=None):
    """model/blocks.py:1157-1176.  Returns (x_next, skip)."""
    d = _lin(W, p + "diffusion_projection", step).unsqueeze(-1)
    c = _conv(W, p + "conditioner_projection", cond)
    resid = x + d
    h = resid + c
    if (p + "speaker_projection.linear.weight") in W:
        h = h + _lin(W, p + "speaker_projection", spk).unsqueeze(-1)
    z = _conv(W, p + "conv_layer", h, padding=1)
    g, f = torch.chunk(z, 2, dim=1)
    y = torch.sigmoid(g) * torch.tanh(f)
    o = _conv(W, p + "output_projection", y)
    nx, skip = torch.chunk(o, 2, dim=1)
    return (nx + resid) / math.sqrt(2.0), skip


def denoiser_n_layers(W, p=""):
    n = 0
    while (p + "residual_layers.%d.conv_layer.conv.weight" % n) in W:
        n += 1
    return n


def denoiser_forward(W, p, mel, t, cond, spk=None, stack_skips=True):
    """model/modules.py:420-446.  mel [B,1,M,L], t int64 [B], cond [B,H,L] -> [B,1,M,L].

    `stack_skips=True` reproduces the reference's `torch.stack(skip)` reduction (the op the
    CPU baseline must pay for).
    """
    x = F.relu(_conv(W, p + "input_projection.0", mel[:, 0]))
    x = F.relu(x)
    s = step_embedding(t, W[p + "mlp.0.linear.weight"].shape[1])
    s = _lin(W, p + "mlp.2", mish(_lin(W, p + "mlp.0", s)))
    n = denoiser_n_layers(W, p)
    skips = []
    for i in range(n):
        x, sk = resblock_forward(W, p + "residual_layers.%d." % i, x, cond, s, spk)
        skips.append(sk)
    if stack_skips:
        x = torch.sum(torch.stack(skips), dim=0) / math.sqrt(n)
    else:
        acc = skips[0]
        for sk in skips[1:]:
            acc = acc + sk
        x = acc / math.sqrt(n)
    x = F.relu(_conv(W, p + "skip_projection", x))
    x = _conv(W, p + "output_projection", x)
    return x[:, None, :, :]


# ----------------------------------------------------------------------------- GaussianDiffusion
class NoiseTape:
    """Replays pre-drawn tensors in call order (stands in for torch.randn*/randint)."""

    def __init__(self, items):
        self.items = list(items)
        self.i = 0

    def __call__(self, shape=None):
        x = self.items[self.i]
        self.i += 1
        if shape is not None:
            assert tuple(x.shape) == tuple(shape), (tuple(x.shape), tuple(shape))
        return x


def p_sample(W, buf, x_t, t, cond, spk, noise, clip=True):
    """model/diffusion.py:121-129."""
    with torch.no_grad():
        x0 = denoiser_forward(W, "denoise_fn.", x_t, t, cond, spk)
        if clip:
            x0 = x0.clamp(-1.0, 1.0)
        return q_posterior_sample(buf, x0, x_t, t, noise)


def sampling(W, buf, cond, spk, T, tape, noise=None, keep_all=True):
    """model/diffusion.py:155-165.  cond [B,H,L].  Returns the list of T+1 denormalised mels."""
    B, _, L = cond.shape
    M = buf["spec_min"].shape[-1]
    xs = [tape((B, 1, M, L)) if noise is None else noise]
    for i in reversed(range(T)):
        t = torch.full((B,), i, dtype=torch.long)
        xs.append(p_sample(W, buf, xs[-1], t, cond, spk, tape((B, 1, M, L))))
    outs = [denorm_spec(x[:, 0].transpose(1, 2), buf["spec_min"], buf["spec_max"]) for x in xs]
    return outs if keep_all else outs[-1:]


def diffusion_forward(W, buf, model, T, mel, cond, spk, mel_mask, coarse_mel, tape, t=None, clip=True):
    """model/diffusion.py:187-226.  `mel_mask` is True = pad on entry, as MixGANTTS passes it.

    RNG order: t, noise(x_t), noise(x_{t-1}), noise(posterior) in training;
    [noise(x_T)], then one noise per step at inference.  All taken from `tape` (t too unless given).
    """
    B = cond.shape[0]
    valid = ~mel_mask.unsqueeze(-1)                       # [B,L,1] True = keep
    condT = cond.transpose(1, 2)
    if mel is None:
        if model != "shallow":
            start = None
        else:
            tt = torch.full((B,), T - 1, dtype=torch.long)
            M = buf["spec_min"].shape[-1]
            start = diffuse_fn(buf, coarse_mel, tt, tape((B, 1, M, cond.shape[1]))) \
                * valid.unsqueeze(-1).transpose(1, -1)
        x0 = sampling(W, buf, condT, spk, T, tape, noise=start)[-1] * valid
        return x0, None, None, None, (None if model != "shallow" else tt)
    vm = valid.unsqueeze(-1).transpose(1, -1)             # [B,1,1,L]
    if t is None:
        t = tape((B,))
    shp = (B, 1, mel.shape[2], mel.shape[1])
    x_t = diffuse_fn(buf, mel, t, tape(shp)) * vm
    x_prev = diffuse_fn(buf, mel, t - 1, tape(shp)) * vm
    x0 = denoiser_forward(W, "denoise_fn.", x_t, t, condT, spk) * vm
    if clip:
        x0 = x0.clamp(-1.0, 1.0)
    if model != "shallow":
        start = x0
    else:
        start = norm_spec(coarse_mel, buf["spec_min"], buf["spec_max"]).transpose(1, 2)[:, None, :, :]
    x_prev_pred = q_posterior_sample(buf, start, x_t, t, tape(shp)) * vm
    back = lambda a: a[:, 0].transpose(1, 2)
    return back(x0), back(x_t), back(x_prev), back(x_prev_pred), t


def diffuse_trace(buf, x_start, mask, T, tape):
    """model/diffusion.py:167-175 (aux only).  mask True = pad."""
    B, L, M = x_start.shape
    keep = ~mask.unsqueeze(-1)
    out = [norm_spec(x_start, buf["spec_min"], buf["spec_max"]).clamp(-1.0, 1.0) * keep]
    for i in range(T):
        t = torch.full((B,), i, dtype=torch.long, device=x_start.device)
        out.append(diffuse_fn(buf, x_start, t, tape((B, 1, M, L)))[:, 0].transpose(1, 2) * keep)
    return out


# ----------------------------------------------------------------------------- JCU discriminator
JCU_KERNELS = (3, 5, 5, 5, 3)
JCU_STRIDES = (1, 2, 2, 1, 1)


def jcu_forward(W, x_ts, x_t_prevs, s, t, kernels=JCU_KERNELS, strides=JCU_STRIDES, n_layer=3):
    """model/mixgantts.py:256-288.  Returns (cond_feats[5], uncond_feats[5])."""
    x = _lin(W, "input_projection", torch.cat([x_t_prevs, x_ts], dim=-1)).transpose(1, 2)
    e = step_embedding(t, W["mlp.0.linear.weight"].shape[1])
    e = _lin(W, "mlp.2", mish(_lin(W, "mlp.0", e))).unsqueeze(-1)
    cond_feats, uncond_feats = [], []
    for i in range(n_layer):
        k = kernels[i]
        x = F.leaky_relu(_conv(W, "conv_block.%d" % i, x, strides[i], (k - 1) // 2), 0.2)
        cond_feats.append(x)
        uncond_feats.append(x)
    xc = x + e
    if "spk_mlp.0.linear.weight" in W:
        xc = xc + _lin(W, "spk_mlp.0", s).unsqueeze(-1)
    xu = x
    n_tail = len(kernels) - n_layer
    for j in range(n_tail):
        k = kernels[n_layer + j]
        st = strides[n_layer + j]
        xc = F.leaky_relu(_conv(W, "cond_conv_block.%d" % j, xc, st, (k - 1) // 2), 0.2)
        cond_feats.append(xc)
    for j in range(n_tail):
        k = kernels[n_layer + j]
        st = strides[n_layer + j]
        xu = F.leaky_relu(_conv(W, "uncond_conv_block.%d" % j, xu, st, (k - 1) // 2), 0.2)
        uncond_feats.append(xu)
    return cond_feats, uncond_feats


# ----------------------------------------------------------------------------- losses on the path
def jcu_lsgan(logit_c, logit_u, target):
    """model/loss.py:14-19 (mask=None branch, the only one train.py uses)."""
    lc = F.mse_loss(logit_c, torch.full_like(logit_c, target))
    lu = F.mse_loss(logit_u, torch.full_like(logit_u, target))
    return 0.5 * (lc + lu)


def d_loss(real_c, real_u, fake_c, fake_u):
    """model/loss.py:21-24."""
    return jcu_lsgan(real_c, real_u, 1.0), jcu_lsgan(fake_c, fake_u, 0.0)


def g_loss(fake_c, fake_u):
    """model/loss.py:26-28."""
    return jcu_lsgan(fake_c, fake_u, 1.0)


def fm_loss(real_c, real_u, fake_c, fake_u, n_layers=5):
    """model/loss.py:221-227: sum_{j<len-1} (4/(n_layers+1)) * 0.5 * (L1 + L1); unscaled by lambda_fm."""
    w = 4.0 / (n_layers + 1)
    tot = 0
    for j in range(len(fake_c) - 1):
        tot = tot + w * 0.5 * (F.l1_loss(real_c[j].detach(), fake_c[j]) + F.l1_loss(real_u[j].detach(), fake_u[j]))
    return tot


def mel_l1(pred, target, pad_mask):
    """model/loss.py:229-242,255-259: masked_fill pads with 0, L1 weighted by non-zero target rows."""
    pred = pred.masked_fill(pad_mask.unsqueeze(-1), 0)
    target = target.masked_fill(pad_mask.unsqueeze(-1), 0)
    per = F.l1_loss(pred, target, reduction="none")
    w = target.abs().sum(-1, keepdim=True).ne(0).float().repeat(1, 1, target.size(-1))
    return (per * w).sum() / w.sum()


# ----------------------------------------------------------------------------- FFT blocks (shallow / aux)
def sinusoid_table(n_position, d_hid):
    """transformer/Models.py:10-30 (float64 numpy semantics, cast to fp32)."""
    pos = torch.arange(n_position, dtype=torch.float64)[:, None]
    j = torch.arange(d_hid, dtype=torch.float64)[None, :]
    ang = pos / torch.pow(torch.tensor(10000.0, dtype=torch.float64), 2 * torch.div(j, 2, rounding_mode="floor") / d_hid)
    tab = torch.zeros(n_position, d_hid, dtype=torch.float64)
    tab[:, 0::2] = torch.sin(ang[:, 0::2])
    tab[:, 1::2] = torch.cos(ang[:, 1::2])
    return tab.float()


def _drop(x, p, drop):
    """nn.Dropout(p) / F.dropout(p, training=True) with the keep-mask taken from `drop(shape, p)` (a tape in the
    tests; None = eval / identity)."""
    if drop is None or p <= 0:
        return x
    return x * drop(tuple(x.shape), p).to(x.dtype) / (1.0 - p)


def mha_forward(W, p, x, pad_mask, n_head=2, drop=None, p_drop=0.2):
    """transformer/SubLayers.py:29-57 + Modules.py:16-23 (drop=None: eval, dropout = identity).

    x [B,L,D]; pad_mask bool [B,L] True = pad (keys masked with -inf for every query row).
    """
    B, L, D = x.shape
    dk = D // n_head
    lin = lambda n, a: F.linear(a, W[p + n + ".weight"], W[p + n + ".bias"])
    heads = lambda a: a.view(B, L, n_head, dk).permute(2, 0, 1, 3).reshape(n_head * B, L, dk)
    q, k, v = heads(lin("w_qs", x)), heads(lin("w_ks", x)), heads(lin("w_vs", x))
    att = torch.bmm(q, k.transpose(1, 2)) / (dk ** 0.5)
    km = pad_mask.unsqueeze(1).expand(-1, L, -1).repeat(n_head, 1, 1)
    att = torch.softmax(att.masked_fill(km, float("-inf")), dim=2)
    o = torch.bmm(att, v).view(n_head, B, L, dk).permute(1, 2, 0, 3).reshape(B, L, D)
    o = _drop(lin("fc", o), p_drop, drop)
    return F.layer_norm(o + x, (D,), W[p + "layer_norm.weight"], W[p + "layer_norm.bias"], 1e-5)


def ffn_forward(W, p, x, drop=None, p_drop=0.2):
    """transformer/SubLayers.py:85-93."""
    D = x.shape[-1]
    k = W[p + "w_1.weight"].shape[-1]
    h = F.relu(F.conv1d(x.transpose(1, 2), W[p + "w_1.weight"], W[p + "w_1.bias"], padding=(k - 1) // 2))
    o = _drop(F.conv1d(h, W[p + "w_2.weight"], W[p + "w_2.bias"]).transpose(1, 2), p_drop, drop)
    return F.layer_norm(o + x, (D,), W[p + "layer_norm.weight"], W[p + "layer_norm.bias"], 1e-5)


def fft_block(W, p, x, pad_mask, n_head=2, drop=None, p_drop=0.2):
    """transformer/Layers.py:21-30."""
    y = mha_forward(W, p + "slf_attn.", x, pad_mask, n_head, drop, p_drop).masked_fill(pad_mask.unsqueeze(-1), 0)
    return ffn_forward(W, p + "pos_ffn.", y, drop, p_drop).masked_fill(pad_mask.unsqueeze(-1), 0)


def decoder_forward(W, p, x, pad_mask, max_seq_len, n_layers=6, n_head=2, training=False, drop=None, p_drop=0.2):
    """transformer/Models.py:139-171."""
    B, L, D = x.shape
    if (not training) and L > max_seq_len:
        y = x + sinusoid_table(L, D)[None, :L, :]
    else:
        L = min(L, max_seq_len)
        y = x[:, :L, :] + W[p + "position_enc"][:, :L, :]
        pad_mask = pad_mask[:, :L]
    for i in range(n_layers):
        y = fft_block(W, p + "layer_stack.%d." % i, y, pad_mask, n_head, drop if training else None, p_drop)
    return y


def postnet_forward(W, p, x, n_conv=5, training=False, drop=None):
    """transformer/Layers.py:129-137.  eval: BatchNorm running stats, dropout off.  training: batch statistics
    (the running_mean / running_var entries of W are updated in place like nn.BatchNorm1d, momentum 0.1) and
    F.dropout(., 0.5) after every layer with keep-masks from `drop`."""
    y = x.transpose(1, 2)
    for i in range(n_conv):
        c = p + "convolutions.%d." % i
        k = W[c + "0.conv.weight"].shape[-1]
        y = F.conv1d(y, W[c + "0.conv.weight"], W[c + "0.conv.bias"], padding=(k - 1) // 2)
        y = F.batch_norm(y, W[c + "1.running_mean"], W[c + "1.running_var"], W[c + "1.weight"], W[c + "1.bias"],
                         training, 0.1, 1e-5)
        if i < n_conv - 1:
            y = torch.tanh(y)
        if training:
            y = _drop(y, 0.5, drop)
    return y.transpose(1, 2)


def coarse_mel(W, x, pad_mask, max_seq_len, n_layers=6, n_head=2, training=False, drop=None):
    """model/mixgantts.py:140-143: Decoder -> mel_linear -> PostNet residual."""
    h = decoder_forward(W, "decoder.", x, pad_mask, max_seq_len, n_layers, n_head, training, drop)
    m = F.linear(h, W["mel_linear.weight"], W["mel_linear.bias"])
    return postnet_forward(W, "postnet.", m, training=training, drop=drop) + m


def aux_acoustic_losses(W, buf, x, mel_targets, pad_mask, max_seq_len, T, tape, drop):
    """The acoustic part of the aux-mode loss (model/loss.py:153-161): mae(postnet_output, mel) + sum over the
    diffuse_trace entries of the masked mel L1; returns (mel_loss, postnet_loss, coarse)."""
    coarse = coarse_mel(W, x, pad_mask, max_seq_len, training=True, drop=drop)
    pad = pad_mask[:, :coarse.shape[1]]
    target = mel_targets[:, :coarse.shape[1], :]
    trace = diffuse_trace(buf, coarse, pad, T, tape)
    mel_loss = sum(mel_l1(denorm_spec(tr, buf["spec_min"], buf["spec_max"]), target, pad) for tr in trace)
    return mel_loss, F.l1_loss(coarse, target), coarse


def mixgantts_forward(W, buf, model, T, enc_out, src_masks, src_w_masks, src_lens, speaker_emb, mels, coarse_training,
                      tape, drop=None, max_seq_len=1000, p_targets=None):
    """model/mixgantts.py:55-183 downstream of the linguistic encoder (whose nine outputs `enc_out` are given:
    model/linguistic_encoder.py:373-383).  Returns ([16 slots], p_targets, coarse_mels) with the reference's
    detach pattern: in `shallow` the diffusion sees detached cond / speaker / mask / coarse mel, slots 2, 5, 7-11 and
    the third return value are detached, while slot 15 (`postnet_outputs`) is NOT (:140-143,180) -- it carries the
    decoder's graph into postnet_loss (model/loss.py:165-167).
    src_masks / src_w_masks: get_mask_from_lengths of :77-78 (True = valid); mels None = inference;
    coarse_training: whether Decoder / PostNet run in train mode (dropout masks from `drop`, batch-stat BatchNorm)."""
    output, p_pred, e_pred, log_d_pred, d_rounded, mel_lens, mel_masks, alignments, logprobs = enc_out
    det = (lambda a: a.detach() if (a is not None and model == "shallow") else a)
    mel_masks = ~mel_masks                                  # True = pad from here on (:123,138)
    x_ts = x_prevs = x_prev_preds = t = None
    coarse = postnet_outputs = None
    if model == "naive":
        output, x_ts, x_prevs, x_prev_preds, t = diffusion_forward(W_sub(W, "diffusion."), buf, model, T, mels, output,
                                                                   speaker_emb, mel_masks, None, tape)
    else:
        cond = output.clone()
        coarse = coarse_mel(W, output, mel_masks, max_seq_len, training=coarse_training, drop=drop)
        postnet_outputs = coarse
        if model == "aux":
            output = diffuse_trace(buf, coarse, mel_masks[:, :coarse.shape[1]], T, tape)
        else:
            output, x_ts, x_prevs, x_prev_preds, t = diffusion_forward(
                W_sub(W, "diffusion."), buf, model, T, mels, det(cond), det(speaker_emb), det(mel_masks), det(coarse), tape)
    return [output, (x_ts, x_prevs, x_prev_preds), det(speaker_emb), t, p_pred, det(e_pred), log_d_pred, det(d_rounded),
            det(src_masks), det(mel_masks), det(src_lens), det(mel_lens), alignments, logprobs, src_w_masks,
            postnet_outputs], p_targets, det(coarse)


def W_sub(W, prefix):
    """The sub-dict of a flat weight dict under `prefix`, with the prefix stripped."""
    n = len(prefix)
    return {k[n:]: v for k, v in W.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------- linguistic-encoder index ops
# (the four host-loop functions of the out-of-scope LinguisticEncoder that
#  serialise the GPU with one .item() per phoneme.  Plain loops here, like the reference.)
def word_level_pooling(src_seq, src_len, wb, src_w_len, reduce="sum"):
    """utils/tools.py:394-413.  src_seq [B,Tp,H]; wb [B,Tw] phones per word -> [B, max(src_w_len), H]."""
    outs = []
    for s, sl, w, wl in zip(src_seq, src_len, wb, src_w_len):
        sizes = [int(v) for v in w[: int(wl)]]
        rows, p = [], 0
        for n in sizes:
            seg = s[p:p + n]
            acc = torch.zeros_like(s[0])
            for r in seg:                      # in-order accumulation, as torch.sum over the padded dim
                acc = acc + r
            rows.append(acc / n if reduce == "mean" else acc)
            p += n
        outs.append(torch.stack(rows) if rows else s.new_zeros(0, s.shape[1]))
    Wm = max(o.shape[0] for o in outs)
    return torch.stack([F.pad(o, (0, 0, 0, Wm - o.shape[0])) for o in outs])


def length_regulate(x, duration, max_len=None):
    """model/linguistic_encoder.py:383-416.  x [B,Tw,H], duration [B,Tw] -> ([B,Lmax,H], mel_len int64 [B])."""
    outs, lens = [], []
    for xb, db in zip(x, duration):
        rows = [xb[i].expand(max(int(db[i]), 0), -1) for i in range(xb.shape[0])]
        e = torch.cat(rows, 0)
        outs.append(e)
        lens.append(e.shape[0])
    Lm = max_len if max_len else max(lens)
    outs = [F.pad(e, (0, 0, 0, Lm - e.shape[0])) for e in outs]     # negative pad crops, as F.pad does
    return torch.stack(outs), torch.tensor(lens, dtype=torch.long)


def mapping_mask(Lq, Lkv, dur_w, wb, src_w_len):
    """model/linguistic_encoder.py:185-199: True inside the (frames of word i) x (phonemes of word i) blocks."""
    B = dur_w.shape[0]
    m = torch.ones(B, Lq, Lkv)
    for b in range(B):
        l = int(src_w_len[b])
        cw = [0] + [int(v) for v in torch.cumsum(dur_w[b, :l], 0)]
        cp = [0] + [int(v) for v in torch.cumsum(wb[b, :l], 0)]
        for i in range(1, len(cw)):
            m[b, cw[i - 1]:cw[i], cp[i - 1]:cp[i]] = 0
    return m == 0.0


def rel_coef(dur, dur_len, mask):
    """model/linguistic_encoder.py:222-236: position-in-segment / segment length, 0 on padding."""
    idx, seg = [], []
    for d, dl in zip(dur, dur_len):
        d = d[: int(dl)].long()
        seg.append(torch.repeat_interleave(d, d))
        ib = []
        for di in d:
            ib += list(range(int(di)))
        idx.append(torch.tensor(ib, dtype=torch.long))
    Lm = max(i.shape[0] for i in idx)
    pi = torch.stack([F.pad(i, (0, Lm - i.shape[0])) for i in idx])
    ps = torch.stack([F.pad(s_, (0, Lm - s_.shape[0])) for s_ in seg])
    return torch.div(pi, ps.masked_fill(mask == 0.0, 1))


# ----------------------------------------------------------------------------- HiFi-GAN generator (vocoder, row f3)
HIFIGAN_V1 = dict(upsample_rates=(8, 8, 2, 2), upsample_kernel_sizes=(16, 16, 4, 4), upsample_initial_channel=512,
                  resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)))


def _wn(W, name):
    """weight_norm (dim=0): w = g * v / ||v|| over all dims but the first; or the plain weight after
    remove_weight_norm (hifigan/models.py:167-173)."""
    if name + ".weight" in W:
        return W[name + ".weight"]
    v, g = W[name + ".weight_v"], W[name + ".weight_g"]
    return v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))


def hifigan_forward(W, mel, h=HIFIGAN_V1):
    """hifigan/models.py:145-165.  mel [B, 80, L] -> wav [B, 1, L * prod(upsample_rates)]."""
    nk = len(h["resblock_kernel_sizes"])
    x = F.conv1d(mel, _wn(W, "conv_pre"), W["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, _wn(W, "ups.%d" % i), W["ups.%d.bias" % i], stride=u, padding=(k - u) // 2)
        xs = None
        for j, (ks, dils) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            p = "resblocks.%d." % (i * nk + j)
            r = x
            for m, d in enumerate(dils):
                t = F.leaky_relu(r, 0.1)
                t = F.conv1d(t, _wn(W, p + "convs1.%d" % m), W[p + "convs1.%d.bias" % m], dilation=d,
                             padding=(ks * d - d) // 2)
                t = F.leaky_relu(t, 0.1)
                t = F.conv1d(t, _wn(W, p + "convs2.%d" % m), W[p + "convs2.%d.bias" % m], padding=(ks - 1) // 2)
                r = t + r
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)          # default slope 0.01, not LRELU_SLOPE (hifigan/models.py:161)
    x = F.conv1d(x, _wn(W, "conv_post"), W["conv_post.bias"], padding=3)
    return torch.tanh(x)
