"""The GAN training step of train.py:91-184 for the hot path (generator = GaussianDiffusion on a
given conditioner, discriminator = JCUDiscriminator), with the reference's quirks kept:

  * D phase first: generator forward (graph built, outputs detached), D(fake), D(real), LSGAN
    d_loss, backward, clip_grad_norm_(D, 1.0), optD.step(), then zero_grad (train.py:75-85);
  * G phase: a SECOND generator forward (new t, new noise), D(fake)/D(real) without freezing D,
    adv + mel L1 + lambda_fm * FM, backward, clip, optG.step(), zero_grad.  The G-phase backward
    therefore also deposits gradients into D's parameters, which are NOT cleared before the next
    D-phase backward (SURVEY.md section 3.1 "D-grad leak") -- reproduced.
  * multi-GPU: gradients are all-reduced (mean) per optimizer before clipping (distributed.py).

The linguistic encoder is upstream of the path: `cond` is whatever produced
the [B, L, 256] conditioner; its gradient is returned to the caller's graph as usual.
"""
import os

import torch

from . import losses
from .distributed import GradBucket
from .optimizer import FlatAdam


class HotPathTrainer:
    def __init__(self, diffusion, discriminator, train_config, model_config, extra_g_params=(), g_param_order=None,
                 resume=None):
        """extra_g_params: generator parameters outside `diffusion` that the G optimizer also steps (the injected
        linguistic encoder, decoder ...).  g_param_order: the list the reference builds optG over (`model.parameters()`,
        utils/model.py:33) so that `optG.state_dict()` indexes parameters the same way (default: diffusion's, then the
        extras).  resume: (optG, optD, sdlG, sdlD) as returned by `get_model(..., train=True)` -- their restored state
        (moments, step counts, learning rates, scheduler epochs) is taken over."""
        self.G, self.D = diffusion, discriminator
        oc = train_config["optimizer"]
        self.grad_clip = oc["grad_clip_thresh"]
        self.lambda_fm = train_config["loss"]["lambda_fm" if diffusion.model != "shallow" else "lambda_fm_shallow"]
        self.n_layers = model_config["discriminator"]["n_layer"] + model_config["discriminator"]["n_cond_layer"]
        if oc.get("grad_acc_step", 1) != 1:
            raise NotImplementedError("HotPathTrainer steps the optimizers every call (train.py:77-85 with "
                                      "grad_acc_step = 1, the value every shipped config sets)")
        g_params = list(diffusion.parameters()) + list(extra_g_params)
        # the denoiser's backward writes its weight gradients straight into the G bucket (no gather copy)
        self.bucketG = GradBucket(g_params, order=diffusion.denoise_fn.grad_order())
        self.bucketD = GradBucket(list(discriminator.parameters()))
        # utils/model.py:32-40: Adam(lr, betas) per network.  On the GPU: FlatAdam (optimizer.py) -- parameters and
        # moments flat like the gradients, clip + step in two launches; on CPU (tests) torch's own
        if self.bucketG.flat.is_cuda:
            diffusion.denoise_fn.bind_grad_buffer(self.bucketG.flat, self.bucketG.offsets)
            self.optG = FlatAdam(self.bucketG, lr=oc["init_lr_G"], betas=oc["betas"], param_order=g_param_order)
            self.optD = FlatAdam(self.bucketD, lr=oc["init_lr_D"], betas=oc["betas"],
                                 param_order=list(discriminator.parameters()))
        else:
            self.optG = torch.optim.Adam(g_param_order or g_params, lr=oc["init_lr_G"], betas=oc["betas"])
            self.optD = torch.optim.Adam(discriminator.parameters(), lr=oc["init_lr_D"], betas=oc["betas"])
        if resume is not None:
            for mine, theirs in ((self.optG, resume[0]), (self.optD, resume[1])):
                if isinstance(mine, FlatAdam):
                    mine.adopt(theirs)
                else:
                    mine.load_state_dict(theirs.state_dict())
        self.sdlG = torch.optim.lr_scheduler.ExponentialLR(self.optG, gamma=oc["gamma"])    # stepped per EPOCH
        self.sdlD = torch.optim.lr_scheduler.ExponentialLR(self.optD, gamma=oc["gamma"])
        if resume is not None:
            self.sdlG.load_state_dict(resume[2].state_dict())
            self.sdlD.load_state_dict(resume[3].state_dict())
        self.d_loss_fn, self.g_loss_fn = losses.get_adversarial_losses_fn(train_config["loss"]["adv_loss_mode"])

    # LSGAN and feature-matching sums as one launch each way (losses.weighted_means); False keeps one launch pair per
    # term through d_loss_fn / g_loss_fn / get_fm_loss (same values up to the order of the additions)
    fused_losses = True
    pair_forwards = True      # False: the two generator forwards of a step as two launches
    grad_hook = None      # optional callable(name, bucket) after the gradient exchange, before clipping (tests, logging)

    def _update(self, params, bucket, opt):
        bucket.all_reduce_mean()                      # no-op on one process
        if self.grad_hook is not None:
            self.grad_hook("G" if bucket is self.bucketG else "D", bucket)
        if isinstance(opt, FlatAdam):
            # clip_grad_norm_(params, clip) of train.py:81 folded into the update: one norm pass, one Adam pass
            opt.step(max_grad_norm=self.grad_clip)
        else:
            # ... on the flat bucket every .grad now aliases: one norm, one scale
            total = torch.linalg.vector_norm(bucket.flat)
            bucket.flat.mul_(torch.clamp(self.grad_clip / (total + 1e-6), max=1.0))
            opt.step()
        opt.zero_grad()                               # after step, as train.py:84-85

    def _d_fake_and_real(self, x_ts, x_fake, x_real, spk, t):
        """D(x_ts, x_fake, s, t) and D(x_ts, x_real, s, t) (train.py:139-140,156-157) as ONE pass over a batch of 2B:
        the discriminator has no batch-coupled op, so the feature maps are the same as from two calls, and every
        conv of its 1/4-rate tail sees twice the frames per launch (at B=8 those launches fill a quarter of the
        GPU).  Returns (fake_cond, fake_uncond, real_cond, real_uncond) lists of feature maps."""
        B = x_ts.shape[0]
        both = lambda a, b: None if a is None else torch.cat([a, b], 0)  # noqa: E731
        cond_maps, uncond_maps = self.D(both(x_ts, x_ts), both(x_fake, x_real), both(spk, spk), both(t, t))
        halves = lambda maps, k: [m[k * B:(k + 1) * B] for m in maps]  # noqa: E731
        return halves(cond_maps, 0), halves(uncond_maps, 0), halves(cond_maps, 1), halves(uncond_maps, 1)

    def step(self, mel, cond, spk, mel_pad_mask, coarse_mel=None, extra_loss=None):
        """One D phase + one G phase on a batch.  mel [B,L,M]; cond [B,L,H]; mel_pad_mask True = pad.
        extra_loss: added to the generator loss before its backward -- the terms of recon_loss that come from
        modules upstream of the path (model/loss.py:195: lambda_d * duration + lambda_p * pitch + lambda_e * energy
        + helper for the linguistic encoder), so an injected encoder trains jointly through this step.
        shallow: when `coarse_mel` carries gradient (MixGANTTS computes it with grad in training, as the reference
        does, model/mixgantts.py:140-143), postnet_loss = L1(coarse_mel, mel) joins the loss (model/loss.py:165-167)."""
        G, D = self.G, self.D
        # ---------------- D phase (train.py:133-146)
        # train.py:133 builds (and discards) the generator's autograd graph here; every output is detached
        # before use (train.py:135-137), so running it under no_grad gives identical results and skips
        # the activation saves of the grad-enabled forward.
        # ... and it is launched TOGETHER with the G phase's forward below (same weights: only D is stepped in between;
        # GaussianDiffusion.pair_forward): one grid of 64-frame tiles over both instead of two of 32-frame tiles
        G.pair_forward = self.pair_forwards and os.environ.get("MG_PAIR_FORWARDS", "1") != "0"
        try:
            with torch.no_grad():
                x0, x_ts, x_prevs, x_prev_preds, t = G(mel, cond, spk, mel_pad_mask, coarse_mel)
        finally:
            G.pair_forward = False
        x_ts_d, x_prevs_d, x_pp_d = x_ts.detach(), x_prevs.detach(), x_prev_preds.detach()
        spk_d = spk.detach() if spk is not None else None
        f_c, f_u, r_c, r_u = self._d_fake_and_real(x_ts_d, x_pp_d, x_prevs_d, spk_d, t)
        if self.fused_losses:
            d_loss, _, _ = losses.d_loss_total(r_c[-1], r_u[-1], f_c[-1], f_u[-1])
        else:
            d_real, d_fake = self.d_loss_fn(r_c[-1], r_u[-1], f_c[-1], f_u[-1])
            d_loss = d_real + d_fake
        d_loss.backward()
        self._update(list(D.parameters()), self.bucketD, self.optD)
        # ---------------- G phase (train.py:153-184)
        x0, x_ts, x_prevs, x_prev_preds, t = G(mel, cond, spk, mel_pad_mask, coarse_mel)
        f_c, f_u, r_c, r_u = self._d_fake_and_real(x_ts, x_prev_preds, x_prevs, spk, t)
        target = coarse_mel.detach() if G.model == "shallow" else mel
        mel_loss = losses.get_mel_loss(G.denorm_spec(x0), target, mel_pad_mask)
        if self.fused_losses:
            adv_fm, adv, fm = losses.g_adv_fm_total(r_c, r_u, f_c, f_u, self.lambda_fm, self.n_layers)
            g_loss = adv_fm + mel_loss
        else:
            adv = self.g_loss_fn(f_c[-1], f_u[-1])
            fm = self.lambda_fm * losses.get_fm_loss(r_c, r_u, f_c, f_u, self.n_layers)
            g_loss = adv + mel_loss + fm
        out = {}
        if G.model == "shallow" and coarse_mel is not None and coarse_mel.requires_grad:
            postnet_loss = losses._L1Fn.apply(coarse_mel, mel[:, :coarse_mel.shape[1], :].contiguous())
            g_loss = g_loss + postnet_loss
            out["postnet_loss"] = postnet_loss.detach()
        if extra_loss is not None:
            g_loss = g_loss + extra_loss
        g_loss.backward()
        self._update(self.bucketG.params, self.bucketG, self.optG)
        out.update({"d_loss": d_loss.detach(), "adv_loss": adv.detach(), "mel_loss": mel_loss.detach(),
                    "fm_loss": fm.detach() if torch.is_tensor(fm) else fm})
        return out

    def end_epoch(self):
        self.sdlG.step()
        self.sdlD.step()


class AuxTrainer:
    """The `--model aux` step of train.py:97-128 for the modules on the HIP path: coarse mel (Decoder ->
    mel_linear -> PostNet, train mode) -> diffuse_trace -> acoustic reconstruction loss (model/loss.py:153-161:
    mae(postnet_output, mel) + sum_t masked-L1(denorm(trace_t), mel)) -> backward -> gradient all-reduce ->
    clip_grad_norm_ -> ScheduledOptim (model/optimizer.py).  The linguistic encoder's own loss terms
    (duration / pitch / energy / alignment helper, out of scope) enter through `extra_loss`.

    model: a module with `.coarse_mel(cond, pad_mask)` and `.diffusion` (mixgan_tts_amd.MixGANTTS built with
    args.model == "aux"); `params` defaults to model.parameters() (include the linguistic encoder's there)."""

    def __init__(self, model, train_config, model_config, current_step=0, params=None):
        from .optimizer import ScheduledOptim
        self.model = model
        self.params = [p for p in (params if params is not None else model.parameters()) if p.requires_grad]
        holder = type("_P", (), {"parameters": lambda s: iter(self.params)})()
        self.opt = ScheduledOptim(holder, train_config, model_config, current_step)
        self.grad_clip = train_config["optimizer"]["grad_clip_thresh"]
        self.bucket = GradBucket(self.params)
        if self.bucket.flat.is_cuda:      # clip + Adam as two launches on flat buffers (optimizer.FlatAdam)
            self.opt.use_flat(self.bucket)

    def acoustic_losses(self, cond, mel_targets, mel_pad_mask):
        m = self.model
        coarse = m.coarse_mel(cond, mel_pad_mask)
        Lc = coarse.shape[1]
        pad, target = mel_pad_mask[:, :Lc], mel_targets[:, :Lc, :].contiguous()
        trace = m.diffusion.diffuse_trace(coarse, pad)
        mel_loss = 0
        for tr in trace:
            mel_loss = mel_loss + losses.get_mel_loss(m.diffusion.denorm_spec(tr), target, pad)
        return mel_loss, losses._L1Fn.apply(coarse, target), coarse

    def step(self, cond, mel_targets, mel_pad_mask, extra_loss=None):
        mel_loss, postnet_loss, _ = self.acoustic_losses(cond, mel_targets, mel_pad_mask)
        loss = mel_loss + postnet_loss
        if extra_loss is not None:
            loss = loss + extra_loss
        loss.backward()
        self.bucket.all_reduce_mean()
        if isinstance(self.opt._optimizer, FlatAdam):
            lr = self.opt.step(max_grad_norm=self.grad_clip)
        else:
            torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip)
            lr = self.opt.step()
        self.opt.zero_grad()
        return {"mel_loss": mel_loss.detach(), "postnet_loss": postnet_loss.detach(), "lr": lr}
