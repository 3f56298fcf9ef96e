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
                 resume=None, first_step=1):
        """extra_g_params: generator parameters outside `diffusion` that the G optimizer also steps (the injected
        linguistic encoder, decoder ...).  g_param_order: the list the reference builds optG over (`model.parameters()`,
        utils/model.py:33) so that `optG.state_dict()` indexes parameters the same way (default: diffusion's, then the
        extras).  resume: (optG, optD, sdlG, sdlD) as returned by `get_model(..., train=True)` -- their restored state
        (moments, step counts, learning rates, scheduler epochs) is taken over.  first_step: train.py's `step` of the
        first call (args.restore_step + 1, train.py:67); it only matters for grad_acc_step > 1 (train.py:80)."""
        self.G, self.D = diffusion, discriminator
        oc = train_config["optimizer"]
        self.grad_clip = oc["grad_clip_thresh"]
        self.lambda_fm = train_config["loss"]["lambda_fm" if diffusion.model != "shallow" else "lambda_fm_shallow"]
        self.n_layers = model_config["discriminator"]["n_layer"] + model_config["discriminator"]["n_cond_layer"]
        # train.py:77-85 (model_update): loss / grad_acc_step, backward; clip + step + zero_grad only when
        # step % grad_acc_step == 0 -- gradients accumulate in .grad in between, and so does the D-gradient leak
        self.grad_acc = int(oc.get("grad_acc_step", 1))
        if self.grad_acc < 1:
            raise ValueError("grad_acc_step must be >= 1")
        self.step_no = int(first_step)
        g_params = list(diffusion.parameters()) + list(extra_g_params)
        # the denoiser's backward writes its weight gradients straight into the G bucket (no gather copy)
        self.bucketG = GradBucket(g_params, order=diffusion.denoise_fn.grad_order())
        self.bucketD = GradBucket(list(discriminator.parameters()))
        # utils/model.py:32-40: Adam(lr, betas) per network.  On the GPU: FlatAdam (optimizer.py) -- parameters and
        # moments flat like the gradients, clip + step in two launches; on CPU (tests) torch's own.  Either way
        # state_dict() indexes the parameters in g_param_order / g_params order, NOT in the bucket's layout order.
        order_g = list(g_param_order) if g_param_order is not None else g_params
        if self.bucketG.flat.is_cuda:
            diffusion.denoise_fn.bind_grad_buffer(self.bucketG.flat, self.bucketG.offsets)
            self.optG = FlatAdam(self.bucketG, lr=oc["init_lr_G"], betas=oc["betas"], param_order=order_g)
            self.optD = FlatAdam(self.bucketD, lr=oc["init_lr_D"], betas=oc["betas"],
                                 param_order=list(discriminator.parameters()))
        else:
            self.optG = torch.optim.Adam(order_g, lr=oc["init_lr_G"], betas=oc["betas"])
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
        # the slice of the G bucket the denoiser's backward finishes first (k=3 conv weight + bias gradients of all
        # layers, contiguous in grad_order): its all-reduce starts while the remaining gradients are still computed
        self._early = None
        den = diffusion.denoise_fn
        if self.bucketG.flat.is_cuda and len(den.residual_layers) > 0:
            first, last = den.residual_layers[0].conv_layer.conv, den.residual_layers[-1].conv_layer.conv
            lo = self.bucketG.offsets.get(id(first.weight))
            hi = self.bucketG.offsets.get(id(last.bias))
            if lo is not None and hi is not None and hi + last.bias.numel() > lo:
                self._early = (lo, hi + last.bias.numel())

    # LSGAN and feature-matching sums as one launch each way (losses.weighted_means); False keeps one launch pair per
    # term through d_loss_fn / g_loss_fn / get_fm_loss (same values up to the order of the additions)
    fused_losses = True
    pair_forwards = True      # False: the two generator forwards of a step as two launches
    overlap_exchange = True   # multi-GPU: start the all-reduce of the k=3 gradients while the backward is still running
    grad_hook = None      # optional callable(name, bucket) after the gradient exchange, before clipping (tests, logging)

    def _update(self, bucket, opt):
        bucket.all_reduce_mean()                      # no-op on one process; waits for chunks already in flight
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
        if self._static_grads and bucket is self.bucketD:
            # captured step: the discriminator's .grad tensors stay the bucket's views (static addresses) and are zeroed in
            # place -- the G phase's deposit then ADDS to zeros where train.py:84-85's zero_grad(set_to_none) makes it assign
            bucket.flat.zero_()
        else:
            opt.zero_grad()                           # after step, as train.py:84-85

    _static_grads = False

    def capture(self, mel, cond, spk, mel_pad_mask, coarse_mel=None, warmup=3):
        """The whole step() -- both generator forwards, four discriminator passes, both backwards, both clip + Adam updates,
        ~270 launches -- as ONE captured hipGraph, replayed per batch.  The discriminator's half of a step is ~100 small
        launches per phase that the host issues more slowly than the GPU runs them (section 4.2 of DESIGN.md); a replay
        has no host in the loop.  Returns a callable with step()'s signature (same shapes as the example batch, tensors
        copied into static buffers) and return value (the loss tensors are static too: read them before the next call).

        What differs from step(): t / noise come from torch's graph-safe generator state; the optimizers' step counts and
        learning rates travel through device memory (FlatAdam.enable_device_hyper), so lr schedulers keep working; the
        discriminator's gradients are zeroed in place instead of set to None (same numbers).  Single process only (the
        gradient exchange is not captured), grad_acc_step = 1, no grad_hook / t_fn / noise_fn."""
        if not (isinstance(self.optG, FlatAdam) and isinstance(self.optD, FlatAdam)):
            raise RuntimeError("capture() needs the GPU trainer (FlatAdam optimizers)")
        if self.bucketG.exchanging() or self.grad_acc != 1 or self.grad_hook is not None or self.G.t_fn or self.G.noise_fn:
            raise RuntimeError("capture(): single process, grad_acc_step = 1, no hooks")
        dev = mel.device
        static = [None if a is None else a.detach().clone() for a in (mel, cond, spk, mel_pad_mask, coarse_mel)]
        self.optG.enable_device_hyper()
        self.optD.enable_device_hyper()
        self._static_grads = True
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):          # warm-up on the capture stream: workspaces and scratch are keyed by stream
            for _ in range(max(2, warmup)):
                self.step(*static)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.check(sync=False)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out = self.step(*static)
        # the capture itself ran nothing: undo its host-side bookkeeping (it will be redone per replay)
        self.optG._steps -= 1
        self.optD._steps -= 1
        self.step_no -= 1
        return _GraphedStep(self, graph, static, out)

    def _model_update(self, loss, bucket, opt):
        """train.py:75-85."""
        updating = self.step_no % self.grad_acc == 0
        den = self.G.denoise_fn
        hook_set = False
        if (updating and bucket is self.bucketG and self._early is not None and self.overlap_exchange
                and bucket.exchanging() and os.environ.get("MG_OVERLAP_EXCHANGE", "1") != "0"):
            lo, hi = self._early
            den.after_conv3_grads = lambda event: bucket.all_reduce_chunk_async(lo, hi, event)
            hook_set = True
        try:
            (loss if self.grad_acc == 1 else loss / self.grad_acc).backward()
        finally:
            if hook_set:
                den.after_conv3_grads = None
        if updating:
            self._update(bucket, opt)

    def _d_fake_and_real(self, x_ts, x_fake, x_real, spk, t):
        """D(x_ts, x_fake, s, t) and D(x_ts, x_real, s, t) (train.py:139-140,156-157) as ONE pass over a batch of 2B:
        the discriminator has no batch-coupled op, so the feature maps are the same as from two calls, and every
        conv of its 1/4-rate tail sees twice the frames per launch (at B=8 those launches fill a quarter of the
        GPU).  Returns (fake_cond, fake_uncond, real_cond, real_uncond) lists of feature maps."""
        cond_maps, uncond_maps, B = self._d_both(x_ts, x_fake, x_real, spk, t)
        halves = lambda maps, k: [m[k * B:(k + 1) * B] for m in maps]  # noqa: E731
        return halves(cond_maps, 0), halves(uncond_maps, 0), halves(cond_maps, 1), halves(uncond_maps, 1)

    def _d_both(self, x_ts, x_fake, x_real, spk, t):
        """The one discriminator pass over [fake; real]: whole feature maps (rows 0..B-1 fake, B..2B-1 real) and B."""
        B = x_ts.shape[0]
        both = lambda a, b: None if a is None else torch.cat([a, b], 0)  # noqa: E731
        cond_maps, uncond_maps = self.D(both(x_ts, x_ts), both(x_fake, x_real), both(spk, spk), both(t, t))
        return cond_maps, uncond_maps, B

    # ------------------------------------------------------------------ the two phases on given generator outputs
    def _d_loss(self, x_ts, x_prevs, x_prev_preds, spk, t):
        """train.py:135-144 / evaluate.py:80-88: everything the generator produced is detached."""
        det = lambda a: None if a is None else a.detach()  # noqa: E731
        if self.fused_losses:      # the maps go into the loss whole: no slicing for autograd to undo
            cm, um, B = self._d_both(det(x_ts), det(x_prev_preds), det(x_prevs), det(spk), t)
            return losses.d_loss_total_2b(cm[-1], um[-1], B)[0]
        f_c, f_u, r_c, r_u = self._d_fake_and_real(det(x_ts), det(x_prev_preds), det(x_prevs), det(spk), t)
        d_real, d_fake = self.d_loss_fn(r_c[-1], r_u[-1], f_c[-1], f_u[-1])
        return d_real + d_fake

    def _g_loss(self, x0, x_ts, x_prevs, x_prev_preds, spk, t, mel, mel_pad_mask, coarse_mel, extra_loss):
        """train.py:156-182 with model/loss.py:153-167,196-199: adv + mel L1 (+ postnet L1 in shallow) + lambda_fm * FM
        (+ the caller's upstream terms).  Returns (g_loss, dict of the parts)."""
        G = self.G
        target = coarse_mel.detach() if G.model == "shallow" else mel
        mel_loss = losses.get_mel_loss(G.denorm_spec(x0), target, mel_pad_mask)
        if self.fused_losses:
            cm, um, B = self._d_both(x_ts, x_prev_preds, x_prevs, spk, t)
            adv_fm, adv, fm = losses.g_adv_fm_total_2b(cm, um, B, self.lambda_fm, self.n_layers)
            g_loss = adv_fm + mel_loss
        else:
            f_c, f_u, r_c, r_u = self._d_fake_and_real(x_ts, x_prev_preds, x_prevs, spk, t)
            adv = self.g_loss_fn(f_c[-1], f_u[-1])
            fm = self.lambda_fm * losses.get_fm_loss(r_c, r_u, f_c, f_u, self.n_layers)
            g_loss = adv + mel_loss + fm
        parts = {"adv_loss": adv.detach(), "mel_loss": mel_loss.detach(),
                 "fm_loss": fm.detach() if torch.is_tensor(fm) else fm}
        if G.model == "shallow" and coarse_mel is not None and (coarse_mel.requires_grad or not torch.is_grad_enabled()):
            postnet_loss = losses._L1Fn.apply(coarse_mel, mel[:, :coarse_mel.shape[1], :].contiguous())
            g_loss = g_loss + postnet_loss
            parts["postnet_loss"] = postnet_loss.detach()
        if extra_loss is not None:
            g_loss = g_loss + extra_loss
        return g_loss, parts

    # ------------------------------------------------------------------ train.py:131-184 on a given conditioner
    def step(self, mel, cond, spk, mel_pad_mask, coarse_mel=None, extra_loss=None, cond_d=None, spk_d=None,
             coarse_mel_d=None):
        """One D phase + one G phase on a batch.  mel [B,L,M]; cond [B,L,H]; mel_pad_mask True = pad.
        cond_d / spk_d / coarse_mel_d: the D phase's generator inputs when they differ from the G phase's -- the
        reference calls `model(*(batch[2:]))` once per phase (train.py:133,153), i.e. the train-mode linguistic encoder
        runs twice and its dropout gives two different conditioners; None = the same tensors for both phases.
        extra_loss: added to the generator loss before its backward -- the terms of recon_loss that come from
        modules upstream of the path (model/loss.py:195: lambda_d * duration + lambda_p * pitch + lambda_e * energy
        + helper for the linguistic encoder), so an injected encoder trains jointly through this step.
        shallow: when `coarse_mel` carries gradient (MixGANTTS computes it with grad in training, as the reference
        does, model/mixgantts.py:140-143), postnet_loss = L1(coarse_mel, mel) joins the loss (model/loss.py:165-167).

        Random draws: the D-phase forward's (t, three noises), then the G-phase forward's.  With pair_forwards (one
        launch for both forwards) the G-phase set is drawn right behind the D-phase set; the caller has produced both
        conditioners before this call either way, so relative to train.py only the encoder's second-pass draws have
        moved in front of the first diffusion draws (step_from_model(pair=False) keeps train.py's order exactly)."""
        G = self.G
        cd = cond if cond_d is None else cond_d
        sd = spk if spk_d is None else spk_d
        cmd = coarse_mel if coarse_mel_d is None else coarse_mel_d
        # ---------------- D phase (train.py:133-146)
        # train.py:133 builds (and discards) the generator's autograd graph here; every output is detached
        # before use (train.py:135-137), so running it under no_grad gives identical results and skips
        # the activation saves of the grad-enabled forward.
        # ... and it is launched TOGETHER with the G phase's forward below (same weights: only D is stepped in between;
        # GaussianDiffusion.pair_forward): one grid of 64-frame tiles over both instead of two of 32-frame tiles
        G.pair_forward = self.pair_forwards and os.environ.get("MG_PAIR_FORWARDS", "1") != "0"
        G.pair_inputs = (cond, spk, coarse_mel)
        try:
            with torch.no_grad():
                _, x_ts, x_prevs, x_prev_preds, t = G(mel, cd, sd, mel_pad_mask, cmd)
        finally:
            G.pair_forward = False
            G.pair_inputs = None
        d_loss = self._d_loss(x_ts, x_prevs, x_prev_preds, sd, t)
        self._model_update(d_loss, self.bucketD, self.optD)
        # ---------------- G phase (train.py:153-184)
        x0, x_ts, x_prevs, x_prev_preds, t = G(mel, cond, spk, mel_pad_mask, coarse_mel)
        g_loss, out = self._g_loss(x0, x_ts, x_prevs, x_prev_preds, spk, t, mel, mel_pad_mask, coarse_mel, extra_loss)
        self._model_update(g_loss, self.bucketG, self.optG)
        self.step_no += 1
        out["d_loss"] = d_loss.detach()
        return out

    # ------------------------------------------------------------------ train.py:131-184 around a whole model
    @staticmethod
    def _unpack(output):
        """The slots train.py:135,155 / evaluate.py:79,97 read from MixGANTTS.forward's 16-slot list."""
        (x_ts, x_prevs, x_prev_preds), spk, t, mel_pad_mask = output[1], output[2], output[3], output[9]
        return output[0], x_ts, x_prevs, x_prev_preds, spk, t, mel_pad_mask, output[15]

    def step_from_model(self, model, batch, upstream_loss=None, pair=False):
        """train.py:131-184 verbatim around `model` (mixgan_tts_amd.MixGANTTS whose .diffusion is this trainer's
        generator; the linguistic encoder injected): `model(*(batch[2:]))` for the D phase, D update,
        `model(*(batch[2:]))` again for the G phase (a train-mode encoder gives a different conditioner and consumes
        its random draws between the two sets of diffusion draws, exactly as in the reference), `batch[9] = p_targets`,
        G update.  upstream_loss(batch, output, step_no) -> tensor | None supplies model/loss.py:195's duration / pitch /
        energy / helper terms of the (out-of-scope) linguistic encoder.
        pair=True launches both generator forwards together: the encoder then runs twice up front, which moves its
        second-pass draws ahead of the first diffusion draws (statistically the same step, not draw-for-draw)."""
        if getattr(model, "diffusion", None) is not self.G:
            raise ValueError("step_from_model: model.diffusion is not this trainer's generator")
        if pair:
            return self._step_from_model_paired(model, batch, upstream_loss)
        with torch.no_grad():                    # every D-phase use is detached (train.py:135-137)
            output, *_ = model(*(batch[2:]))
        _, x_ts, x_prevs, x_prev_preds, spk, t, _, _ = self._unpack(output)
        d_loss = self._d_loss(x_ts, x_prevs, x_prev_preds, spk, t)
        self._model_update(d_loss, self.bucketD, self.optD)
        output, p_targets, coarse_mels = model(*(batch[2:]))
        batch[9] = p_targets                     # train.py:155
        x0, x_ts, x_prevs, x_prev_preds, spk, t, mel_pad_mask, slot15 = self._unpack(output)
        mel = batch[11][:, :mel_pad_mask.shape[1], :]
        extra = upstream_loss(batch, output, self.step_no) if upstream_loss is not None else None
        g_loss, out = self._g_loss(x0, x_ts, x_prevs, x_prev_preds, spk, t, mel, mel_pad_mask,
                                   slot15 if self.G.model == "shallow" else None, extra)
        self._model_update(g_loss, self.bucketG, self.optG)
        self.step_no += 1
        out["d_loss"] = d_loss.detach()
        return out

    def _step_from_model_paired(self, model, batch, upstream_loss):
        """Both encoder passes first (the second with grad), then step() with the two conditioners: one launch for both
        generator forwards.  Runs model.forward with its diffusion swapped for a recorder, so the encoder-side code
        is the model's own."""
        rec = _DiffusionInputs()
        real = model.diffusion
        model.diffusion = rec
        try:
            with torch.no_grad():
                model(*(batch[2:]))
            d_in = rec.args
            output, p_targets, _ = model(*(batch[2:]))
            g_in = rec.args
        finally:
            model.diffusion = real
        batch[9] = p_targets
        mel, cond, spk, pad, coarse = g_in
        extra = upstream_loss(batch, output, self.step_no) if upstream_loss is not None else None
        slot15 = output[15] if self.G.model == "shallow" else None
        return self.step(mel, cond, spk, pad, slot15 if slot15 is not None else coarse, extra,
                         cond_d=d_in[1], spk_d=d_in[2], coarse_mel_d=d_in[4])

    @torch.no_grad()
    def evaluate_step(self, mel, cond, spk, mel_pad_mask, coarse_mel=None, extra_loss=None, cond_d=None, spk_d=None,
                      coarse_mel_d=None):
        """The validation step of evaluate.py:70-120 for the path: the same two generator forwards and four
        discriminator passes as step(), under no_grad, no update; returns the same loss dict."""
        G = self.G
        cd = cond if cond_d is None else cond_d
        sd = spk if spk_d is None else spk_d
        cmd = coarse_mel if coarse_mel_d is None else coarse_mel_d
        _, x_ts, x_prevs, x_prev_preds, t = G(mel, cd, sd, mel_pad_mask, cmd)
        d_loss = self._d_loss(x_ts, x_prevs, x_prev_preds, sd, t)
        x0, x_ts, x_prevs, x_prev_preds, t = G(mel, cond, spk, mel_pad_mask, coarse_mel)
        _, out = self._g_loss(x0, x_ts, x_prevs, x_prev_preds, spk, t, mel, mel_pad_mask, coarse_mel, extra_loss)
        out["d_loss"] = d_loss
        return out

    @torch.no_grad()
    def evaluate_from_model(self, model, batch, upstream_loss=None):
        """evaluate.py:76-120 around `model`: two `model(*(batch[2:]))` calls, losses only."""
        output, *_ = model(*(batch[2:]))
        _, x_ts, x_prevs, x_prev_preds, spk, t, _, _ = self._unpack(output)
        d_loss = self._d_loss(x_ts, x_prevs, x_prev_preds, spk, t)
        output, p_targets, coarse_mels = model(*(batch[2:]))
        batch[9] = p_targets
        x0, x_ts, x_prevs, x_prev_preds, spk, t, mel_pad_mask, slot15 = self._unpack(output)
        mel = batch[11][:, :mel_pad_mask.shape[1], :]
        extra = upstream_loss(batch, output, self.step_no) if upstream_loss is not None else None
        _, out = self._g_loss(x0, x_ts, x_prevs, x_prev_preds, spk, t, mel, mel_pad_mask,
                              slot15 if self.G.model == "shallow" else None, extra)
        out["d_loss"] = d_loss
        return out

    def log_scalars(self, out):
        """The host's read-back of a step's losses (train.py:198-199 `.item()`s): one synchronisation, after which the
        single-launch kernels' failure word is exact -- raises MixganHipError if a hand-off timed out in this step
        (its gradients were NaN then)."""
        vals = {k: (float(v) if torch.is_tensor(v) else v) for k, v in out.items()}
        self.check(sync=False)
        return vals

    def check(self, sync=True):
        from .denoiser import raise_if_failed
        if self.bucketG.flat.is_cuda:
            raise_if_failed((self.G.denoise_fn,), sync)

    def end_epoch(self):
        self.sdlG.step()
        self.sdlD.step()


class _GraphedStep:
    """A captured HotPathTrainer.step (HotPathTrainer.capture)."""

    def __init__(self, trainer, graph, static, out):
        self.trainer, self.graph, self.static, self.out = trainer, graph, static, out
        self._params = list(trainer.bucketG.params) + list(trainer.bucketD.params)

    def __call__(self, mel, cond, spk, mel_pad_mask, coarse_mel=None):
        tr = self.trainer
        for dst, src in zip(self.static, (mel, cond, spk, mel_pad_mask, coarse_mel)):
            if (dst is None) != (src is None) or (dst is not None and dst.shape != src.shape):
                raise ValueError("captured step: batch shapes differ from the captured example")
            if dst is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        for opt in (tr.optD, tr.optG):         # the step counts / learning rates this replay's updates use
            opt.write_hyper(float(opt._steps) + 1.0)
        self.graph.replay()
        tr.optD._steps += 1
        tr.optG._steps += 1
        tr.step_no += 1
        torch.autograd.graph.increment_version(self._params)     # derived caches (packed weights) of eager users
        return self.out


class _DiffusionInputs(torch.nn.Module):
    """Stands in for model.diffusion while _step_from_model_paired records what MixGANTTS.forward would hand it."""
    model = None

    def forward(self, mel, cond, spk_emb, mel_mask, coarse_mel=None):
        self.args = (mel, cond, spk_emb, mel_mask, coarse_mel)
        return None, None, None, None, None

    def diffuse_trace(self, *a):
        raise RuntimeError("the GAN step does not run the aux model")


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
