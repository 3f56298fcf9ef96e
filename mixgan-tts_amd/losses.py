"""Losses that enter the hot path's backward (model/loss.py:12-30, 221-242, 255-259): LSGAN
on the last feature map of the JCU discriminator, feature matching over the first four maps and
the masked mel L1.  Reductions and their gradients run in the HIP library."""
import torch

from . import _lib
from ._lib import fptr, iptr, check, stream_ptr


class _MseConstFn(torch.autograd.Function):
    """F.mse_loss(x, full_like(x, c)) -- mean (x - c)^2."""

    @staticmethod
    def forward(ctx, x, c):
        x = x.contiguous()
        out = torch.empty(1, device=x.device)
        check(_lib.lib().mg_loss_sum(fptr(x), None, float(c), 0, x.numel(), fptr(out), stream_ptr()))
        ctx.save_for_backward(x)
        ctx.c = float(c)
        return out[0] / x.numel()

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        check(_lib.lib().mg_loss_grad(fptr(x), None, ctx.c, 0, fptr(g.reshape(1).contiguous()), 1.0 / x.numel(),
                                      x.numel(), fptr(dx), stream_ptr()))
        return dx, None


class _L1Fn(torch.autograd.Function):
    """F.l1_loss(target, pred) with gradient to `pred` only (the reference detaches the real maps)."""

    @staticmethod
    def forward(ctx, pred, target):
        pred, target = pred.contiguous(), target.detach().contiguous()
        out = torch.empty(1, device=pred.device)
        check(_lib.lib().mg_loss_sum(fptr(pred), fptr(target), 0.0, 1, pred.numel(), fptr(out), stream_ptr()))
        ctx.save_for_backward(pred, target)
        return out[0] / pred.numel()

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        d = torch.empty_like(pred)
        check(_lib.lib().mg_loss_grad(fptr(pred), fptr(target), 0.0, 1, fptr(g.reshape(1).contiguous()),
                                      1.0 / pred.numel(), pred.numel(), fptr(d), stream_ptr()))
        return d, None


def _jcu_loss(logit_cond, logit_uncond, label, mask=None):
    if mask is not None:
        raise NotImplementedError("train.py never passes a mask to the adversarial losses (train.py:142,162)")
    return 0.5 * (_MseConstFn.apply(logit_cond, label) + _MseConstFn.apply(logit_uncond, label))


def get_lsgan_losses_fn():
    """model/loss.py:12-30."""

    def d_loss_fn(r_logit_cond, r_logit_uncond, f_logit_cond, f_logit_uncond, mask=None):
        return _jcu_loss(r_logit_cond, r_logit_uncond, 1.0, mask), _jcu_loss(f_logit_cond, f_logit_uncond, 0.0, mask)

    def g_loss_fn(f_logit_cond, f_logit_uncond, mask=None):
        return _jcu_loss(f_logit_cond, f_logit_uncond, 1.0, mask)

    return d_loss_fn, g_loss_fn


def get_adversarial_losses_fn(mode):
    if mode == "lsgan":
        return get_lsgan_losses_fn()
    raise NotImplementedError(mode)


def get_fm_loss(D_real_cond, D_real_uncond, D_fake_cond, D_fake_uncond, n_layers=5):
    """model/loss.py:221-227 (unscaled by lambda_fm)."""
    w = 4.0 / (n_layers + 1)
    tot = 0
    for j in range(len(D_fake_cond) - 1):
        tot = tot + w * 0.5 * (_L1Fn.apply(D_fake_cond[j], D_real_cond[j]) + _L1Fn.apply(D_fake_uncond[j], D_real_uncond[j]))
    return tot


class _MelL1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, pad):
        pred, target = pred.contiguous(), target.detach().contiguous()
        B, L, M = pred.shape
        pad8 = pad.to(torch.uint8).contiguous()
        out = torch.empty(2, device=pred.device)
        check(_lib.lib().mg_mel_l1_fwd(fptr(pred), fptr(target), iptr(pad8, torch.uint8), B * L, M, fptr(out),
                                       stream_ptr()))
        ctx.save_for_backward(pred, target, pad8, out)
        return out[0] / out[1]

    @staticmethod
    def backward(ctx, g):
        pred, target, pad8, out = ctx.saved_tensors
        B, L, M = pred.shape
        d = torch.empty_like(pred)
        check(_lib.lib().mg_mel_l1_bwd(fptr(pred), fptr(target), iptr(pad8, torch.uint8), B * L, M,
                                       fptr(g.reshape(1).contiguous()), fptr(out[1:2].contiguous()), fptr(d),
                                       stream_ptr()))
        return d, None, None


def get_mel_loss(mel_predictions, mel_targets, mel_masks_fill):
    """model/loss.py:229-242: masked_fill(pad, 0) on both, L1 weighted by non-zero target rows.
    mel_masks_fill: bool [B, L], True = pad."""
    return _MelL1Fn.apply(mel_predictions, mel_targets, mel_masks_fill)
