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


_SCRATCH = {}


def _multi_scratch(dev):
    """Zero-initialised once per (device, stream): the kernel keeps its ticket there and re-arms it."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _SCRATCH:
        _SCRATCH[key] = torch.zeros(_lib.lib().mg_multi_loss_scratch_floats(), device=dev, dtype=torch.float32)
    return _SCRATCH[key]


class _WeightedMeansFn(torch.autograd.Function):
    """total = sum_k weight_k * mean_k in ONE launch (mg_multi_loss_fwd), gradients of all terms in one more.
    spec: tuple of (mode, c, weight, group) per term, mode 0 = mean (x - c)^2, mode 1 = mean |x - y| with gradient
    to x only; tensors: the x of every term, then the y of the mode-1 terms in order.
    Returns the [1 + groups + nterms] result vector (total, group subtotals, per-term means); differentiate [0]."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        nt = len(spec)
        if nt < 1 or nt > _lib.MG_LOSS_MAX_TERMS:
            raise ValueError("1..%d terms per call" % _lib.MG_LOSS_MAX_TERMS)
        xs = [t.contiguous() for t in tensors[:nt]]
        ys = iter([t.detach().contiguous() for t in tensors[nt:]])
        targets = [next(ys) if s[0] == 1 else None for s in spec]
        terms = (_lib.LossTerm * nt)()
        for k, (s, x, y) in enumerate(zip(spec, xs, targets)):
            if y is not None and y.shape != x.shape:
                raise ValueError("term %d: shapes %s vs %s" % (k, tuple(x.shape), tuple(y.shape)))
            terms[k] = _lib.LossTerm(fptr(x).value, fptr(y, True).value if y is not None else None, None, x.numel(),
                                     float(s[1]), float(s[2]), int(s[0]), int(s[3]))
        dev = xs[0].device
        out = torch.empty(1 + _lib.MG_LOSS_GROUPS + nt, device=dev, dtype=torch.float32)
        check(_lib.lib().mg_multi_loss_fwd(terms, nt, fptr(_multi_scratch(dev)), fptr(out), stream_ptr()))
        ctx.spec = spec
        ctx.save_for_backward(*xs, *[y for y in targets if y is not None])
        return out

    @staticmethod
    def backward(ctx, g):
        spec, nt = ctx.spec, len(ctx.spec)
        saved = ctx.saved_tensors
        xs, ys = saved[:nt], iter(saved[nt:])
        terms = (_lib.LossTerm * nt)()
        grads = []
        for k, (s, x) in enumerate(zip(spec, xs)):
            y = next(ys) if s[0] == 1 else None
            d = torch.empty_like(x) if ctx.needs_input_grad[1 + k] else None
            grads.append(d)
            terms[k] = _lib.LossTerm(fptr(x).value, fptr(y).value if y is not None else None,
                                     fptr(d).value if d is not None else None, x.numel(), float(s[1]), float(s[2]),
                                     int(s[0]), int(s[3]))
        # only the total (element 0) is a loss; the subtotals and means are read-outs of the same numbers
        check(_lib.lib().mg_multi_loss_bwd(terms, nt, fptr(g[:1].contiguous()), stream_ptr()))
        return (None, *grads, *([None] * (len(saved) - nt)))


def weighted_means(terms):
    """terms: list of ("mse", x, c, weight[, group]) / ("l1", x, y, weight[, group]).  Returns the result vector of
    _WeightedMeansFn: [0] total (differentiable w.r.t. every x), [1:1+4] group subtotals, then per-term means."""
    spec, xs, ys = [], [], []
    for t in terms:
        kind, x, other, w = t[:4]
        grp = t[4] if len(t) > 4 else 0
        if kind == "mse":
            spec.append((0, float(other), float(w), grp))
        elif kind == "l1":
            spec.append((1, 0.0, float(w), grp))
            ys.append(other)
        else:
            raise ValueError(kind)
        xs.append(x)
    return _WeightedMeansFn.apply(tuple(spec), *xs, *ys)


class _RangeMeansFn(torch.autograd.Function):
    """weighted_means over BATCH RANGES of whole feature maps: term k reads rows [a_lo, a_hi) of tensor `ti` (mode 0:
    mean (x - c)^2; mode 1: mean |x[a] - x[b]| against rows [b_lo, b_lo + a_hi - a_lo) of the same tensor, gradient to
    the a-range only).  The trainer runs D(fake) and D(real) as one pass over 2B items; slicing the maps apart for the
    losses made autograd rebuild every map's gradient from two zero-padded halves (a fill, a copy and an add per half
    and map: ~55 launches per step).  Here the maps go in whole, and the backward is one zero-fill of one flat buffer
    plus one launch that writes every term's gradient into its range."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        nt = len(spec)
        if nt < 1 or nt > _lib.MG_LOSS_MAX_TERMS:
            raise ValueError("1..%d terms per call" % _lib.MG_LOSS_MAX_TERMS)
        xs = [t.contiguous() for t in tensors]
        terms = (_lib.LossTerm * nt)()
        for k, (ti, mode, c, w, grp, a_lo, a_hi, b_lo) in enumerate(spec):
            x = xs[ti]
            per = x[0].numel()
            n = (a_hi - a_lo) * per
            if not (0 <= a_lo < a_hi <= x.shape[0]) or (mode == 1 and not (0 <= b_lo and b_lo + a_hi - a_lo <= x.shape[0])):
                raise ValueError("term %d: batch range outside the tensor" % k)
            base = fptr(x).value
            terms[k] = _lib.LossTerm(base + 4 * a_lo * per, (base + 4 * b_lo * per) if mode == 1 else None, None, n,
                                     float(c), float(w), int(mode), int(grp))
        dev = xs[0].device
        out = torch.empty(1 + _lib.MG_LOSS_GROUPS + nt, device=dev, dtype=torch.float32)
        check(_lib.lib().mg_multi_loss_fwd(terms, nt, fptr(_multi_scratch(dev)), fptr(out), stream_ptr()))
        ctx.spec = spec
        ctx.save_for_backward(*xs)
        return out

    @staticmethod
    def backward(ctx, g):
        spec, nt = ctx.spec, len(ctx.spec)
        xs = ctx.saved_tensors
        need = [ctx.needs_input_grad[1 + i] for i in range(len(xs))]
        sizes = [x.numel() if nd else 0 for x, nd in zip(xs, need)]
        flat = torch.zeros(sum(sizes), device=xs[0].device, dtype=torch.float32)      # rows no term writes: zero
        grads, at = [], 0
        for x, nd, n in zip(xs, need, sizes):
            grads.append(flat[at:at + n].view_as(x) if nd else None)
            at += n
        terms = (_lib.LossTerm * nt)()
        for k, (ti, mode, c, w, grp, a_lo, a_hi, b_lo) in enumerate(spec):
            x = xs[ti]
            per = x[0].numel()
            base = fptr(x).value
            da = (fptr(grads[ti]).value + 4 * a_lo * per) if grads[ti] is not None else None
            terms[k] = _lib.LossTerm(base + 4 * a_lo * per, (base + 4 * b_lo * per) if mode == 1 else None, da,
                                     (a_hi - a_lo) * per, float(c), float(w), int(mode), int(grp))
        check(_lib.lib().mg_multi_loss_bwd(terms, nt, fptr(g[:1].contiguous()), stream_ptr()))
        return (None, *grads)


def _range_means(spec, tensors):
    """spec entries (tensor index, mode, c, weight, group, a_lo, a_hi, b_lo); no two terms may write the same rows."""
    return _RangeMeansFn.apply(tuple(spec), *tensors)


def d_loss_total_2b(logit_cond, logit_uncond, B):
    """d_loss_total on the last maps of ONE discriminator pass over [fake (rows 0..B-1); real (rows B..2B-1)]."""
    out = _range_means([(0, 0, 1.0, 0.5, 0, B, 2 * B, 0), (1, 0, 1.0, 0.5, 0, B, 2 * B, 0),
                        (0, 0, 0.0, 0.5, 1, 0, B, 0), (1, 0, 0.0, 0.5, 1, 0, B, 0)], [logit_cond, logit_uncond])
    return out[0], out[1], out[2]


def g_adv_fm_total_2b(cond_maps, uncond_maps, B, lambda_fm, n_layers=5):
    """g_adv_fm_total on the maps of ONE discriminator pass over [fake; real]: LSGAN on the fake rows of the last maps,
    feature matching |fake - real| on the others (gradient to the fake rows only, as model/loss.py:221-227 with the real
    maps as targets)."""
    w = lambda_fm * (4.0 / (n_layers + 1)) * 0.5
    nm = len(cond_maps)
    tensors = list(cond_maps) + list(uncond_maps)
    spec = [(nm - 1, 0, 1.0, 0.5, 0, 0, B, 0), (2 * nm - 1, 0, 1.0, 0.5, 0, 0, B, 0)]
    for j in range(nm - 1):
        spec.append((j, 1, 0.0, w, 1, 0, B, B))
        spec.append((nm + j, 1, 0.0, w, 1, 0, B, B))
    out = _range_means(spec, tensors)
    return out[0], out[1], out[2]


def d_loss_total(r_logit_cond, r_logit_uncond, f_logit_cond, f_logit_uncond):
    """d_real + d_fake of get_lsgan_losses_fn()'s d_loss_fn (model/loss.py:12-30, train.py:142-143) as one fused sum.
    Returns (total, d_real, d_fake); differentiate total."""
    out = weighted_means([("mse", r_logit_cond, 1.0, 0.5, 0), ("mse", r_logit_uncond, 1.0, 0.5, 0),
                          ("mse", f_logit_cond, 0.0, 0.5, 1), ("mse", f_logit_uncond, 0.0, 0.5, 1)])
    return out[0], out[1], out[2]


def g_adv_fm_total(D_real_cond, D_real_uncond, D_fake_cond, D_fake_uncond, lambda_fm, n_layers=5):
    """adv + lambda_fm * fm of the generator loss (train.py:162-170: g_loss_fn on the last maps, get_fm_loss on the
    others) as one fused sum.  Returns (total, adv, lambda_fm * fm); differentiate total."""
    w = lambda_fm * (4.0 / (n_layers + 1)) * 0.5
    terms = [("mse", D_fake_cond[-1], 1.0, 0.5, 0), ("mse", D_fake_uncond[-1], 1.0, 0.5, 0)]
    for j in range(len(D_fake_cond) - 1):
        terms.append(("l1", D_fake_cond[j], D_real_cond[j], w, 1))
        terms.append(("l1", D_fake_uncond[j], D_real_uncond[j], w, 1))
    out = weighted_means(terms)
    return out[0], out[1], out[2]


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
