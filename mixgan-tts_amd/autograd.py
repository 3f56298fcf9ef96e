"""torch.autograd.Function wrappers: forward and backward both run in the HIP library."""
import torch

from . import ops, _lib


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.MixganHipError("tensor on %s: the HIP path has no CPU fallback" % t.device)


class _TransposeFn(torch.autograd.Function):
    """[B,L,C] <-> [B,C,L] through mg_transpose_bml; the gradient is the opposite transpose."""

    @staticmethod
    def forward(ctx, x, to_blm):
        ctx.to_blm = to_blm
        return ops.transpose_bml(x.contiguous(), to_blm)

    @staticmethod
    def backward(ctx, g):
        return ops.transpose_bml(g.contiguous(), not ctx.to_blm), None


def transpose_to_bml(x):
    _require_cuda(x)
    return _TransposeFn.apply(x, False)


def transpose_to_blm(x):
    _require_cuda(x)
    return _TransposeFn.apply(x, True)


class _SpecAffineFn(torch.autograd.Function):
    """norm_spec / denorm_spec (model/diffusion.py:228-232) on [..., M] tensors."""

    @staticmethod
    def forward(ctx, x, spec_min, spec_max, norm):
        ctx.save_for_backward(spec_min, spec_max)
        ctx.norm = norm
        return ops.spec_affine(x.contiguous(), spec_min, spec_max, 1 if norm else 2)

    @staticmethod
    def backward(ctx, g):
        spec_min, spec_max = ctx.saved_tensors
        return ops.spec_affine(g.contiguous(), spec_min, spec_max, 3 if ctx.norm else 4), None, None, None


def spec_affine(x, spec_min, spec_max, norm):
    _require_cuda(x)
    return _SpecAffineFn.apply(x, spec_min, spec_max, norm)


class DenoiserFn(torch.autograd.Function):
    """Denoiser.forward with activations saved in the module's workspace; backward = mg_denoiser_bwd."""

    @staticmethod
    def forward(ctx, module, x, t, cond, spk, *params):
        _require_cuda(x, cond)
        out = module.run(x, t, cond, spk, save=True)
        ctx.module = module
        ctx.gen = module._save_gen
        ctx.save_for_backward(x, t, cond, spk if spk is not None else x.new_empty(0))
        ctx.has_spk = spk is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, t, cond, spk = ctx.saved_tensors
        need = ctx.needs_input_grad
        d_x, d_cond, d_spk, pg = ctx.module.run_backward(g.contiguous(), x, t, cond, spk if ctx.has_spk else None,
                                                         ctx.gen, need[1], need[3], need[4])
        return (None, d_x, None, d_cond, d_spk) + tuple(pg)


def denoise_and_posterior(diff, x_t, t, cond_t, spk, post_noise, keep, clip, coarse_mel):
    """Training branch of GaussianDiffusion.forward with autograd (model/diffusion.py:210-220)."""
    den = diff.denoise_fn
    x0 = DenoiserFn.apply(den, x_t, t, cond_t, spk, *[p for p in den._weight_table() if p is not None])
    buf = diff._buf()
    if coarse_mel is None:
        x0c, xpp = _PosteriorFn.apply(x0, x_t, t, post_noise, keep, buf["posterior_mean_coef1"],
                                      buf["posterior_mean_coef2"], buf["posterior_log_variance_clipped"], clip, True)
        return x0c, xpp
    # shallow: x_{t-1} is sampled around the detached coarse mel; only x0c carries gradient
    start = ops.transpose_bml(coarse_mel.detach().contiguous(), False, 1, diff.spec_min, diff.spec_max)
    xpp = ops.posterior_sample(start, x_t, t, post_noise, keep, buf, clip=False)
    x0c, _ = _PosteriorFn.apply(x0, x_t, t, post_noise, keep, buf["posterior_mean_coef1"],
                                buf["posterior_mean_coef2"], buf["posterior_log_variance_clipped"], clip, False)
    return x0c, xpp


class _PosteriorFn(torch.autograd.Function):
    """(x0c, x_{t-1}) = mg_posterior_sample_fwd(x0, ...); d/dx0 only (everything else is data)."""

    @staticmethod
    def forward(ctx, x0, x_t, t, noise, keep, c1, c2, lv, clip, through_posterior):
        buf = {"posterior_mean_coef1": c1, "posterior_mean_coef2": c2, "posterior_log_variance_clipped": lv}
        xpp, x0c = ops.posterior_sample(x0.contiguous(), x_t, t, noise, keep, buf, clip=clip, want_x0c=True)
        ctx.save_for_backward(x0, t, keep, c1)
        ctx.clip = clip
        ctx.through = through_posterior
        ctx.mark_non_differentiable(xpp) if not through_posterior else None
        return x0c, xpp

    @staticmethod
    def backward(ctx, g_x0c, g_xpp):
        x0, t, keep, c1 = ctx.saved_tensors
        g = ops.posterior_sample_bwd(x0, t, keep, c1, g_x0c.contiguous(),
                                     g_xpp.contiguous() if ctx.through else None, ctx.clip)
        return (g,) + (None,) * 9


# --------------------------------------------------------------------------------------------------
# differentiable building blocks for the JCU discriminator and the FFT blocks: every forward and
# backward below is a call into the HIP library; torch.autograd only chains them.
# --------------------------------------------------------------------------------------------------
class _Conv1dFn(torch.autograd.Function):
    """y = act(conv1d(x (+ in_vec), W, b)); x [B,Ci,L] channel-major."""

    @staticmethod
    def forward(ctx, x, weight, bias, in_vec, stride, padding, act):
        _require_cuda(x)
        x = x.contiguous()
        Co, Ci, K = weight.shape
        y = ops.conv1d_packed(x, ops.pack_cached(weight), None if bias is None else bias.detach(), Co, K, stride,
                              padding, act, in_vec=None if in_vec is None else in_vec.detach().contiguous())
        ctx.save_for_backward(x, weight, y if act else None, in_vec)
        ctx.cfg = (stride, padding, act, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y, in_vec = ctx.saved_tensors
        stride, padding, act, has_bias = ctx.cfg
        Co, Ci, K = weight.shape
        gy = gy.contiguous()
        dpre = ops.act_bwd(gy, y, act) if act else gy
        need = ctx.needs_input_grad
        vec = None if in_vec is None else in_vec.detach().contiguous()
        dw = ops.conv1d_wgrad(dpre, x, K, stride, padding, x_vec=vec) if need[1] else None
        db = ops.rowsum(dpre) if (has_bias and need[2]) else None
        dx = dvec = None
        if need[0] or (in_vec is not None and need[3]):
            Lin = x.shape[2]
            src = dpre
            if stride > 1:
                src = ops.upsample_zero(dpre, stride, (dpre.shape[2] - 1) * stride + 1)
            dx = ops.conv1d_packed(src, ops.pack_cached(weight, ops.PACK_DGRAD), None, Ci, K, 1, K - 1 - padding,
                                   Lout=Lin)
            if in_vec is not None and need[3]:
                dvec = ops.rowsum(dx, per_batch=True)
        return dx if need[0] else None, dw, db, dvec, None, None, None


def conv1d(x, weight, bias=None, stride=1, padding=0, act=None, in_vec=None):
    return _Conv1dFn.apply(x, weight, bias, in_vec, stride, padding, act)


class _StepMlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, freq, W0, W2):
        out, emb, pre, h = ops.step_mlp_fwd(t.contiguous(), freq, W0.detach().contiguous(), W2.detach().contiguous())
        ctx.save_for_backward(emb, pre, h, W2)
        return out

    @staticmethod
    def backward(ctx, g):
        emb, pre, h, W2 = ctx.saved_tensors
        dW0, dW2 = ops.step_mlp_bwd(g.contiguous(), emb, pre, h, W2.detach().contiguous())
        return None, None, dW0, dW2


def step_mlp(t, freq, W0, W2):
    _require_cuda(W0)
    return _StepMlpFn.apply(t, freq, W0, W2)


class _LinearSmallFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)
        return ops.linear_small_fwd(x.contiguous(), W.detach().contiguous())

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        dx, dW = ops.linear_small_bwd(g.contiguous(), x.contiguous(), W.detach().contiguous(), ctx.needs_input_grad[0])
        return dx, dW


def linear_small(x, W):
    _require_cuda(x)
    return _LinearSmallFn.apply(x, W)


class _CatTransposeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.M = a.shape[-1]
        return ops.cat_transpose(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, g):
        da, db = ops.split_transpose(g.contiguous(), ctx.M)
        return da, db


def cat_transpose(a, b):
    _require_cuda(a, b)
    return _CatTransposeFn.apply(a, b)
