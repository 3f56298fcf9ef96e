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
