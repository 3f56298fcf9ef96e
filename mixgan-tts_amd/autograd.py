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
    """Denoiser.forward with the layer activations saved in a workspace that this node keeps alive (a later forward
    of the module at any shape, or a cache eviction, cannot take it away); backward = mg_denoiser_bwd."""

    @staticmethod
    def forward(ctx, module, x, t, cond, spk, pre, *params):
        """pre: None, or (out, ws) of a forward that already ran on exactly these inputs (Denoiser.run_pair): the node
        then only adopts its output and saved activations."""
        _require_cuda(x, cond)
        if pre is None:
            out = module.run(x, t, cond, spk, save=True)
            ctx.ws = module.last_ws
        else:
            out, ctx.ws = pre
        ctx.module = module
        ctx.gen = ctx.ws._mg_gen
        ctx.save_for_backward(x, t, cond, spk if spk is not None else x.new_empty(0))
        ctx.has_spk = spk is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, t, cond, spk = ctx.saved_tensors
        need = ctx.needs_input_grad
        d_x, d_cond, d_spk, pg = ctx.module.run_backward(g.contiguous(), x, t, cond, spk if ctx.has_spk else None,
                                                         ctx.ws, ctx.gen, need[1], need[3], need[4])
        return (None, d_x, None, d_cond, d_spk, None) + tuple(pg)


def denoise_and_posterior(diff, x_t, t, cond_t, spk, post_noise, keep, clip, coarse_mel, pre=None):
    """Training branch of GaussianDiffusion.forward with autograd (model/diffusion.py:210-220)."""
    den = diff.denoise_fn
    x0 = DenoiserFn.apply(den, x_t, t, cond_t, spk, pre, *[p for p in den._weight_table() if p is not None])
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
                              padding, act, in_vec=None if in_vec is None else in_vec.detach().contiguous(), split=True)
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
                                   Lout=Lin, split=True)
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


class _ResBlockFn(torch.autograd.Function):
    """One gated residual block (model/blocks.py:1157-1176) stand-alone: forward = the fused layer kernel the Denoiser
    launches per layer (mg_resblock_fwd), backward = the same data-gradient / weight-gradient GEMMs mg_denoiser_bwd
    uses, one layer's worth.  hvec = Wd s (+ Wp spk) and dvec = Wd s are inputs (their Linear layers are separate
    autograd nodes)."""

    @staticmethod
    def forward(ctx, x, cond, hvec, dvec, Wc, bc, W3, b3, Wo, bo):
        _require_cuda(x, cond)
        x, cond = x.contiguous(), cond.contiguous()
        save = any(ctx.needs_input_grad)
        x_out, skip, saves = ops.resblock_fwd(
            x, cond, ops.pack_cached(Wc), ops.pack_cached(W3, ops.PACK_GATE), ops.pack_cached(Wo), bc.detach(),
            b3.detach(), bo.detach(), hvec.detach().contiguous(), dvec.detach().contiguous(), save)
        if save:
            ctx.save_for_backward(cond, Wc, W3, Wo, *saves)
        return x_out, skip

    @staticmethod
    def backward(ctx, g_x, g_skip):
        cond, Wc, W3, Wo, h, g, sig, tnh = ctx.saved_tensors
        C, H = Wc.shape[0], Wc.shape[1]
        rs2 = 0.70710678118654752440
        g_xs = (g_x * rs2).contiguous()                     # d/d(x + Wd s) through the residual
        dout = torch.cat([g_xs, g_skip.contiguous()], 1)   # gradient of o = Wo g + bo, [B, 2C, L]
        dg = ops.conv1d_packed(dout, ops.pack_cached(Wo, ops.PACK_DGRAD), None, C, 1)
        dz = ops.gate_bwd(dg, sig, tnh)
        dh = ops.conv1d_packed(dz, ops.pack_cached(W3, ops.PACK_DGRAD), None, C, 3, 1, 1)
        need = ctx.needs_input_grad
        dx = ops.conv1d_packed(dz, ops.pack_cached(W3, ops.PACK_DGRAD), None, C, 3, 1, 1, add=g_xs) if need[0] else None
        dcond = ops.conv1d_packed(dh, ops.pack_cached(Wc, ops.PACK_DGRAD), None, H, 1) if need[1] else None
        return (dx, dcond,
                ops.rowsum(dh, per_batch=True) if need[2] else None,
                ops.rowsum(g_xs, per_batch=True) if need[3] else None,
                ops.conv1d_wgrad(dh, cond, 1) if need[4] else None, ops.rowsum(dh) if need[5] else None,
                ops.conv1d_wgrad(dz, h, 3, 1, 1) if need[6] else None, ops.rowsum(dz) if need[7] else None,
                ops.conv1d_wgrad(dout, g, 1) if need[8] else None, ops.rowsum(dout) if need[9] else None)


def residual_block(x, cond, hvec, dvec, Wc, bc, W3, b3, Wo, bo):
    return _ResBlockFn.apply(x, cond, hvec, dvec, Wc, bc, W3, b3, Wo, bo)


class _MishFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_cuda(x)
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.mish_fwd(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.mish_bwd(g.contiguous(), x)


def mish(x):
    return _MishFn.apply(x)


# --------------------------------------------------------------------------------------------------
# aux pre-training: train-mode FFT-block / PostNet pieces with their backward
# --------------------------------------------------------------------------------------------------
class _AttentionTrainFn(torch.autograd.Function):
    """softmax(QK^T / sqrt(d) | key mask) V on channel-major qkv [B, 3HD, L], probabilities kept for the backward."""

    @staticmethod
    def forward(ctx, qkv, key_pad, n_head, d):
        _require_cuda(qkv)
        qkv = qkv.contiguous()
        out, P = ops.attention_train_fwd(qkv, key_pad, n_head, d)
        ctx.save_for_backward(qkv, P)
        ctx.cfg = (n_head, d)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, P = ctx.saved_tensors
        return ops.attention_train_bwd(qkv, P, g.contiguous(), *ctx.cfg), None, None, None


def attention_train(qkv, key_pad, n_head, d):
    return _AttentionTrainFn.apply(qkv, key_pad, n_head, d)


class _LayerNormTrainFn(torch.autograd.Function):
    """out = pad ? 0 : LayerNorm_c(dropout(a) + res) on [B, C, L]; keep: uint8 keep-mask or None."""

    @staticmethod
    def forward(ctx, a, res, gamma, beta, pad, keep, drop_scale, eps):
        _require_cuda(a, res)
        out, pre = ops.layernorm_cm_train(a.contiguous(), keep, drop_scale, res.contiguous(), gamma.detach(),
                                          beta.detach(), pad, eps)
        ctx.save_for_backward(pre, gamma, pad if pad is not None else pre.new_empty(0, dtype=torch.uint8),
                              keep if keep is not None else pre.new_empty(0, dtype=torch.uint8))
        ctx.cfg = (pad is not None, keep is not None, drop_scale, eps)
        return out

    @staticmethod
    def backward(ctx, g):
        pre, gamma, pad, keep = ctx.saved_tensors
        has_pad, has_keep, scale, eps = ctx.cfg
        d_pre, d_a, dg, db = ops.layernorm_cm_bwd(pre, g.contiguous(), gamma.detach(), pad if has_pad else None,
                                                  keep if has_keep else None, scale, eps)
        return d_a, d_pre, dg, db, None, None, None, None


def layernorm_train(a, res, gamma, beta, pad, keep, drop_scale, eps):
    return _LayerNormTrainFn.apply(a, res, gamma, beta, pad, keep, drop_scale, eps)


class _BatchNormActFn(torch.autograd.Function):
    """dropout(act(BatchNorm1d(x))) with batch statistics on [B, C, L].  Returns (out, mean, biased var); with a
    process group the statistics -- and in the backward the two per-channel sums -- are all-reduced (SyncBN)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, keep, drop_scale, act, eps, group):
        _require_cuda(x)
        x = x.contiguous()
        mean, var = ops.bn_stats(x)
        count = x.shape[0] * x.shape[2]
        if group is not None:
            import torch.distributed as dist
            world = dist.get_world_size(group)
            packed = torch.stack([mean, var + mean * mean])          # E[x], E[x^2]; equal counts per rank
            dist.all_reduce(packed, group=group)
            mean = packed[0] / world
            var = (packed[1] / world - mean * mean).clamp_min_(0.0)
            count *= world
        invstd = torch.rsqrt(var + eps)
        out, y = ops.bn_act_fwd(x, mean, invstd, gamma.detach(), beta.detach(), keep, drop_scale, act)
        ctx.save_for_backward(x, mean, invstd, gamma, y if y is not None else x.new_empty(0),
                              keep if keep is not None else x.new_empty(0, dtype=torch.uint8))
        ctx.cfg = (keep is not None, drop_scale, act, count, group)
        ctx.mark_non_differentiable(mean, var)
        return out, mean, var

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        x, mean, invstd, gamma, y, keep = ctx.saved_tensors
        has_keep, scale, act, count, group = ctx.cfg
        g = g.contiguous()
        keep = keep if has_keep else None
        y = y if act == "tanh" else None
        dg, db = ops.bn_act_bwd_reduce(g, keep, scale, y, x, mean, invstd, act)
        dg_all, db_all = dg, db
        if group is not None:
            import torch.distributed as dist
            packed = torch.stack([dg, db])
            dist.all_reduce(packed, group=group)
            dg_all, db_all = packed[0], packed[1]
        dx = ops.bn_act_bwd_apply(g, keep, scale, y, x, mean, invstd, gamma.detach(), dg_all, db_all, 1.0 / count, act)
        return dx, dg, db, None, None, None, None, None


def batchnorm_act(x, gamma, beta, keep, drop_scale, act, eps, group=None):
    return _BatchNormActFn.apply(x, gamma, beta, keep, drop_scale, act, eps, group)
