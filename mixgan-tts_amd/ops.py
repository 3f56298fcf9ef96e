"""Thin tensor-level wrappers over the C ABI (include/mixgan_hip.h).  Forward only here;
differentiable entry points live next to the modules that own the parameters."""
import ctypes
import os

import torch

from . import _lib
from ._lib import fptr, iptr, check, stream_ptr

ACT = {None: 0, "none": 0, "relu": 1, "lrelu": 2, "tanh": 3, "lrelu_s": 4}
PACK_PLAIN, PACK_GATE, PACK_DGRAD = 0, 1, 2


def pack_conv_weight(w, mode=PACK_PLAIN, out=None):
    """[Co, Ci, K] (or [Co, Ci]) fp32 -> MFMA-fragment-ordered cache tensor (into `out` when it fits)."""
    L = _lib.lib()
    if w.dim() == 2:
        w = w[:, :, None]
    w = w.detach().contiguous()
    Co, Ci, K = w.shape
    n = L.mg_conv_packed_floats(Co, Ci, K, mode)
    if n == 0:
        raise _lib.MixganHipError("unsupported conv weight shape %s" % (tuple(w.shape),))
    if out is None or out.numel() != n or out.device != w.device:
        out = torch.empty(n, device=w.device, dtype=torch.float32)
    check(L.mg_conv_pack(fptr(w), fptr(out), Co, Ci, K, mode, stream_ptr()))
    return out


def pack_cached(w, mode=PACK_PLAIN):
    """pack_conv_weight with the result cached ON the owning parameter object (so its lifetime is the
    parameter's: an address-keyed cache goes stale when the allocator reuses freed storage).  The
    packed form is a derived cache, rebuilt when the parameter's version changes (optimizer step,
    load_state_dict)."""
    owner = w._base if w._base is not None else w
    cache = owner.__dict__.setdefault("_mg_pack", {})
    key = (mode, tuple(w.shape), owner.data_ptr())
    hit = cache.get(key)
    if hit is not None and hit[0] == owner._version:
        return hit[1]
    # refreshed IN PLACE when the buffer exists: launches captured in a hipGraph keep reading the same address
    p = pack_conv_weight(w, mode, out=None if hit is None else hit[1])
    cache[key] = (owner._version, p)
    return p


PACK_TPOSE = "tpose"   # polyphase ConvTranspose1d pack (pack_conv_transpose_weight)


def pack_conv_transpose_weight(w):
    """ConvTranspose1d weight [Ci, Co, 2u] (stride u, padding u/2) -> polyphase MFMA pack."""
    L = _lib.lib()
    w = w.detach().contiguous()
    Ci, Co, K = w.shape
    u = K // 2
    n = L.mg_conv_transpose_packed_floats(Ci, Co, u) if K == 2 * u else 0
    if n == 0:
        raise _lib.MixganHipError("unsupported transposed-conv weight shape %s" % (tuple(w.shape),))
    out = torch.empty(n, device=w.device, dtype=torch.float32)
    check(L.mg_conv_transpose_pack(fptr(w), fptr(out), Ci, Co, u, stream_ptr()))
    return out


def conv_transpose1d_packed(x, packed, bias, Co, u, in_slope=1.0, alpha=1.0):
    """alpha * conv_transpose1d(leaky_relu(x, in_slope), w, stride=u, padding=u/2) + bias; x [B,Ci,L] -> [B,Co,u*L]."""
    L = _lib.lib()
    B, Ci, Lin = x.shape
    out = torch.empty(B, Co, u * Lin, device=x.device, dtype=torch.float32)
    check(L.mg_conv_transpose1d_fwd(fptr(x), fptr(packed), fptr(bias, True), fptr(out), B, Ci, Lin, Co, u,
                                    float(in_slope), float(alpha), stream_ptr()))
    return out


_SPLIT_SCRATCH = {}
SPLIT_SCRATCH_FLOATS = 12 << 20


def split_scratch(dev):
    """The split-reduction scratch of mg_conv1d_fwd_split for the current stream.  One per (device, stream): launches
    on one stream are ordered, launches on different streams must not share partial tiles."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    buf = _SPLIT_SCRATCH.get(key)
    if buf is None:
        buf = _SPLIT_SCRATCH[key] = torch.empty(SPLIT_SCRATCH_FLOATS, device=dev, dtype=torch.float32)
    return buf


def conv1d_packed(x, packed, bias, Co, K, stride=1, padding=0, act=None, alpha=1.0, add=None, in_vec=None,
                  out=None, accumulate=False, Lout=None, dilation=1, in_slope=1.0, act_slope=0.0, split=False):
    """act: None | "relu" | "lrelu" (0.2) | "tanh" | "lrelu_s" (slope act_slope); in_slope: leaky ReLU
    applied to the input while staging (1 = identity).  split=True: let the library split a deep reduction over
    workgroups when the output is only a few tiles (mg_conv1d_fwd_split; same result up to summation order)."""
    L = _lib.lib()
    B, Ci, Lin = x.shape
    if Lout is None:
        Lout = (Lin + 2 * padding - dilation * (K - 1) - 1) // stride + 1
    if out is None:
        out = torch.empty(B, Co, Lout, device=x.device, dtype=torch.float32)
    scratch = None
    if split and os.environ.get("MG_CONV_SPLIT", "1") != "0" and not torch.cuda.is_current_stream_capturing():
        scratch = split_scratch(x.device)
    check(L.mg_conv1d_fwd_split(fptr(x), fptr(in_vec, True), fptr(packed), fptr(bias, True), fptr(add, True), fptr(out),
                                B, Ci, Lin, Co, Lout, K, stride, padding, dilation, float(in_slope), ACT[act],
                                float(act_slope), float(alpha), int(accumulate), fptr(scratch, True),
                                0 if scratch is None else scratch.numel(), stream_ptr()))
    return out


def conv1d(x, weight, bias=None, stride=1, padding=0, act=None):
    Co, _, K = weight.shape
    return conv1d_packed(x, pack_conv_weight(weight), None if bias is None else bias.detach(), Co, K, stride,
                         padding, act)


def linear(x, weight, bias=None):
    """x [..., in] -> [..., out] through the k=1 conv kernel (frames = flattened leading dims)."""
    shp = x.shape
    x2 = x.reshape(1, -1, shp[-1]).transpose(1, 2).contiguous()
    y = conv1d(x2, weight[:, :, None], bias)
    return y.transpose(1, 2).reshape(*shp[:-1], weight.shape[0])


def diffuse(mel, t, noise, keep, buf):
    """mg_diffuse_fwd: mel [B,L,M], noise [B,M,L] -> x_t [B,M,L]."""
    L = _lib.lib()
    B, Lf, M = mel.shape
    out = torch.empty(B, M, Lf, device=mel.device, dtype=torch.float32)
    check(L.mg_diffuse_fwd(fptr(mel), iptr(t, torch.int64), fptr(noise), iptr(keep, torch.uint8, True),
                           fptr(buf["spec_min"]), fptr(buf["spec_max"]), fptr(buf["sqrt_alphas_cumprod"]),
                           fptr(buf["sqrt_one_minus_alphas_cumprod"]), fptr(out), B, Lf, M,
                           buf["sqrt_alphas_cumprod"].numel(), stream_ptr()))
    return out


def posterior_sample(x0, x_t, t, noise, keep, buf, clip=True, want_x0c=False, out=None):
    L = _lib.lib()
    B, M, Lf = x_t.shape
    if out is None:
        out = torch.empty_like(x_t)
    x0c = torch.empty_like(x_t) if want_x0c else None
    check(L.mg_posterior_sample_fwd(fptr(x0), fptr(x_t), iptr(t, torch.int64), fptr(noise),
                                    iptr(keep, torch.uint8, True), fptr(buf["posterior_mean_coef1"]),
                                    fptr(buf["posterior_mean_coef2"]), fptr(buf["posterior_log_variance_clipped"]),
                                    fptr(out), fptr(x0c, True), int(clip), B, Lf, M,
                                    buf["posterior_mean_coef1"].numel(), stream_ptr()))
    return (out, x0c) if want_x0c else out


def transpose_bml(x, to_blm, mode=0, spec_min=None, spec_max=None, keep=None):
    """to_blm: [B,M,L] -> [B,L,M] (mode 2 = denorm_spec); else [B,L,M] -> [B,M,L] (mode 1 = norm_spec)."""
    L = _lib.lib()
    if to_blm:
        B, M, Lf = x.shape
        out = torch.empty(B, Lf, M, device=x.device, dtype=torch.float32)
    else:
        B, Lf, M = x.shape
        out = torch.empty(B, M, Lf, device=x.device, dtype=torch.float32)
    check(L.mg_transpose_bml(fptr(x), fptr(out), fptr(spec_min, True), fptr(spec_max, True),
                             iptr(keep, torch.uint8, True), int(to_blm), mode, B, Lf, M, stream_ptr()))
    return out


def posterior_sample_bwd(x0, t, keep, coef1, g_x0c, g_xpp, clip):
    L = _lib.lib()
    B, M, Lf = x0.shape
    out = torch.empty_like(x0)
    check(L.mg_posterior_sample_bwd(fptr(x0), iptr(t, torch.int64), iptr(keep, torch.uint8, True), fptr(coef1),
                                    fptr(g_x0c, True), fptr(g_xpp, True), fptr(out), int(clip), B, Lf, M,
                                    coef1.numel(), stream_ptr()))
    return out


def spec_affine(x, spec_min, spec_max, mode):
    L = _lib.lib()
    M = x.shape[-1]
    out = torch.empty_like(x)
    check(L.mg_spec_affine(fptr(x), fptr(out), fptr(spec_min.reshape(-1)), fptr(spec_max.reshape(-1)), mode,
                           x.numel(), M, stream_ptr()))
    return out


def conv1d_wgrad(dy, x, K, stride=1, padding=0, x_vec=None, alpha=1.0):
    """dW [Co, Ci, K] of conv1d(x, W): dy [B,Co,Ldy], x [B,Ci,Lx]."""
    L = _lib.lib()
    B, Co, Ldy = dy.shape
    _, Ci, Lx = x.shape
    dw = torch.empty(Co, Ci, K, device=x.device, dtype=torch.float32)
    scratch = torch.empty(L.mg_conv1d_wgrad_scratch_floats(Co, Ci, K), device=x.device, dtype=torch.float32)
    check(L.mg_conv1d_wgrad(fptr(dy), fptr(x), fptr(x_vec, True), fptr(dw), fptr(scratch), B, Co, Ci, Ldy, Lx, K,
                            stride, padding, float(alpha), 0, stream_ptr()))
    return dw


def rowsum(x, per_batch=False, alpha=1.0):
    """x [B,R,L] -> [R] (sum over b,l) or [B,R] (sum over l)."""
    L = _lib.lib()
    B, R, Lf = x.shape
    out = torch.empty((B, R) if per_batch else (R,), device=x.device, dtype=torch.float32)
    check(L.mg_rowsum(fptr(x), 0, B, R, Lf, fptr(None if per_batch else out, True), fptr(out if per_batch else None, True),
                      float(alpha), 0, stream_ptr()))
    return out


def act_bwd(dy, y, act):
    L = _lib.lib()
    out = torch.empty_like(dy)
    check(L.mg_act_bwd(fptr(dy), fptr(y), fptr(out), ACT[act], dy.numel(), stream_ptr()))
    return out


def upsample_zero(x, stride, Lup, slope=1.0):
    L = _lib.lib()
    B, C, Lin = x.shape
    out = torch.empty(B, C, Lup, device=x.device, dtype=torch.float32)
    check(L.mg_upsample_zero_act(fptr(x), fptr(out), B * C, Lin, stride, Lup, float(slope), stream_ptr()))
    return out


def cat_transpose(a, b):
    """torch.cat([a, b], -1).transpose(1, 2) for a, b [B,L,M] -> [B,2M,L] (model/mixgantts.py:262-264)."""
    L = _lib.lib()
    B, Lf, M = a.shape
    out = torch.empty(B, 2 * M, Lf, device=a.device, dtype=torch.float32)
    for i, src in enumerate((a, b)):
        dst = ctypes.c_void_p(out.data_ptr() + i * M * Lf * 4)
        check(L.mg_transpose_bml_strided(fptr(src), dst, None, None, None, 0, 0, B, Lf, M, 2 * M * Lf, stream_ptr()))
    return out


def split_transpose(g, M):
    """inverse of cat_transpose for gradients: g [B,2M,L] -> two [B,L,M]."""
    L = _lib.lib()
    B, M2, Lf = g.shape
    outs = []
    for i in range(2):
        o = torch.empty(B, Lf, M, device=g.device, dtype=torch.float32)
        src = ctypes.c_void_p(g.data_ptr() + i * M * Lf * 4)
        check(L.mg_transpose_bml_strided(src, fptr(o), None, None, None, 1, 0, B, Lf, M, M2 * Lf, stream_ptr()))
        outs.append(o)
    return outs


def step_mlp_fwd(t, freq, W0, W2):
    L = _lib.lib()
    B = t.shape[0]
    D1, D0 = W0.shape
    D2 = W2.shape[0]
    dev = W0.device
    emb, pre, h = (torch.empty(B, d, device=dev) for d in (D0, D1, D1))
    out = torch.empty(B, D2, device=dev)
    check(L.mg_step_mlp_fwd(iptr(t, torch.int64), fptr(freq), fptr(W0), fptr(W2), fptr(emb), fptr(pre), fptr(h),
                            fptr(out), B, D0, D1, D2, stream_ptr()))
    return out, emb, pre, h


def step_mlp_bwd(g, emb, pre, h, W2):
    L = _lib.lib()
    B, D2 = g.shape
    D0, D1 = emb.shape[1], pre.shape[1]
    dW0 = torch.empty(D1, D0, device=g.device)
    dW2 = torch.empty(D2, D1, device=g.device)
    scratch = torch.empty(2 * B * D1, device=g.device)
    check(L.mg_step_mlp_bwd(fptr(g), fptr(emb), fptr(pre), fptr(h), fptr(W2), fptr(dW0), fptr(dW2), fptr(scratch),
                            B, D0, D1, D2, stream_ptr()))
    return dW0, dW2


def linear_small_fwd(x, W):
    L = _lib.lib()
    B, K = x.shape
    N = W.shape[0]
    out = torch.empty(B, N, device=x.device)
    check(L.mg_linear_small_fwd(fptr(x), fptr(W), fptr(out), B, N, K, stream_ptr()))
    return out


def linear_small_bwd(g, x, W, want_dx=True):
    L = _lib.lib()
    B, N = g.shape
    K = x.shape[1]
    dx = torch.empty_like(x) if want_dx else None
    dW = torch.empty_like(W)
    check(L.mg_linear_small_bwd(fptr(g), fptr(x), fptr(W), fptr(dx, True), fptr(dW), B, N, K, stream_ptr()))
    return dx, dW


def attention(qkv, key_pad, n_head, d_head, precision="fp32"):
    """qkv [B, 3*n_head*d_head, L] -> [B, n_head*d_head, L] (mg_attention_fwd; precision "f16": fp16 MFMA operands)."""
    L = _lib.lib()
    B, _, Lf = qkv.shape
    out = torch.empty(B, n_head * d_head, Lf, device=qkv.device, dtype=torch.float32)
    if precision not in ("fp32", "f16"):
        raise ValueError("attention precision must be 'fp32' or 'f16', got %r" % (precision,))
    fn = L.mg_attention_fwd if precision == "fp32" else L.mg_attention_fwd_f16
    check(fn(fptr(qkv), iptr(key_pad, torch.uint8, True), fptr(out), B, Lf, n_head, d_head,
                             float(d_head) ** -0.5, stream_ptr()))
    return out


def layernorm_cm(a, res, gamma, beta, pad, eps=1e-5):
    L = _lib.lib()
    B, C, Lf = a.shape
    out = torch.empty_like(a)
    check(L.mg_layernorm_cm_fwd(fptr(a), fptr(res, True), fptr(gamma), fptr(beta), iptr(pad, torch.uint8, True),
                                fptr(out), B, C, Lf, float(eps), stream_ptr()))
    return out


# ------------------------------------------------------------------ aux pre-training (train_ops.hip, bgemm.hip)
def _u8(t, allow_none=True):
    return iptr(t, torch.uint8, allow_none)


def attention_train_fwd(qkv, key_pad, n_head, d):
    """Train-mode attention keeping the probabilities: qkv [B,3HD,L] -> (out [B,HD,L], P [B*H,L,L])."""
    Lb = _lib.lib()
    B, _, L = qkv.shape
    HD, scale, st = n_head * d, float(d) ** -0.5, stream_ptr()
    P = torch.empty(B * n_head, L, L, device=qkv.device, dtype=torch.float32)
    q, k, v = qkv, qkv[:, HD:], qkv[:, 2 * HD:]
    bs, hs = 3 * HD * L, d * L
    # S[q,k] = sum_d Q[d,q] K[d,k]
    check(Lb.mg_bgemm(ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(k.data_ptr()), fptr(P), L, L, d, B, n_head,
                      1, L, bs, hs, L, 1, bs, hs, L, n_head * L * L, L * L, 1.0, 0, st))
    check(Lb.mg_softmax_rows_fwd(fptr(P), _u8(key_pad), B, n_head, L, scale, st))
    out = torch.empty(B, HD, L, device=qkv.device, dtype=torch.float32)
    # O[d,q] = sum_k V[d,k] P[q,k]
    check(Lb.mg_bgemm(ctypes.c_void_p(v.data_ptr()), fptr(P), fptr(out), d, L, L, B, n_head,
                      L, 1, bs, hs, 1, L, n_head * L * L, L * L, L, HD * L, d * L, 1.0, 0, st))
    return out, P


def attention_train_bwd(qkv, P, d_out, n_head, d):
    """-> d_qkv [B,3HD,L]."""
    Lb = _lib.lib()
    B, _, L = qkv.shape
    HD, scale, st = n_head * d, float(d) ** -0.5, stream_ptr()
    q, k, v = qkv, qkv[:, HD:], qkv[:, 2 * HD:]
    bs, hs = 3 * HD * L, d * L
    pbs, phs = n_head * L * L, L * L
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv, dqkv[:, HD:], dqkv[:, 2 * HD:]
    cp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    # dV[d,k] = sum_q dO[d,q] P[q,k]
    check(Lb.mg_bgemm(fptr(d_out), fptr(P), cp(dv), d, L, L, B, n_head, L, 1, HD * L, d * L, L, 1, pbs, phs,
                      L, bs, hs, 1.0, 0, st))
    # dP[q,k] = sum_d dO[d,q] V[d,k]
    dS = torch.empty_like(P)
    check(Lb.mg_bgemm(fptr(d_out), cp(v), fptr(dS), L, L, d, B, n_head, 1, L, HD * L, d * L, L, 1, bs, hs,
                      L, pbs, phs, 1.0, 0, st))
    check(Lb.mg_softmax_rows_bwd(fptr(P), fptr(dS), B, n_head, L, scale, st))
    # dQ[d,q] = sum_k K[d,k] dS[q,k];  dK[d,k] = sum_q Q[d,q] dS[q,k]
    check(Lb.mg_bgemm(cp(k), fptr(dS), cp(dq), d, L, L, B, n_head, L, 1, bs, hs, 1, L, pbs, phs, L, bs, hs, 1.0, 0, st))
    check(Lb.mg_bgemm(cp(q), fptr(dS), cp(dk), d, L, L, B, n_head, L, 1, bs, hs, L, 1, pbs, phs, L, bs, hs, 1.0, 0, st))
    return dqkv


def layernorm_cm_train(a, keep, drop_scale, res, gamma, beta, pad, eps=1e-5):
    Lb = _lib.lib()
    B, C, L = a.shape
    pre, out = torch.empty_like(a), torch.empty_like(a)
    check(Lb.mg_layernorm_cm_train_fwd(fptr(a), _u8(keep), float(drop_scale), fptr(res, True), fptr(gamma), fptr(beta),
                                       _u8(pad), fptr(pre), fptr(out), B, C, L, float(eps), stream_ptr()))
    return out, pre


def layernorm_cm_bwd(pre, dy, gamma, pad, keep, drop_scale, eps=1e-5):
    """-> (d_pre, d_a, dgamma, dbeta)."""
    Lb = _lib.lib()
    B, C, L = pre.shape
    d_pre = torch.empty_like(pre)
    d_a = torch.empty_like(pre) if keep is not None else None
    part = torch.zeros(2, 32, C, device=pre.device, dtype=torch.float32)     # 32 partial-sum slots per vector
    check(Lb.mg_layernorm_cm_bwd(fptr(pre), fptr(dy), fptr(gamma), _u8(pad), _u8(keep), float(drop_scale), fptr(d_pre),
                                 fptr(d_a, True), fptr(part[0]), fptr(part[1]), B, C, L, float(eps), stream_ptr()))
    dgb = part.sum(1)
    return d_pre, (d_a if d_a is not None else d_pre), dgb[0], dgb[1]


def bn_stats(x):
    Lb = _lib.lib()
    B, C, L = x.shape
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    var = torch.empty(C, device=x.device, dtype=torch.float32)
    check(Lb.mg_bn_stats(fptr(x), fptr(mean), fptr(var), B, C, L, stream_ptr()))
    return mean, var


def bn_act_fwd(x, mean, invstd, gamma, beta, keep, drop_scale, act):
    Lb = _lib.lib()
    B, C, L = x.shape
    y = torch.empty_like(x) if act == "tanh" else None
    out = torch.empty_like(x)
    check(Lb.mg_bn_act_fwd(fptr(x), fptr(mean), fptr(invstd), fptr(gamma), fptr(beta), _u8(keep), float(drop_scale),
                           ACT[act], fptr(y, True), fptr(out), B, C, L, stream_ptr()))
    return out, y


def bn_act_bwd_reduce(dout, keep, drop_scale, y, x, mean, invstd, act):
    Lb = _lib.lib()
    B, C, L = x.shape
    dg = torch.empty(C, device=x.device, dtype=torch.float32)
    db = torch.empty(C, device=x.device, dtype=torch.float32)
    check(Lb.mg_bn_act_bwd_reduce(fptr(dout), _u8(keep), float(drop_scale), fptr(y, True), fptr(x), fptr(mean),
                                  fptr(invstd), ACT[act], fptr(dg), fptr(db), B, C, L, stream_ptr()))
    return dg, db


def bn_act_bwd_apply(dout, keep, drop_scale, y, x, mean, invstd, gamma, dg, db, inv_count, act):
    Lb = _lib.lib()
    B, C, L = x.shape
    dx = torch.empty_like(x)
    check(Lb.mg_bn_act_bwd_apply(fptr(dout), _u8(keep), float(drop_scale), fptr(y, True), fptr(x), fptr(mean),
                                 fptr(invstd), fptr(gamma), fptr(dg), fptr(db), float(inv_count), ACT[act], fptr(dx),
                                 B, C, L, stream_ptr()))
    return dx


def resblock_fwd(x, cond, wc_p, w3_p, wo_p, bc, b3, bo, hvec, dvec, save):
    """mg_resblock_fwd: one fused residual layer.  Returns (x_out, skip, saves) with saves = (h, g, sig, tnh) | None."""
    L = _lib.lib()
    B, C, Lf = x.shape
    H = cond.shape[1]
    x_out, skip = torch.empty_like(x), torch.empty_like(x)
    saves = tuple(torch.empty_like(x) for _ in range(4)) if save else None
    sp = [fptr(t) for t in saves] if save else [fptr(None, True)] * 4
    check(L.mg_resblock_fwd(fptr(x), fptr(cond), fptr(wc_p), fptr(w3_p), fptr(wo_p), fptr(bc), fptr(b3), fptr(bo),
                            fptr(hvec), fptr(dvec), fptr(x_out), fptr(skip), *sp, B, C, H, Lf, stream_ptr()))
    return x_out, skip, saves


def gate_bwd(dg, sig, tnh):
    L = _lib.lib()
    B, C, Lf = dg.shape
    dz = torch.empty(B, 2 * C, Lf, device=dg.device, dtype=torch.float32)
    check(L.mg_gate_bwd(fptr(dg), fptr(sig), fptr(tnh), fptr(dz), B, C, Lf, stream_ptr()))
    return dz


def mish_fwd(x):
    y = torch.empty_like(x)
    check(_lib.lib().mg_mish_fwd(fptr(x), fptr(y), x.numel(), stream_ptr()))
    return y


def mish_bwd(gy, x):
    gx = torch.empty_like(x)
    check(_lib.lib().mg_mish_bwd(fptr(gy), fptr(x), fptr(gx), x.numel(), stream_ptr()))
    return gx


def step_embed(t, freq):
    """DiffusionEmbedding: t int64 [B], freq [D/2] -> [B, D]."""
    B, D = t.shape[0], 2 * freq.shape[0]
    emb = torch.empty(B, D, device=t.device, dtype=torch.float32)
    check(_lib.lib().mg_step_embed(iptr(t, torch.int64), fptr(freq), fptr(emb), B, D, stream_ptr()))
    return emb


def grad_norm(flat, max_norm, out=None, scratch=None):
    """(||flat||_2, clip factor min(1, max_norm / (norm + 1e-6))) as a 2-element device tensor (no host sync)."""
    L = _lib.lib()
    if out is None:
        out = torch.empty(2, device=flat.device, dtype=torch.float32)
    if scratch is None:
        scratch = torch.empty(L.mg_grad_norm_scratch_floats(), device=flat.device, dtype=torch.float32)
    check(L.mg_grad_norm(fptr(flat), flat.numel(), float(max_norm), fptr(scratch), fptr(out), stream_ptr()))
    return out


def adam_flat(p, g, m, v, lr, betas, eps, weight_decay, step, grad_scale=None, hyper=None):
    """One torch.optim.Adam step on flat buffers, in place on p, m, v; g * grad_scale[0] is the gradient used.
    hyper: device pair {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} read by the kernel instead of lr / step (graphs)."""
    L = _lib.lib()
    check(L.mg_adam_flat_dev(fptr(p), fptr(g), fptr(m), fptr(v), p.numel(), float(lr), float(betas[0]), float(betas[1]),
                             float(eps), float(weight_decay), int(step), fptr(grad_scale, True), fptr(hyper, True),
                             stream_ptr()))
