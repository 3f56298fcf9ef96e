"""FFT blocks of the coarse-mel decoder used by `--model aux|shallow` (transformer/Models.py:103-171,
Layers.py:11-30,67-137, SubLayers.py:8-93, Modules.py:6-25): same class names, constructor
arguments and state_dict keys; eval-mode forward on the HIP path.

Internally everything runs channel-major [B, C, L] (the conv kernels' layout): Q/K/V are one
k=1 GEMM with 768 output rows, attention is the streaming-softmax MFMA kernel, the k=9 FFN conv
and the PostNet convs are the generic conv kernel (eval-mode BatchNorm folded into weights/bias).

Train mode (aux pre-training): dropout p = decoder_dropout after the attention `fc`
and after the FFN (fused in front of the LayerNorm kernel), attention with the probabilities kept for the
backward (batched MFMA GEMMs), PostNet with BatchNorm batch statistics + tanh + dropout 0.5, and the backward
of every piece in the HIP library (autograd.py only chains them).  Dropout keep-masks come from `dropout_fn`
(module attribute; default: `torch.rand(shape) >= p` on the device) so tests can inject the reference's masks.
"""
import math

import numpy as np
import torch
from torch import nn

from . import ops, _lib, autograd as ag
from .blocks import _ConvParams


def default_dropout_fn(shape, p, device):
    """uint8 keep-mask (1 = keep) for nn.Dropout(p) / F.dropout(p)."""
    return (torch.rand(shape, device=device) >= p).to(torch.uint8)


DROPOUT_FN = None    # module-wide override of the keep-mask source (tests replay the reference's masks through it)


def _keep_mask(module, shape, p, device):
    if p <= 0.0:
        return None, 1.0
    fn = getattr(module, "dropout_fn", None) or DROPOUT_FN or default_dropout_fn
    return fn(shape, p, device).to(device=device, dtype=torch.uint8).contiguous(), 1.0 / (1.0 - p)


def get_sinusoid_encoding_table(n_position, d_hid, padding_idx=None):
    """transformer/Models.py:10-30 (float64 numpy, cast to fp32)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    tab = pos / np.power(10000, 2 * (j // 2) / d_hid)
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    if padding_idx is not None:
        tab[padding_idx] = 0.0
    return torch.FloatTensor(tab)


class _Linear(nn.Module):
    """nn.Linear parameter holder (weight [out,in], bias [out]) with nn.Linear's default init."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)



class MultiHeadAttention(nn.Module):
    """transformer/SubLayers.py:8-57."""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1):
        super().__init__()
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.w_qs = _Linear(d_model, n_head * d_k)
        self.w_ks = _Linear(d_model, n_head * d_k)
        self.w_vs = _Linear(d_model, n_head * d_v)
        self.layer_norm = nn.LayerNorm(d_model)      # parameter holder; evaluated by mg_layernorm_cm_fwd
        self.fc = _Linear(n_head * d_v, d_model)
        self.dropout_p = dropout
        # eval-mode contraction precision: "fp32" (exact, default) or "f16" (fp16 MFMA operands, fp32 accumulate:
        # BASELINE configs[4]'s long-form path; set through Decoder.set_attention_precision)
        self.precision = "fp32"

    def forward_cm(self, x, pad8, fill=False):
        """x [B, D, L] channel-major, pad8 uint8 [B, L] -> LayerNorm(fc(attn) + x), [B, D, L].
        fill=True also zeroes padded frames (the masked_fill of Layers.py:25)."""
        if self.training:
            return self._forward_cm_train(x, pad8, fill)
        wq = torch.cat([ops.pack_cached(w.weight[:, :, None]) for w in (self.w_qs, self.w_ks, self.w_vs)])
        bq = torch.cat([self.w_qs.bias, self.w_ks.bias, self.w_vs.bias]).detach()
        D = self.w_qs.weight.shape[1]
        qkv = ops.conv1d_packed(x, wq, bq, 3 * self.n_head * self.d_k, 1)
        att = ops.attention(qkv, pad8, self.n_head, self.d_k, self.precision)
        y = ops.conv1d_packed(att, ops.pack_cached(self.fc.weight[:, :, None]), self.fc.bias.detach(), D, 1)
        return ops.layernorm_cm(y, x, self.layer_norm.weight.detach(), self.layer_norm.bias.detach(),
                                pad8 if fill else None, self.layer_norm.eps)


    def _forward_cm_train(self, x, pad8, fill):
        w = torch.cat([self.w_qs.weight, self.w_ks.weight, self.w_vs.weight])[:, :, None]
        b = torch.cat([self.w_qs.bias, self.w_ks.bias, self.w_vs.bias])
        qkv = ag.conv1d(x, w, b)
        att = ag.attention_train(qkv, pad8, self.n_head, self.d_k)
        y = ag.conv1d(att, self.fc.weight[:, :, None], self.fc.bias)
        keep, scale = _keep_mask(self, tuple(y.shape), self.dropout_p, y.device)
        return ag.layernorm_train(y, x, self.layer_norm.weight, self.layer_norm.bias, pad8 if fill else None, keep,
                                  scale, self.layer_norm.eps)


class PositionwiseFeedForward(nn.Module):
    """transformer/SubLayers.py:60-93: Conv1d(k=9) -> ReLU -> Conv1d(k=1) -> +residual -> LayerNorm."""

    def __init__(self, d_in, d_hid, kernel_size, dropout=0.1):
        super().__init__()
        self.kernel_size = kernel_size
        self.w_1 = _ConvParams(d_in, d_hid, kernel_size)
        self.w_2 = _ConvParams(d_hid, d_in, 1)
        self.layer_norm = nn.LayerNorm(d_in)
        self.dropout_p = dropout

    def forward_cm(self, x, pad8):
        k = self.kernel_size
        if self.training:
            h = ag.conv1d(x, self.w_1.weight, self.w_1.bias, 1, (k - 1) // 2, "relu")
            y = ag.conv1d(h, self.w_2.weight, self.w_2.bias)
            keep, scale = _keep_mask(self, tuple(y.shape), self.dropout_p, y.device)
            return ag.layernorm_train(y, x, self.layer_norm.weight, self.layer_norm.bias, pad8, keep, scale,
                                      self.layer_norm.eps)
        h = ops.conv1d_packed(x, ops.pack_cached(self.w_1.weight), self.w_1.bias.detach(), self.w_1.weight.shape[0], k,
                              1, (k - 1) // 2, "relu")
        y = ops.conv1d_packed(h, ops.pack_cached(self.w_2.weight), self.w_2.bias.detach(), self.w_2.weight.shape[0], 1)
        return ops.layernorm_cm(y, x, self.layer_norm.weight.detach(), self.layer_norm.bias.detach(), pad8,
                                self.layer_norm.eps)


class FFTBlock(nn.Module):
    """transformer/Layers.py:11-30."""

    def __init__(self, d_model, n_head, d_k, d_v, d_inner, kernel_size, dropout=0.1):
        super().__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner, kernel_size, dropout=dropout)

    def forward_cm(self, x, pad8):
        y = self.slf_attn.forward_cm(x, pad8, fill=True)      # Layers.py:25: masked_fill(pad, 0)
        return self.pos_ffn.forward_cm(y, pad8)                # Layers.py:28: and again after the FFN

    def forward(self, enc_input, mask=None, slf_attn_mask=None):
        """enc_input [B, L, D]; mask bool [B, L] True = pad.  Returns (output [B,L,D], None)."""
        if not enc_input.is_cuda:
            raise _lib.MixganHipError("FFTBlock.forward on %s: the HIP path has no CPU fallback" % enc_input.device)
        pad8 = mask.to(torch.uint8).contiguous() if mask is not None else None
        if self.training:
            return ag.transpose_to_blm(self.forward_cm(ag.transpose_to_bml(enc_input), pad8)), None
        with torch.no_grad():
            x = ops.transpose_bml(enc_input.detach().contiguous(), False)
            y = self.forward_cm(x, pad8)
            return ops.transpose_bml(y, True), None


class Decoder(nn.Module):
    """transformer/Models.py:103-171."""

    def __init__(self, config):
        super().__init__()
        tc = config["transformer"]
        n_position = config["max_seq_len"] + 1
        d_model = tc["decoder_hidden"]
        d_k = d_v = tc["decoder_hidden"] // tc["decoder_head"]
        self.max_seq_len = config["max_seq_len"]
        self.d_model = d_model
        self.position_enc = nn.Parameter(get_sinusoid_encoding_table(n_position, d_model).unsqueeze(0),
                                         requires_grad=False)
        self.layer_stack = nn.ModuleList([
            FFTBlock(d_model, tc["decoder_head"], d_k, d_v, tc["conv_filter_size"], tc["conv_kernel_size"],
                     dropout=tc["decoder_dropout"]) for _ in range(tc["decoder_layer"])])

    def set_attention_precision(self, precision):
        """"fp32" | "f16" for the eval-mode attention of every layer (training always runs fp32)."""
        if precision not in ("fp32", "f16"):
            raise ValueError("attention precision must be 'fp32' or 'f16', got %r" % (precision,))
        for layer in self.layer_stack:
            layer.slf_attn.precision = precision

    def forward_cm(self, enc_seq, mask):
        """enc_seq [B, L, D], mask bool [B, L] True = pad -> channel-major [B, D, L'] and the (possibly
        truncated) pad mask."""
        if not enc_seq.is_cuda:
            raise _lib.MixganHipError("Decoder.forward on %s: the HIP path has no CPU fallback" % enc_seq.device)
        B, L = enc_seq.shape[0], enc_seq.shape[1]
        if self.training:          # Models.py:153-162: clip to max_seq_len, stored table
            L = min(L, self.max_seq_len)
            x = enc_seq[:, :L, :] + self.position_enc[:, :L, :]
            pad8 = mask[:, :L].to(torch.uint8).contiguous()
            y = ag.transpose_to_bml(x)
            for layer in self.layer_stack:
                y = layer.forward_cm(y, pad8)
            return y, pad8
        if L > self.max_seq_len:   # eval: table rebuilt on the fly (Models.py:145-152)
            pos = get_sinusoid_encoding_table(L, self.d_model)[:L, :].unsqueeze(0).to(enc_seq.device)
            x = enc_seq + pos
        else:
            x = enc_seq[:, :L, :] + self.position_enc[:, :L, :]
        pad8 = mask[:, :x.shape[1]].to(torch.uint8).contiguous()
        y = ops.transpose_bml(x.detach().contiguous(), False)
        for layer in self.layer_stack:
            y = layer.forward_cm(y, pad8)
        return y, pad8

    def forward(self, enc_seq, mask, return_attns=False):
        if self.training:
            return ag.transpose_to_blm(self.forward_cm(enc_seq, mask)[0])
        with torch.no_grad():
            y, _ = self.forward_cm(enc_seq, mask)
            return ops.transpose_bml(y, True)


class _PostNetConv(nn.Module):
    """`ConvNorm` of transformer/Layers.py:33-64: parameters under `.conv.{weight,bias}`."""

    def __init__(self, cin, cout, k, gain):
        super().__init__()
        self.conv = _ConvParams(cin, cout, k)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(gain))


class PostNet(nn.Module):
    """transformer/Layers.py:67-137: five Conv1d(k=5) + BatchNorm1d, tanh on all but the last."""

    def __init__(self, n_mel_channels=80, postnet_embedding_dim=512, postnet_kernel_size=5, postnet_n_convolutions=5):
        super().__init__()
        self.kernel_size = postnet_kernel_size
        dims = [n_mel_channels] + [postnet_embedding_dim] * (postnet_n_convolutions - 1) + [n_mel_channels]
        self.convolutions = nn.ModuleList()
        for i in range(postnet_n_convolutions):
            gain = "tanh" if i < postnet_n_convolutions - 1 else "linear"
            self.convolutions.append(nn.Sequential(_PostNetConv(dims[i], dims[i + 1], postnet_kernel_size, gain),
                                                   nn.BatchNorm1d(dims[i + 1])))

    def _folded(self, i):
        """Eval-mode BatchNorm folded into the conv: W' = W * s, b' = (b - mean) * s + beta, s = gamma / sqrt(var + eps).
        Cached on the conv weight until any of the six tensors changes."""
        conv, bn = self.convolutions[i][0].conv, self.convolutions[i][1]
        key = tuple(t._version for t in (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var))
        hit = conv.__dict__.get("_mg_folded")
        if hit is not None and hit[0] == key and hit[1].device == conv.weight.device:
            return hit[1], hit[2]
        with torch.no_grad():
            s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            w = (conv.weight * s[:, None, None]).contiguous()
            b = ((conv.bias - bn.running_mean) * s + bn.bias).contiguous()
            wp = ops.pack_conv_weight(w)
        conv.__dict__["_mg_folded"] = (key, wp, b)
        return wp, b

    bn_group = None      # set to a torch.distributed process group for cross-rank BatchNorm statistics
    dropout_p = 0.5      # F.dropout(..., 0.5, self.training), transformer/Layers.py:133-134

    def _forward_cm_train(self, x):
        n = len(self.convolutions)
        k = self.kernel_size
        for i in range(n):
            conv, bn = self.convolutions[i][0].conv, self.convolutions[i][1]
            h = ag.conv1d(x, conv.weight, conv.bias, 1, (k - 1) // 2)
            keep, scale = _keep_mask(self, tuple(h.shape), self.dropout_p, h.device)
            x, mean, var = ag.batchnorm_act(h, bn.weight, bn.bias, keep, scale, "tanh" if i < n - 1 else None, bn.eps,
                                            self.bn_group)
            with torch.no_grad():    # running statistics as nn.BatchNorm1d (momentum 0.1, unbiased variance)
                cnt = h.shape[0] * h.shape[2] * (1 if self.bn_group is None
                                                 else torch.distributed.get_world_size(self.bn_group))
                m = bn.momentum
                bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                bn.running_var.mul_(1 - m).add_(var * (cnt / max(cnt - 1, 1)), alpha=m)
                bn.num_batches_tracked += 1
        return x

    def forward_cm(self, x):
        """x [B, M, L] channel-major -> [B, M, L]."""
        if self.training:
            return self._forward_cm_train(x)
        n = len(self.convolutions)
        k = self.kernel_size
        for i in range(n):
            wp, b = self._folded(i)
            co = self.convolutions[i][0].conv.weight.shape[0]
            x = ops.conv1d_packed(x, wp, b, co, k, 1, (k - 1) // 2, "tanh" if i < n - 1 else None)
        return x

    def forward(self, x):
        """x [B, L, M] -> [B, L, M] (transformer/Layers.py:129-137)."""
        if not x.is_cuda:
            raise _lib.MixganHipError("PostNet.forward on %s: the HIP path has no CPU fallback" % x.device)
        if self.training:
            return ag.transpose_to_blm(self.forward_cm(ag.transpose_to_bml(x)))
        with torch.no_grad():
            return ops.transpose_bml(self.forward_cm(ops.transpose_bml(x.detach().contiguous(), False)), True)
