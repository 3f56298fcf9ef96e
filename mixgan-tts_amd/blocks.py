"""Parameter-holding mirrors of the reference primitives on the hot path.

Same class names, constructor arguments, initialisation and state_dict keys as
model/blocks.py (`ConvNorm` :326-371, `LinearNorm` :278-291, `DiffusionEmbedding` :899-913,
`Mish` :894-896, `ResidualBlock` :1133-1176) so reference checkpoints load unchanged
(SURVEY.md section 5).  The stored parameter layout is the reference's; the MFMA-fragment
packing the kernels consume is a derived cache owned by the calling module.

Compute goes through the HIP library only.  The fused Denoiser path does not call these
modules' forward(); the stand-alone forward()s below exist for drop-in use of a single layer
and route through the same kernels (ResidualBlock: the fused layer kernel itself).
"""
import math

import torch
from torch import nn

from . import ops


class _ConvParams(nn.Module):
    """`weight` [Co, Ci, K] + `bias` [Co] with nn.Conv1d's default initialisation."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(in_channels * kernel_size)
            nn.init.uniform_(self.bias, -bound, bound)


class _LinearParams(nn.Module):
    def __init__(self, in_features, out_features, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None


class ConvNorm(nn.Module):
    """1D convolution (model/blocks.py:326-371); parameters under `.conv.{weight,bias}`."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1, bias=True,
                 w_init_gain=None, channel_last=False):
        super().__init__()
        if dilation != 1:
            raise NotImplementedError("the hot path only uses dilation=1 (model/blocks.py:1139-1146)")
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.conv = _ConvParams(in_channels, out_channels, kernel_size, bias)
        if w_init_gain is not None:
            nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(w_init_gain))
        self.channel_last = channel_last

    def forward(self, x):
        if self.channel_last:
            x = x.transpose(1, 2)
        y = ops.conv1d(x.contiguous(), self.conv.weight, self.conv.bias, self.stride, self.padding)
        return y.transpose(1, 2) if self.channel_last else y


class LinearNorm(nn.Module):
    """Bias-free Linear with xavier init (model/blocks.py:278-291); parameter `.linear.weight`."""

    def __init__(self, in_features, out_features, bias=False):
        super().__init__()
        self.linear = _LinearParams(in_features, out_features, bias)
        nn.init.xavier_uniform_(self.linear.weight)

    def forward(self, x):
        return ops.linear(x, self.linear.weight, self.linear.bias)


class Mish(nn.Module):
    """x * tanh(softplus(x)) (model/blocks.py:894-896).  Inside the Denoiser it is evaluated by the step-MLP kernel;
    called on its own it runs mg_mish_fwd / mg_mish_bwd."""

    def forward(self, x):
        from . import autograd as ag
        return ag.mish(x)


class DiffusionEmbedding(nn.Module):
    """Sinusoidal step embedding (model/blocks.py:899-913).  Holds no parameters; the frequency
    table is computed here exactly as the reference computes it and handed to the kernels."""

    def __init__(self, d_denoiser):
        super().__init__()
        self.dim = d_denoiser

    def frequencies(self, device=None):
        """exp(-i ln(1e4) / (half - 1)) of the reference's fp32 products, CORRECTLY ROUNDED to fp32 (exp taken in
        fp64).  The reference's own `torch.exp` is a ~1-ulp vectorised routine whose last bit depends on the CPU
        model; one ulp of a frequency is 6e-5 rad at t = 999, so a host-computed table makes T=1000 results differ
        between machines.  This table is the same everywhere and within that ambiguity of any reference run."""
        half = self.dim // 2
        emb = math.log(10000) / (half - 1)
        return torch.exp((torch.arange(half) * -emb).double()).float().to(device)

    def forward(self, x):
        """x: diffusion steps [B] (int64 in the reference's callers) -> [B, dim] = cat(sin, cos) (mg_step_embed)."""
        if not x.is_cuda:
            from . import _lib
            raise _lib.MixganHipError("DiffusionEmbedding.forward on %s: the HIP path has no CPU fallback" % x.device)
        return ops.step_embed(x.to(torch.int64).contiguous(), self.frequencies(x.device).contiguous())


class ResidualBlock(nn.Module):
    """Parameter holder of one gated residual block (model/blocks.py:1133-1176).
    Registration order matches the reference so optimizer state indices line up."""

    def __init__(self, d_encoder, residual_channels, dropout, multi_speaker=True):
        super().__init__()
        self.multi_speaker = multi_speaker
        self.conv_layer = ConvNorm(residual_channels, 2 * residual_channels, kernel_size=3, stride=1, padding=1)
        self.diffusion_projection = LinearNorm(residual_channels, residual_channels)
        if multi_speaker:
            self.speaker_projection = LinearNorm(d_encoder, residual_channels)
        self.conditioner_projection = ConvNorm(d_encoder, residual_channels, kernel_size=1)
        self.output_projection = ConvNorm(residual_channels, 2 * residual_channels, kernel_size=1)

    def forward(self, x, conditioner, diffusion_step, speaker_emb, mask=None):
        """x [B,C,L], conditioner [B,H,L], diffusion_step [B,C], speaker_emb [B,H]|None -> ((x + residual)/sqrt2, skip)
        (model/blocks.py:1157-1176).  The Denoiser runs all its layers through mg_denoiser_fwd; this stand-alone form
        launches the same fused layer kernel once, with autograd."""
        from . import autograd as ag
        d = ag.linear_small(diffusion_step, self.diffusion_projection.linear.weight)
        hvec = d + ag.linear_small(speaker_emb, self.speaker_projection.linear.weight) if self.multi_speaker else d
        return ag.residual_block(x, conditioner, hvec, d,
                                 self.conditioner_projection.conv.weight, self.conditioner_projection.conv.bias,
                                 self.conv_layer.conv.weight, self.conv_layer.conv.bias,
                                 self.output_projection.conv.weight, self.output_projection.conv.bias)
