"""JCUDiscriminator (model/mixgantts.py:186-288): same constructor, state_dict keys,
initialisation and forward contract as the reference; every convolution, the step MLP and the
layout changes run in the HIP library (forward and backward, through mixgan_tts_amd.autograd)."""
import os

import torch
from torch import nn

from . import autograd as A
from .blocks import ConvNorm, LinearNorm, DiffusionEmbedding, Mish


class JCUDiscriminator(nn.Module):
    """Joint conditional / unconditional discriminator -- drop-in for model.mixgantts.JCUDiscriminator."""

    def __init__(self, preprocess_config, model_config, train_config):
        super().__init__()
        n_mel_channels = preprocess_config["preprocessing"]["mel"]["n_mel_channels"]
        residual_channels = model_config["denoiser"]["residual_channels"]
        dc = model_config["discriminator"]
        n_layer, n_uncond_layer, n_cond_layer = dc["n_layer"], dc["n_uncond_layer"], dc["n_cond_layer"]
        n_channels, kernel_sizes, strides = dc["n_channels"], dc["kernel_sizes"], dc["strides"]
        self.multi_speaker = model_config["multi_speaker"]

        self.input_projection = LinearNorm(2 * n_mel_channels, 2 * n_mel_channels)
        self.diffusion_embedding = DiffusionEmbedding(residual_channels)
        self.mlp = nn.Sequential(
            LinearNorm(residual_channels, residual_channels * 4),
            Mish(),
            LinearNorm(residual_channels * 4, n_channels[n_layer - 1]),
        )
        if self.multi_speaker:
            self.spk_mlp = nn.Sequential(LinearNorm(residual_channels, n_channels[n_layer - 1]))

        def conv(i):
            return ConvNorm(n_channels[i - 1] if i != 0 else 2 * n_mel_channels, n_channels[i],
                            kernel_size=kernel_sizes[i], stride=strides[i], dilation=1)

        self.conv_block = nn.ModuleList([conv(i) for i in range(n_layer)])
        self.uncond_conv_block = nn.ModuleList([conv(i) for i in range(n_layer, n_layer + n_uncond_layer)])
        self.cond_conv_block = nn.ModuleList([conv(i) for i in range(n_layer, n_layer + n_cond_layer)])
        self.apply(self.weights_init)
        self._freq = None

    def weights_init(self, m):
        # model/mixgantts.py:251-254: every ConvNorm weight ~ N(0, 0.02)
        if m.__class__.__name__.find("ConvNorm") != -1:
            m.conv.weight.data.normal_(0.0, 0.02)

    @staticmethod
    def _lrelu_conv(layer, x, in_vec=None):
        return A.conv1d(x, layer.conv.weight, layer.conv.bias, layer.stride, layer.padding, "lrelu", in_vec)

    def forward(self, x_ts, x_t_prevs, s, t):
        """x_ts, x_t_prevs [B,T,M]; s [B,H]|None; t int64 [B] -> (cond_feats[5], uncond_feats[5])."""
        dev = x_ts.device
        if self._freq is None or self._freq.device != dev:
            self._freq = self.diffusion_embedding.frequencies(dev).contiguous()
        # Linear over the concatenated mel pair, as a k=1 conv on the channel-major layout
        x = A.cat_transpose(x_t_prevs, x_ts)                                   # [B, 2M, T]
        x = A.conv1d(x, self.input_projection.linear.weight[:, :, None], None)
        step = A.step_mlp(t, self._freq, self.mlp[0].linear.weight, self.mlp[2].linear.weight)   # [B, 512]
        if self.multi_speaker:
            step = step + A.linear_small(s, self.spk_mlp[0].linear.weight)
        cond_feats, uncond_feats = [], []
        for layer in self.conv_block:
            x = self._lrelu_conv(layer, x)
            cond_feats.append(x)
            uncond_feats.append(x)
        x_cond, x_uncond = x, x
        # The two tails are independent and small (512 -> 128 -> 1 channels at L/4 frames: 128 workgroups per launch at
        # B=16, L=1000, half the CUs): the unconditional one runs on a side stream next to the conditional one.  Autograd
        # replays each backward on the stream its forward ran on, so the two backward chains overlap as well.
        overlap = (self.branch_overlap and os.environ.get("MG_JCU_OVERLAP", "1") != "0" and x.is_cuda
                   and not torch.cuda.is_current_stream_capturing())
        if overlap:
            main = torch.cuda.current_stream(dev)
            if self._side is None or self._side.device != dev:
                self._side = torch.cuda.Stream(device=dev)
            side = self._side
            side.wait_stream(main)
            x.record_stream(side)
            with torch.cuda.stream(side):
                for layer in self.uncond_conv_block:
                    x_uncond = self._lrelu_conv(layer, x_uncond)
                    x_uncond.record_stream(main)          # consumed (losses, backward) on the main stream
                    uncond_feats.append(x_uncond)
        for i, layer in enumerate(self.cond_conv_block):
            x_cond = self._lrelu_conv(layer, x_cond, step if i == 0 else None)    # (x + step[:, :, None]) fused
            cond_feats.append(x_cond)
        if overlap:
            main.wait_stream(side)
        else:
            for layer in self.uncond_conv_block:
                x_uncond = self._lrelu_conv(layer, x_uncond)
                uncond_feats.append(x_uncond)
        return cond_feats, uncond_feats

    branch_overlap = True     # False: both tails on the caller's stream, one after the other
    _side = None
