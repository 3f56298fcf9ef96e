"""HiFi-GAN V1 generator (hifigan/models.py:112-173) on the HIP conv kernel -- SURVEY.md section 8 f3,
the step right after the path (mel [B, 80, L] -> waveform [B, 1, 256 L]; `utils/model.py:108-126`).

Same constructor (`Generator(h)` with the attribute-style config of hifigan/config.json), parameter
names (`conv_pre`, `ups.{i}`, `resblocks.{j}.convs{1,2}.{m}`, `conv_post`, each with `weight_g` /
`weight_v` / `bias`, or `weight` / `bias` after `remove_weight_norm()`), and forward as the
reference, so its checkpoints load unchanged (none ship with the reference: `.MISSING_LARGE_BLOBS`).

Kernel mapping: every Conv1d (k in {3,7,11}, dilation in {1,3,5}) is the MFMA implicit-GEMM conv with
the pre-activation leaky ReLU applied while the input tile is staged and the post-activation / residual
add in the epilogue; ConvTranspose1d (k = 2*stride) is a polyphase 3-tap GEMM over the input (leaky ReLU
fused in staging, phases interleaved by the epilogue; other (k, stride) fall back to zero insertion plus a
stride-1 convolution on the transposed, tap-flipped pack); the three ResBlocks of a stage accumulate into
one buffer and the 1/3 is folded into the next convolution (leaky ReLU is positively homogeneous).
Inference only (the vocoder is not trained on this path).
"""
import torch
from torch import nn

from . import ops, _lib

LRELU_SLOPE = 0.1


class _WNConv(nn.Module):
    """Parameter holder of weight_norm(Conv1d / ConvTranspose1d): weight_g [C0,1,1], weight_v [C0,C1,K], bias."""

    def __init__(self, shape, bias_n, std=None):
        super().__init__()
        v = torch.empty(*shape)
        if std is None:
            nn.init.kaiming_uniform_(v, a=5 ** 0.5)
        else:
            v.normal_(0.0, std)           # init_weights (hifigan/models.py:10-13)
        self.weight_v = nn.Parameter(v)
        self.weight_g = nn.Parameter(v.detach().flatten(1).norm(dim=1).view(-1, 1, 1).clone())
        self.bias = nn.Parameter(torch.zeros(bias_n).uniform_(-0.05, 0.05))

    def effective_weight(self):
        if "weight" in self._parameters:
            return self.weight
        v = self.weight_v
        return v * (self.weight_g / v.flatten(1).norm(dim=1).view(-1, 1, 1))

    def remove_weight_norm(self):
        if "weight" not in self._parameters:
            w = self.effective_weight().detach().clone()
            del self.weight_g, self.weight_v
            self.weight = nn.Parameter(w)

    def packed(self, mode):
        """MFMA pack of the effective weight, cached until a parameter changes."""
        ps = [p for p in self.parameters()]
        key = (mode,) + tuple((p.data_ptr(), p._version) for p in ps)
        hit = self.__dict__.get("_mg_pack")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                w = self.effective_weight().contiguous()
                hit = (key, ops.pack_conv_transpose_weight(w) if mode == ops.PACK_TPOSE else ops.pack_conv_weight(w, mode))
            self.__dict__["_mg_pack"] = hit
        return hit[1]


class ResBlock(nn.Module):
    """hifigan/models.py:20-109."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.h, self.kernel_size, self.dilation = h, kernel_size, tuple(dilation)
        self.convs1 = nn.ModuleList([_WNConv((channels, channels, kernel_size), channels, 0.01) for _ in dilation])
        self.convs2 = nn.ModuleList([_WNConv((channels, channels, kernel_size), channels, 0.01) for _ in dilation])

    def remove_weight_norm(self):
        for l in list(self.convs1) + list(self.convs2):
            l.remove_weight_norm()

    def forward_cm(self, x, acc, first):
        """x [B,C,L] -> adds this block's output into `acc` (= on `first`)."""
        k = self.kernel_size
        C = x.shape[1]
        r = x
        n = len(self.dilation)
        for m, d in enumerate(self.dilation):
            c1, c2 = self.convs1[m], self.convs2[m]
            t = ops.conv1d_packed(r, c1.packed(ops.PACK_PLAIN), c1.bias.detach(), C, k, 1, (k * d - d) // 2,
                                  act="lrelu_s", act_slope=LRELU_SLOPE, dilation=d, in_slope=LRELU_SLOPE)
            if m < n - 1:
                r = ops.conv1d_packed(t, c2.packed(ops.PACK_PLAIN), c2.bias.detach(), C, k, 1, (k - 1) // 2, add=r)
            else:   # last pair: write (conv + bias + r) straight into the stage accumulator
                ops.conv1d_packed(t, c2.packed(ops.PACK_PLAIN), c2.bias.detach(), C, k, 1, (k - 1) // 2, add=r,
                                  out=acc, accumulate=not first)
        return acc


class Generator(nn.Module):
    """hifigan/models.py:112-173."""

    def __init__(self, h):
        super().__init__()
        self.h = h
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        c0 = h.upsample_initial_channel
        self.conv_pre = _WNConv((c0, 80, 7), c0)
        self.ups = nn.ModuleList([
            _WNConv((c0 // (2 ** i), c0 // (2 ** (i + 1)), k), c0 // (2 ** (i + 1)), 0.01)
            for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes))])
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
                self.resblocks.append(ResBlock(h, ch, k, d))
        self.conv_post = _WNConv((1, ch, 7), 1, 0.01)

    def remove_weight_norm(self):
        for l in self.ups:
            l.remove_weight_norm()
        for l in self.resblocks:
            l.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()

    @torch.no_grad()
    def forward(self, x):
        """x: mel [B, 80, L] -> [B, 1, L * prod(upsample_rates)]."""
        if not x.is_cuda:
            raise _lib.MixganHipError("hifigan.Generator on %s: the HIP path has no CPU fallback" % x.device)
        h = self.h
        x = x.contiguous()
        c0 = h.upsample_initial_channel
        x = ops.conv1d_packed(x, self.conv_pre.packed(ops.PACK_PLAIN), self.conv_pre.bias.detach(), c0, 7, 1, 3)
        scale = 1.0
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            up = self.ups[i]
            co = c0 // (2 ** (i + 1))
            if k == 2 * u and u in (2, 4, 8):     # polyphase: 3-tap GEMM with co*u rows, lrelu fused in staging
                x = ops.conv_transpose1d_packed(x, up.packed(ops.PACK_TPOSE), up.bias.detach(), co, u,
                                                in_slope=LRELU_SLOPE, alpha=scale)
            else:                                 # general (k, u): lrelu + zero insertion, flipped stride-1 conv
                L = x.shape[2]
                pad = (k - u) // 2
                z = ops.upsample_zero(x, u, (L - 1) * u + 1, slope=LRELU_SLOPE)
                x = ops.conv1d_packed(z, up.packed(ops.PACK_DGRAD), up.bias.detach(), co, k, 1, k - 1 - pad,
                                      alpha=scale, Lout=(L - 1) * u - 2 * pad + k)
            acc = torch.empty_like(x)
            for j in range(self.num_kernels):
                self.resblocks[i * self.num_kernels + j].forward_cm(x, acc, j == 0)
            x = acc                       # = num_kernels * (xs / num_kernels); the division rides on the next conv
            scale = 1.0 / self.num_kernels
        # F.leaky_relu(x) with the DEFAULT slope 0.01 (hifigan/models.py:161), conv_post, tanh
        return ops.conv1d_packed(x, self.conv_post.packed(ops.PACK_PLAIN), self.conv_post.bias.detach(), 1, 7, 1, 3,
                                 act="tanh", alpha=scale, in_slope=0.01)


class AttrDict(dict):
    """hifigan/__init__.py / env.py: json config with attribute access."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


def get_vocoder(config, device, config_path="hifigan/config.json", checkpoint_path=None):
    """utils/model.py:74-105 for `vocoder.model == "HiFi-GAN"`: build from hifigan/config.json, load the
    `generator` state dict of hifigan/generator_<speaker>.pth.tar, eval, remove_weight_norm, move to the device.
    (MelGAN comes from torch.hub -- a download -- and is not mirrored.)"""
    import json
    name = config["vocoder"]["model"]
    speaker = config["vocoder"]["speaker"]
    if name != "HiFi-GAN":
        raise NotImplementedError("vocoder %r: only HiFi-GAN is on the HIP path" % name)
    with open(config_path, "r") as f:
        h = AttrDict(json.load(f))
    vocoder = Generator(h)
    if checkpoint_path is None:
        checkpoint_path = "hifigan/generator_%s.pth.tar" % speaker
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    vocoder.load_state_dict(ckpt["generator"])
    vocoder.eval()
    vocoder.remove_weight_norm()
    vocoder.to(device)
    return vocoder


def vocoder_infer(mels, vocoder, model_config, preprocess_config, lengths=None):
    """utils/model.py:108-126: mels [B, 80, L] -> list of int16 numpy waveforms (cropped to `lengths`).  The scaling
    and the int16 conversion run on the device, so the host copy is half the bytes of the reference's."""
    if model_config["vocoder"]["model"] != "HiFi-GAN":
        raise NotImplementedError("only HiFi-GAN is on the HIP path")
    with torch.no_grad():
        wavs = vocoder(mels).squeeze(1)
        scaled = wavs * float(preprocess_config["preprocessing"]["audio"]["max_wav_value"])
        wavs = scaled.to(torch.int32).to(torch.int16).cpu().numpy()     # truncation toward zero, as numpy's astype
    wavs = [w for w in wavs]
    for i in range(len(mels)):
        if lengths is not None:
            wavs[i] = wavs[i][: lengths[i]]
    return wavs
