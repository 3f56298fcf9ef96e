"""MixGANTTS.forward orchestration (model/mixgantts.py:16-183) around the HIP path.

The linguistic encoder is upstream of the path and out of scope: it is
injected (`linguistic_encoder=`, any module with the reference's call signature,
model/linguistic_encoder.py:238-380) and keeps running as stock PyTorch-ROCm.  Everything
downstream of it -- FFT decoder / mel_linear / PostNet for aux|shallow, GaussianDiffusion -- is
ours.  The 16-slot output list, `p_targets` and `coarse_mels` are laid out exactly as the reference
returns them (consumed positionally by train.py / evaluate.py / synthesize.py / model/loss.py).
"""
import json
import os

import torch
from torch import nn

from . import ops
from .diffusion import GaussianDiffusion
from .transformer import Decoder, PostNet, _Linear


def get_mask_from_lengths(lengths, max_len=None):
    """utils/tools.py:144-153 -- NOTE: returns True = VALID (the reference negates its own mask)."""
    batch_size = lengths.shape[0]
    if max_len is None:
        max_len = torch.max(lengths).item()
    ids = torch.arange(0, max_len, device=lengths.device).unsqueeze(0).expand(batch_size, -1)
    return ~(ids >= lengths.unsqueeze(1).expand(-1, max_len))


class MixGANTTS(nn.Module):
    def __init__(self, args, preprocess_config, model_config, train_config, linguistic_encoder=None):
        super().__init__()
        self.model = args.model
        self.model_config = model_config
        if linguistic_encoder is not None:
            self.linguistic_encoder = linguistic_encoder
        if self.model in ["aux", "shallow"]:
            self.decoder = Decoder(model_config)
            self.mel_linear = _Linear(model_config["transformer"]["decoder_hidden"],
                                      preprocess_config["preprocessing"]["mel"]["n_mel_channels"])
            self.postnet = PostNet()
        self.diffusion = GaussianDiffusion(args, preprocess_config, model_config, train_config)
        self.speaker_emb = None
        if model_config["multi_speaker"]:
            self.embedder_type = preprocess_config["preprocessing"]["speaker_embedder"]
            if self.embedder_type == "none":
                with open(os.path.join(preprocess_config["path"]["preprocessed_path"], "speakers.json")) as f:
                    n_speaker = len(json.load(f))
                self.speaker_emb = nn.Embedding(n_speaker, model_config["transformer"]["encoder_hidden"])
            else:
                self.speaker_emb = nn.Linear(model_config["external_speaker_dim"],
                                             model_config["transformer"]["encoder_hidden"])

    def _detach(self, p):
        return p.detach() if p is not None and self.model == "shallow" else p

    def coarse_mel(self, cond, mel_pad_mask):
        """Decoder -> mel_linear -> PostNet residual (model/mixgantts.py:140-143), channel-major inside.
        In train mode with grad enabled (aux pre-training) every step is a differentiable HIP Function."""
        if self.training and torch.is_grad_enabled():
            from . import autograd as ag
            y, _ = self.decoder.forward_cm(cond, mel_pad_mask)
            m = ag.conv1d(y, self.mel_linear.weight[:, :, None], self.mel_linear.bias)
            return ag.transpose_to_blm(self.postnet.forward_cm(m) + m)
        y, pad8 = self.decoder.forward_cm(cond, mel_pad_mask)
        M = self.mel_linear.weight.shape[0]
        m = ops.conv1d_packed(y, ops.pack_cached(self.mel_linear.weight[:, :, None]), self.mel_linear.bias.detach(), M, 1)
        out = self.postnet.forward_cm(m) + m
        return ops.transpose_bml(out, True)

    def forward(self, speakers, texts, src_lens, max_src_len, word_boundaries, src_w_lens, max_src_w_len,
                speak_embeds=None, attn_priors=None, mels=None, mel_lens=None, max_mel_len=None, p_targets=None,
                e_targets=None, d_targets=None, spker_embeds=None, p_control=1.0, e_control=1.0, d_control=1.0):
        if not hasattr(self, "linguistic_encoder"):
            raise RuntimeError("MixGANTTS needs a linguistic_encoder (out of scope of the HIP path; pass the "
                               "reference's model.linguistic_encoder.LinguisticEncoder instance)")
        src_masks = get_mask_from_lengths(src_lens, max_src_len)
        src_w_masks = get_mask_from_lengths(src_w_lens, max_src_w_len)
        mel_masks = get_mask_from_lengths(mel_lens, max_mel_len) if mel_lens is not None else None
        (output, p_predictions, e_predictions, log_d_predictions, d_rounded, mel_lens, mel_masks, alignments,
         alignment_logprobs) = self.linguistic_encoder(
            texts, src_lens, word_boundaries, src_masks, src_w_lens, src_w_masks, mel_masks, max_mel_len, attn_priors,
            p_targets, e_targets, d_targets, p_control, d_control)
        speaker_emb = None
        if self.speaker_emb is not None:
            if self.embedder_type == "none":
                speaker_emb = self.speaker_emb(speakers)
            else:
                assert spker_embeds is not None, "Speaker embedding should not be None"
                speaker_emb = self.speaker_emb(spker_embeds)
        mel_masks = ~mel_masks                       # now True = pad (model/mixgantts.py:123,138)
        x_ts = x_t_prevs = x_t_prev_preds = diffusion_step = None
        coarse_mels = postnet_outputs = None
        if self.model == "naive":
            output, x_ts, x_t_prevs, x_t_prev_preds, diffusion_step = self.diffusion(mels, output, speaker_emb, mel_masks)
        elif self.model in ["aux", "shallow"]:
            cond = output.clone()
            if self.training and torch.is_grad_enabled():
                # differentiable in aux AND shallow training: the reference detaches only what it hands to the
                # diffusion and its third return value (:155-160,180); slot 15 keeps the decoder's graph, and
                # postnet_loss = L1(slot 15, mel) trains decoder / PostNet / encoder in shallow too (model/loss.py:165-167)
                coarse_mels = self.coarse_mel(output, mel_masks)
            else:
                with torch.no_grad():
                    coarse_mels = self.coarse_mel(output, mel_masks)
            postnet_outputs = coarse_mels
            if self.model == "aux":
                output = self.diffusion.diffuse_trace(coarse_mels, mel_masks)
            else:
                output, x_ts, x_t_prevs, x_t_prev_preds, diffusion_step = self.diffusion(
                    mels, self._detach(cond), self._detach(speaker_emb), self._detach(mel_masks),
                    self._detach(coarse_mels))
        else:
            raise NotImplementedError
        return [
            output, (x_ts, x_t_prevs, x_t_prev_preds), self._detach(speaker_emb), diffusion_step, p_predictions,
            self._detach(e_predictions), log_d_predictions, self._detach(d_rounded), self._detach(src_masks),
            self._detach(mel_masks), self._detach(src_lens), self._detach(mel_lens), alignments, alignment_logprobs,
            src_w_masks, postnet_outputs,
        ], p_targets, self._detach(coarse_mels)
