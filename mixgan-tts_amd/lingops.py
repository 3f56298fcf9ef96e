"""Device-side drop-ins for the four host-loop functions of the reference's LinguisticEncoder
(SURVEY.md section 8 f1).  Same names, arguments and results as

    utils.tools.word_level_pooling                      (utils/tools.py:394-413)
    model.linguistic_encoder.LengthRegulator            (model/linguistic_encoder.py:383-416)
    LinguisticEncoder.get_mapping_mask / get_rel_coef   (model/linguistic_encoder.py:185-199, 222-236)

but without a Python iteration (and a device->host .item()) per phoneme: one kernel launch each.
The only host synchronisation left is the one the output SHAPE needs when the caller does not
supply it (`max_len=None` at inference; the widest word count of the batch for pooling).
"""
import torch
from torch import nn

from . import _lib
from ._lib import fptr, iptr, check, stream_ptr


def _i64(t):
    return t.to(torch.int64).contiguous()


class _WordPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src_seq, wb, src_w_len, w_out, mean):
        src_seq = src_seq.contiguous()
        B, Tp, H = src_seq.shape
        Tw = wb.shape[1]
        out = torch.empty(B, w_out, H, device=src_seq.device, dtype=torch.float32)
        check(_lib.lib().mg_word_pool_fwd(fptr(src_seq), iptr(wb, torch.int64), iptr(src_w_len, torch.int64), fptr(out),
                                          B, Tp, Tw, H, w_out, int(mean), stream_ptr()))
        ctx.save_for_backward(wb, src_w_len)
        ctx.dims = (B, Tp, Tw, H, w_out, int(mean))
        return out

    @staticmethod
    def backward(ctx, g):
        wb, src_w_len = ctx.saved_tensors
        B, Tp, Tw, H, w_out, mean = ctx.dims
        d = torch.zeros(B, Tp, H, device=g.device, dtype=torch.float32)
        check(_lib.lib().mg_word_pool_bwd(fptr(g.contiguous()), iptr(wb, torch.int64), iptr(src_w_len, torch.int64),
                                          fptr(d), B, Tp, Tw, H, w_out, mean, stream_ptr()))
        return d, None, None, None, None


def word_level_pooling(src_seq, src_len, wb, src_w_len, reduce="sum", max_words=None):
    """utils/tools.py:394-413.  `max_words` (= the batch's largest src_w_len, the reference's
    max_src_w_len) avoids the one host sync this function would otherwise need for its output shape."""
    if reduce not in ("sum", "mean"):
        raise ValueError()
    if not src_seq.is_cuda:
        raise _lib.MixganHipError("word_level_pooling on %s: the HIP path has no CPU fallback" % src_seq.device)
    w_out = int(max_words) if max_words is not None else int(src_w_len.max().item())
    w_out = min(w_out, wb.shape[1]) if w_out > 0 else 1
    return _WordPoolFn.apply(src_seq, _i64(wb), _i64(src_w_len), w_out, reduce == "mean")


class _LengthRegulateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, duration, l_max):
        x = x.contiguous()
        B, Tw, H = x.shape
        out = torch.empty(B, l_max, H, device=x.device, dtype=torch.float32)
        mel_len = torch.empty(B, device=x.device, dtype=torch.int64)
        check(_lib.lib().mg_length_regulate_fwd(fptr(x), iptr(duration, torch.int64), fptr(out), iptr(mel_len, torch.int64),
                                                B, Tw, H, l_max, stream_ptr()))
        ctx.save_for_backward(duration)
        ctx.dims = (B, Tw, H, l_max)
        ctx.mark_non_differentiable(mel_len)
        return out, mel_len

    @staticmethod
    def backward(ctx, g, _):
        (duration,) = ctx.saved_tensors
        B, Tw, H, l_max = ctx.dims
        dx = torch.empty(B, Tw, H, device=g.device, dtype=torch.float32)
        check(_lib.lib().mg_length_regulate_bwd(fptr(g.contiguous()), iptr(duration, torch.int64), fptr(dx), B, Tw, H,
                                                l_max, stream_ptr()))
        return dx, None, None


class LengthRegulator(nn.Module):
    """model/linguistic_encoder.py:383-416: forward(x, duration, max_len) -> (output, mel_len)."""

    def forward(self, x, duration, max_len):
        if not x.is_cuda:
            raise _lib.MixganHipError("LengthRegulator on %s: the HIP path has no CPU fallback" % x.device)
        dur = _i64(duration)
        if max_len:
            l_max = int(max_len)
        else:   # the output length is data dependent: one host sync for the whole batch
            l_max = max(1, int(dur.clamp(min=0).sum(1).max().item()))
        return _LengthRegulateFn.apply(x, dur, l_max)


def get_mapping_mask(q, kv, dur_w, wb, src_w_len):
    """model/linguistic_encoder.py:185-199 -> bool [B, q_len, kv_len]."""
    B, Lq, Lkv = q.shape[0], q.shape[1], kv.shape[1]
    out = torch.empty(B, Lq, Lkv, device=kv.device, dtype=torch.uint8)
    check(_lib.lib().mg_mapping_mask(iptr(_i64(dur_w), torch.int64), iptr(_i64(wb), torch.int64),
                                     iptr(_i64(src_w_len), torch.int64), iptr(out, torch.uint8), B, dur_w.shape[1], Lq,
                                     Lkv, stream_ptr()))
    return out.bool()


def get_rel_coef(dur, dur_len, mask):
    """model/linguistic_encoder.py:222-236 -> float [B, mask.shape[1]]; mask True = valid."""
    B, Lout = mask.shape
    out = torch.empty(B, Lout, device=mask.device, dtype=torch.float32)
    check(_lib.lib().mg_rel_coef(iptr(_i64(dur), torch.int64), iptr(_i64(dur_len), torch.int64),
                                 iptr(mask.to(torch.uint8).contiguous(), torch.uint8), fptr(out), B, dur.shape[1], Lout,
                                 stream_ptr()))
    return out
