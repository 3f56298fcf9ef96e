"""ctypes binding of libmixgan_hip.so (C ABI: include/mixgan_hip.h)."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXPORTS = (
    "mg_version", "mg_error_string",
    "mg_conv_packed_floats", "mg_conv_pack", "mg_conv_pack_at", "mg_conv1d_fwd", "mg_conv1d_fwd_ex", "mg_conv1d_fwd_split",
    "mg_upsample_zero_act", "mg_diffuse_trace_bwd", "mg_bgemm", "mg_softmax_rows_fwd", "mg_softmax_rows_bwd", "mg_layernorm_cm_train_fwd",
    "mg_layernorm_cm_bwd", "mg_bn_stats", "mg_bn_act_fwd", "mg_bn_act_bwd_reduce", "mg_bn_act_bwd_apply", "mg_conv_transpose_packed_floats", "mg_conv_transpose_pack", "mg_conv_transpose1d_fwd",
    "mg_conv1d_wgrad_scratch_floats", "mg_conv1d_wgrad", "mg_conv1d_wgrad_strided", "mg_conv1d_wgrad_grouped",
    "mg_conv1d_wgrad_grouped_scratch_floats", "mg_conv1d_wgrad_grouped_bias", "mg_rowsum",
    "mg_diffuse_fwd", "mg_posterior_sample_fwd", "mg_posterior_sample_bwd", "mg_spec_affine", "mg_transpose_bml",
    "mg_denoiser_packed_floats", "mg_denoiser_pack", "mg_denoiser_workspace_floats", "mg_denoiser_fwd",
    "mg_denoiser_bwd_workspace_floats", "mg_denoiser_bwd", "mg_denoiser_bwd_staged",
    "mg_profile_begin", "mg_profile_begin_sampled", "mg_profile_end", "mg_transpose_bml_strided", "mg_act_bwd", "mg_upsample_zero",
    "mg_step_mlp_fwd", "mg_step_mlp_bwd", "mg_linear_small_fwd", "mg_linear_small_bwd",
    "mg_loss_sum", "mg_loss_grad", "mg_mel_l1_fwd", "mg_mel_l1_bwd", "mg_attention_fwd", "mg_attention_fwd_f16", "mg_layernorm_cm_fwd",
    "mg_length_regulate_fwd", "mg_length_regulate_bwd", "mg_word_pool_fwd", "mg_word_pool_bwd", "mg_mapping_mask",
    "mg_rel_coef", "mg_resblock_fwd", "mg_gate_bwd", "mg_mish_fwd", "mg_mish_bwd", "mg_step_embed",
    "mg_denoiser_psample", "mg_denoiser_cond_project", "mg_denoiser_step_vectors_floats", "mg_denoiser_step_vectors", "mg_denoiser_persist_status", "mg_persist_error", "mg_denoiser_fwd_pair",
    "mg_grad_norm_scratch_floats", "mg_grad_norm", "mg_adam_flat", "mg_adam_flat_dev",
    "mg_multi_loss_scratch_floats", "mg_multi_loss_fwd", "mg_multi_loss_bwd",
)


class MixganHipError(RuntimeError):
    pass


MG_LOSS_MAX_TERMS = 16
MG_LOSS_GROUPS = 4


class LossTerm(ctypes.Structure):
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("da", ctypes.c_void_p), ("n", ctypes.c_size_t),
                ("c", ctypes.c_float), ("weight", ctypes.c_float), ("mode", ctypes.c_int32), ("group", ctypes.c_int32)]


class DenoiserDims(ctypes.Structure):
    _fields_ = [("n_layers", ctypes.c_int32), ("channels", ctypes.c_int32), ("cond_channels", ctypes.c_int32),
                ("mel_bins", ctypes.c_int32), ("multi_speaker", ctypes.c_int32)]


class SamplingLoop(ctypes.Structure):
    """mg_sampling_loop (include/mixgan_hip.h): what the steps of one sampling loop share."""
    _fields_ = [("cproj", ctypes.c_void_p), ("cproj_out", ctypes.c_void_p), ("step_vectors", ctypes.c_void_p),
                ("step_index", ctypes.c_int32), ("step_count", ctypes.c_int32)]


def library_path():
    # MG_HIP_LIB: another build of the same library (tools: A/B timing of two builds inside one gpurun call)
    return os.environ.get("MG_HIP_LIB") or os.path.join(_HERE, "libmixgan_hip.so")


def build(force=False):
    """Compile the HIP sources in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(library_path()):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"] + (["-B"] if force else []))
    return library_path()


def lib():
    """Load the HIP library; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise MixganHipError(
                "libmixgan_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C mixgan-tts_amd/csrc`. There is no CPU fallback." % path)
        L = ctypes.CDLL(path)
        L.mg_error_string.restype = ctypes.c_char_p
        _declare(L)
        _LIB = L
    return _LIB


def _declare(L):
    """argtypes/restype for every export of include/mixgan_hip.h."""
    vp, i, f, sz, lg = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_long
    dp = ctypes.POINTER(DenoiserDims)
    sig = {
        "mg_version": (i, []),
        "mg_error_string": (ctypes.c_char_p, [i]),
        "mg_conv_packed_floats": (sz, [i, i, i, i]),
        "mg_conv_pack": (i, [vp, vp, i, i, i, i, vp]),
        "mg_conv1d_fwd": (i, [vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, f, i, vp]),
        "mg_conv1d_fwd_ex": (i, [vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, f, i, f, f, i, vp]),
        "mg_conv1d_fwd_split": (i, [vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, f, i, f, f, i, vp, sz, vp]),
        "mg_upsample_zero_act": (i, [vp, vp, i, i, i, i, f, vp]),
        "mg_diffuse_trace_bwd": (i, [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, vp]),
        "mg_bgemm": (i, [vp, vp, vp, i, i, i, i, i] + [lg] * 11 + [f, i, vp]),
        "mg_softmax_rows_fwd": (i, [vp, vp, i, i, i, f, vp]),
        "mg_softmax_rows_bwd": (i, [vp, vp, i, i, i, f, vp]),
        "mg_layernorm_cm_train_fwd": (i, [vp, vp, f, vp, vp, vp, vp, vp, vp, i, i, i, f, vp]),
        "mg_layernorm_cm_bwd": (i, [vp, vp, vp, vp, vp, f, vp, vp, vp, vp, i, i, i, f, vp]),
        "mg_bn_stats": (i, [vp, vp, vp, i, i, i, vp]),
        "mg_bn_act_fwd": (i, [vp, vp, vp, vp, vp, vp, f, i, vp, vp, i, i, i, vp]),
        "mg_bn_act_bwd_reduce": (i, [vp, vp, f, vp, vp, vp, vp, i, vp, vp, i, i, i, vp]),
        "mg_bn_act_bwd_apply": (i, [vp, vp, f, vp, vp, vp, vp, vp, vp, vp, f, i, vp, i, i, i, vp]),
        "mg_conv_transpose_packed_floats": (sz, [i, i, i]),
        "mg_conv_transpose_pack": (i, [vp, vp, i, i, i, vp]),
        "mg_conv_transpose1d_fwd": (i, [vp, vp, vp, vp, i, i, i, i, i, f, f, vp]),
        "mg_diffuse_fwd": (i, [vp] * 9 + [i, i, i, i, vp]),
        "mg_posterior_sample_fwd": (i, [vp] * 10 + [i, i, i, i, i, vp]),
        "mg_transpose_bml": (i, [vp] * 5 + [i, i, i, i, i, vp]),
        "mg_posterior_sample_bwd": (i, [vp] * 7 + [i, i, i, i, i, vp]),
        "mg_spec_affine": (i, [vp, vp, vp, vp, i, sz, i, vp]),
        "mg_conv_pack_at": (i, [vp, vp, i, i, i, i, i, i, vp]),
        "mg_conv1d_wgrad_scratch_floats": (sz, [i, i, i]),
        "mg_conv1d_wgrad": (i, [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, f, i, vp]),
        "mg_conv1d_wgrad_strided": (i, [vp, ctypes.c_long, vp, ctypes.c_long, vp, vp, vp, i, i, i, i, i, i, i, i, f, i, vp]),
        "mg_rowsum": (i, [vp, ctypes.c_long, i, i, i, vp, vp, f, i, vp]),
        "mg_conv1d_wgrad_grouped_scratch_floats": (sz, [i, i, i, i]),
        "mg_conv1d_wgrad_grouped": (i, [vp, lg, lg, vp, lg, lg, vp, lg, vp, i, i, i, i, i, i, i, i, i, f, i, vp]),
        "mg_conv1d_wgrad_grouped_bias": (i, [vp, lg, lg, vp, lg, lg, vp, lg, vp, lg, vp, i, i, i, i, i, i, i, i, i, f, i, vp]),
        "mg_denoiser_packed_floats": (sz, [dp, i]),
        "mg_denoiser_pack": (i, [dp, vp, vp, vp, i, vp]),
        "mg_denoiser_bwd_workspace_floats": (sz, [dp, i, i]),
        "mg_denoiser_bwd": (i, [dp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, i, i, vp]),
        "mg_denoiser_bwd_staged": (i, [dp, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, i, i, vp, vp]),
        "mg_denoiser_workspace_floats": (sz, [dp, i, i, i]),
        "mg_denoiser_fwd": (i, [dp, vp, vp, vp, vp, vp, vp, vp, sz, i, i, i, vp]),
        "mg_denoiser_fwd_pair": (i, [dp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, sz, i, i, vp]),
        "mg_transpose_bml_strided": (i, [vp] * 5 + [i, i, i, i, i, ctypes.c_long, vp]),
        "mg_act_bwd": (i, [vp, vp, vp, i, sz, vp]),
        "mg_upsample_zero": (i, [vp, vp, i, i, i, i, vp]),
        "mg_step_mlp_fwd": (i, [vp] * 8 + [i, i, i, i, vp]),
        "mg_step_mlp_bwd": (i, [vp] * 8 + [i, i, i, i, vp]),
        "mg_linear_small_fwd": (i, [vp, vp, vp, i, i, i, vp]),
        "mg_linear_small_bwd": (i, [vp, vp, vp, vp, vp, i, i, i, vp]),
        "mg_loss_sum": (i, [vp, vp, f, i, sz, vp, vp]),
        "mg_multi_loss_scratch_floats": (sz, []),
        "mg_multi_loss_fwd": (i, [vp, i, vp, vp, vp]),
        "mg_multi_loss_bwd": (i, [vp, i, vp, vp]),
        "mg_grad_norm_scratch_floats": (sz, []),
        "mg_grad_norm": (i, [vp, sz, f, vp, vp, vp]),
        "mg_adam_flat": (i, [vp, vp, vp, vp, sz, f, f, f, f, f, lg, vp, vp]),
        "mg_adam_flat_dev": (i, [vp, vp, vp, vp, sz, f, f, f, f, f, lg, vp, vp, vp]),
        "mg_loss_grad": (i, [vp, vp, f, i, vp, f, sz, vp, vp]),
        "mg_mel_l1_fwd": (i, [vp, vp, vp, i, i, vp, vp]),
        "mg_mel_l1_bwd": (i, [vp, vp, vp, i, i, vp, vp, vp, vp]),
        "mg_attention_fwd": (i, [vp, vp, vp, i, i, i, i, f, vp]),
        "mg_attention_fwd_f16": (i, [vp, vp, vp, i, i, i, i, f, vp]),
        "mg_layernorm_cm_fwd": (i, [vp, vp, vp, vp, vp, vp, i, i, i, f, vp]),
        "mg_length_regulate_fwd": (i, [vp, vp, vp, vp, i, i, i, i, vp]),
        "mg_length_regulate_bwd": (i, [vp, vp, vp, i, i, i, i, vp]),
        "mg_word_pool_fwd": (i, [vp, vp, vp, vp, i, i, i, i, i, i, vp]),
        "mg_word_pool_bwd": (i, [vp, vp, vp, vp, i, i, i, i, i, i, vp]),
        "mg_mapping_mask": (i, [vp, vp, vp, vp, i, i, i, i, vp]),
        "mg_rel_coef": (i, [vp, vp, vp, vp, i, i, i, vp]),
        "mg_resblock_fwd": (i, [vp] * 16 + [i, i, i, i, vp]),
        "mg_gate_bwd": (i, [vp, vp, vp, vp, i, i, i, vp]),
        "mg_mish_fwd": (i, [vp, vp, sz, vp]),
        "mg_mish_bwd": (i, [vp, vp, vp, sz, vp]),
        "mg_step_embed": (i, [vp, vp, vp, i, i, vp]),
        "mg_denoiser_psample": (i, [dp] + [vp] * 8 + [i, vp, ctypes.c_ulonglong, ctypes.c_ulonglong, i, vp, vp, vp, vp, sz,
                                    i, i, i, vp]),
        "mg_denoiser_step_vectors_floats": (sz, [dp, i, i]),
        "mg_denoiser_step_vectors": (i, [dp, vp, vp, vp, vp, sz, i, i, vp]),
        "mg_denoiser_cond_project": (i, [dp, vp, vp, vp, i, i, vp]),
        "mg_persist_error": (ctypes.c_uint, [i]),
        "mg_denoiser_persist_status": (i, [dp, vp, i, i, vp, vp]),
        "mg_profile_begin": (i, [i]),
        "mg_profile_begin_sampled": (i, [i, i]),
        "mg_profile_end": (i, [vp, i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args


def check(rc):
    if rc != 0:
        raise MixganHipError("libmixgan_hip: %s (code %d)" % (lib().mg_error_string(rc).decode(), rc))


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def fptr(t, allow_none=False):
    """Device pointer of a contiguous fp32 CUDA tensor (the ABI takes plain pointers)."""
    if t is None:
        if allow_none:
            return ctypes.c_void_p(0)
        raise MixganHipError("missing tensor")
    if not t.is_cuda:
        raise MixganHipError("tensor is not on the GPU: the HIP path has no CPU fallback")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise MixganHipError("expected a contiguous fp32 tensor, got %s contiguous=%s" % (t.dtype, t.is_contiguous()))
    return ctypes.c_void_p(t.data_ptr())


def iptr(t, dtype, allow_none=False):
    if t is None:
        if allow_none:
            return ctypes.c_void_p(0)
        raise MixganHipError("missing tensor")
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise MixganHipError("expected a contiguous %s CUDA tensor" % dtype)
    return ctypes.c_void_p(t.data_ptr())
