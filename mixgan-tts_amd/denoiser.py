"""Denoiser (model/modules.py:382-446): same constructor, state_dict keys and forward
signature as the reference; the forward is one call into `mg_denoiser_fwd`."""
import ctypes
import itertools
import os
from collections import OrderedDict

import torch
from torch import nn

from . import _lib
from ._lib import fptr, iptr, check, stream_ptr, DenoiserDims
from .blocks import ConvNorm, LinearNorm, Mish, DiffusionEmbedding, ResidualBlock


# Every workspace instance of the process gets its own number: it is the high word of the in-kernel Philox offset
# (mg_denoiser_psample's noise_stream), the launch count kept inside the workspace is the low word.  A new shape, a
# workspace re-allocated after the cache evicted it and a second captured graph therefore never walk the same stream.
_NOISE_STREAMS = itertools.count(1)


def _rank_salt(dev):
    """Distinct per (rank, device): ranks that seed torch identically must not draw identical in-kernel noise."""
    rank = 0
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank = dist.get_rank()
        else:
            rank = int(os.environ.get("RANK", "0"))
    except (ValueError, RuntimeError):
        rank = 0
    z = (rank * 0x9E3779B97F4A7C15 + ((dev.index or 0) + 1) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
    z ^= z >> 31                                             # splitmix-style finaliser
    z = (z * 0x94D049BB133111EB) & (2 ** 64 - 1)
    return z ^ (z >> 29)


def raise_if_failed(owners=(), sync=False):
    """The single-launch kernels' failure word (include/mixgan_hip.h, mg_persist_error).  Polling it is a host memory
    read; sync=True first waits for the current stream, which makes the answer exact for everything launched so far.
    On a failure the owners' workspaces (whose sticky error words now poison every launch) are dropped and
    MixganHipError is raised."""
    if sync:
        torch.cuda.current_stream().synchronize()
    code = _lib.lib().mg_persist_error(0)
    if code:
        # several workgroups of the failed launch (and launches queued behind it on the poisoned workspace) write the
        # word: wait for all of them before resetting it, or a straggler would be reported against the next call
        torch.cuda.synchronize()
        _lib.lib().mg_persist_error(1)
        for o in owners:
            o.drop_workspaces()
        where = ("forward, layer %d" % (code - 1)) if code < 0x100 else ("backward, layer %d" % (code - 0x100))
        raise _lib.MixganHipError(
            "a neighbour hand-off of the single-launch denoiser kernel timed out (%s): another tenant is holding the "
            "GPU's workgroup slots.  The launch drained and its output is NaN; set MG_DENOISER_PERSIST=0 to use the "
            "launch-per-layer kernels on a shared GPU." % where)


class Denoiser(nn.Module):
    """Conditional diffusion denoiser -- drop-in for `model.modules.Denoiser`."""

    def __init__(self, preprocess_config, model_config):
        super().__init__()
        n_mel_channels = preprocess_config["preprocessing"]["mel"]["n_mel_channels"]
        d_encoder = model_config["transformer"]["encoder_hidden"]
        residual_channels = model_config["denoiser"]["residual_channels"]
        residual_layers = model_config["denoiser"]["residual_layers"]
        dropout = model_config["denoiser"]["denoiser_dropout"]
        multi_speaker = model_config["multi_speaker"]
        self.multi_speaker = bool(multi_speaker)

        self.input_projection = nn.Sequential(ConvNorm(n_mel_channels, residual_channels, kernel_size=1), nn.ReLU())
        self.diffusion_embedding = DiffusionEmbedding(residual_channels)
        self.mlp = nn.Sequential(
            LinearNorm(residual_channels, residual_channels * 4),
            Mish(),
            LinearNorm(residual_channels * 4, residual_channels),
        )
        self.residual_layers = nn.ModuleList(
            [ResidualBlock(d_encoder, residual_channels, dropout=dropout, multi_speaker=multi_speaker)
             for _ in range(residual_layers)])
        self.skip_projection = ConvNorm(residual_channels, residual_channels, kernel_size=1)
        self.output_projection = ConvNorm(residual_channels, n_mel_channels, kernel_size=1)
        nn.init.zeros_(self.output_projection.conv.weight)  # model/modules.py:418

        self._dims = DenoiserDims(residual_layers, residual_channels, d_encoder, n_mel_channels, int(self.multi_speaker))
        self._packed = None
        self._packed_key = None
        self._ws = OrderedDict()      # (B, L, save, device) -> workspace, least recently used first
        self._bws = OrderedDict()
        self._save_gen = 0
        self._grad_sink = None        # (flat buffer, {id(param): offset}) bound by distributed.GradBucket
        self._sink_used = False
        # "fp32": exact fp32 MFMA everywhere.  "bf16x3": inference-only forward whose residual-layer
        # GEMMs run as 3-term bf16-split products on the bf16 MFMA (fp32-grade, ~1e-5 relative);
        # grad-enabled forwards always take the fp32 path (the backward consumes fp32 activations).
        self.precision = "fp32"
        # callable(torch.cuda.Event) | None: called from the backward right after the library call, with an event that
        # marks the k=3 conv weight / bias gradients of all layers final in the bound gradient buffer (the rest of the
        # backward is still queued behind it) -- HotPathTrainer starts their all-reduce there
        self.after_conv3_grads = None

    # ------------------------------------------------------------------ packed-weight cache
    def _weight_table(self):
        """Pointer-table order of include/mixgan_hip.h (mg_denoiser_pack)."""
        t = [self.input_projection[0].conv.weight, self.input_projection[0].conv.bias,
             self.mlp[0].linear.weight, self.mlp[2].linear.weight,
             self.skip_projection.conv.weight, self.skip_projection.conv.bias,
             self.output_projection.conv.weight, self.output_projection.conv.bias]
        for blk in self.residual_layers:
            t += [blk.conv_layer.conv.weight, blk.conv_layer.conv.bias, blk.diffusion_projection.linear.weight,
                  blk.conditioner_projection.conv.weight, blk.conditioner_projection.conv.bias,
                  blk.output_projection.conv.weight, blk.output_projection.conv.bias,
                  blk.speaker_projection.linear.weight if self.multi_speaker else None, None]
        return t

    def packed_weights(self, with_backward=False):
        """The MFMA-ordered weight blob: a derived cache, rebuilt when any parameter changes
        (optimizer step, load_state_dict, .to()).  with_backward adds the transposed packs,
        precision == "bf16x3" the hi/lo bf16 packs."""
        table = self._weight_table()
        if self.precision not in ("fp32", "bf16x3"):
            raise ValueError("Denoiser.precision must be 'fp32' or 'bf16x3'")
        prev = self._packed_key[0] if self._packed_key is not None else 0
        with_backward = (1 if with_backward else 0) | (prev & 1) | (2 if self.precision == "bf16x3" else 0) | (prev & 2)
        # the 16x16x4-MFMA packs (16-frame tiles for single utterances / small batches): inference only -- a module
        # that trains repacks every step and never launches that width
        if not (with_backward & 1) and self._dims.channels == 256 and self._dims.cond_channels == 256:
            with_backward |= 4
        key = (with_backward,) + tuple((p.data_ptr(), p._version) for p in table if p is not None)
        if self._packed is None or key != self._packed_key:
            L = _lib.lib()
            dev = table[0].device
            if dev.type != "cuda":
                raise _lib.MixganHipError("Denoiser parameters are on %s: the HIP path needs them on the GPU" % dev)
            n = L.mg_denoiser_packed_floats(ctypes.byref(self._dims), int(with_backward))
            if self._packed is None or self._packed.numel() != n or self._packed.device != dev:
                self._packed = torch.empty(n, device=dev, dtype=torch.float32)
            ptrs = (ctypes.c_void_p * len(table))(*[None if p is None else fptr(p.detach()).value for p in table])
            freq = self._freq_cache(dev)
            # same tensors into the same buffer as last time (only their contents changed): the job table is resident
            jobs_key = (int(with_backward), self._packed.data_ptr(), freq.data_ptr()) + tuple(k[0] for k in key[1:])
            resident = 8 if jobs_key == getattr(self, "_jobs_key", None) else 0
            check(L.mg_denoiser_pack(ctypes.byref(self._dims), ptrs, fptr(freq), fptr(self._packed),
                                     int(with_backward) | resident, stream_ptr()))
            self._jobs_key = jobs_key
            self._packed_key = key
        return self._packed

    def _freq_cache(self, dev):
        f = getattr(self, "_freq_dev", None)
        if f is None or f.device != dev:
            f = self._freq_dev = self.diffusion_embedding.frequencies(dev).contiguous()
        return f

    def new_workspace(self, B, L, save, dev):
        """A private workspace (not cached): for owners that keep raw pointers into it across calls, e.g. a captured
        hipGraph, which must hold the tensor for as long as the graph lives."""
        n = _lib.lib().mg_denoiser_workspace_floats(ctypes.byref(self._dims), B, L, int(save))
        # zero-filled once: the single-launch forward keeps its ticket / launch counters in here and re-arms them itself
        ws = torch.zeros(n, device=dev, dtype=torch.float32)
        ws._mg_noise_stream = next(_NOISE_STREAMS)
        return ws

    def drop_workspaces(self):
        """Forget every cached workspace (after a reported kernel failure their sticky error words stay set)."""
        self._ws.clear()
        self._bws.clear()

    def check(self, sync=True):
        """Raise MixganHipError if a single-launch kernel reported a hand-off timeout (see raise_if_failed)."""
        raise_if_failed((self,), sync)

    def _workspace(self, B, L, save, dev):
        """Cached per shape.  The cache only saves re-allocation: whoever needs a workspace to outlive the call (a
        pending backward: autograd.DenoiserFn keeps it in ctx; a captured graph: new_workspace) holds the tensor
        itself, so evicting an entry never frees memory that is still referenced.  A save=True workspace whose
        backward has not run yet (`_mg_busy`) is not handed out again: the next grad-enabled forward gets a fresh one."""
        # keyed by stream too: the single-launch kernels keep their tickets and hand-off buffers in the workspace, so two
        # launches in flight on different streams must not share one
        k = (B, L, bool(save), dev, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
        ws = self._ws.get(k)
        if ws is not None and not (save and getattr(ws, "_mg_busy", False)):
            self._ws.move_to_end(k)
            return ws
        ws = self.new_workspace(B, L, save, dev)
        self._ws[k] = ws
        self._ws.move_to_end(k)
        while len(self._ws) > 8:
            self._ws.popitem(last=False)
        return ws

    # ------------------------------------------------------------------ forward
    def run(self, x_t, t, cond, spk, out=None, save=False, packed=None, ws=None):
        """x_t [B,M,L], t int64 [B], cond [B,H,L] contiguous, spk [B,H]|None -> [B,M,L] (no autograd).
        ws: explicit workspace (default: the per-shape cache).  With save=True the workspace that now holds the
        layer activations is left in `self.last_ws` for the caller to keep until its backward."""
        B, M, L = x_t.shape
        raise_if_failed((self,))      # a failure of earlier work that the host has not noticed yet (free: no sync)
        if packed is None:
            packed = self.packed_weights(with_backward=save)
        if ws is None:
            ws = self._workspace(B, L, save, x_t.device)
        if save:
            self._save_gen += 1
            ws._mg_busy = True
            ws._mg_gen = self._save_gen
            self.last_ws = ws
        if out is None:
            out = torch.empty_like(x_t)
        mode = 1 if save else (2 if self.precision == "bf16x3" else 0)
        if self._packed_key is not None and (self._packed_key[0] & 4) and packed is self._packed:
            mode |= 4      # MG_FWD_P16: the blob carries the 16-row packs
        check(_lib.lib().mg_denoiser_fwd(ctypes.byref(self._dims), fptr(packed), fptr(x_t), iptr(t, torch.int64),
                                         fptr(cond), fptr(spk, not self.multi_speaker), fptr(out), fptr(ws),
                                         ws.numel(), B, L, mode, stream_ptr()))
        return out

    def run_pair(self, x_a, t_a, x_b, t_b, cond, spk, cond_b=None, spk_b=None):
        """Two forwards over the same weights in ONE launch (mg_denoiser_fwd_pair): problem a without saves (the GAN
        step's D-phase forward), problem b with the activations kept like run(save=True) (its G-phase forward; the
        workspace is left in `self.last_ws` / returned).  cond_b / spk_b: problem b's conditioner / speakers when they
        differ from a's.  -> (out_a, out_b, ws_b), or None when the single-launch kernel does not take the shape (run
        them separately then)."""
        Bh, M, L = x_a.shape
        raise_if_failed((self,))
        packed = self.packed_weights(with_backward=True)
        dev = x_a.device
        ws_a = self._workspace(2 * Bh, L, False, dev)
        ws_b = self._workspace(Bh, L, True, dev)
        out_a, out_b = torch.empty_like(x_a), torch.empty_like(x_b)
        rc = _lib.lib().mg_denoiser_fwd_pair(ctypes.byref(self._dims), fptr(packed), fptr(x_a), iptr(t_a, torch.int64),
                                             fptr(x_b), iptr(t_b, torch.int64), fptr(cond),
                                             fptr(spk, not self.multi_speaker), fptr(cond_b, True),
                                             fptr(spk_b if self.multi_speaker else None, True), fptr(out_a),
                                             fptr(out_b), fptr(ws_a),
                                             ws_a.numel(), fptr(ws_b), ws_b.numel(), Bh, L, stream_ptr())
        if rc == -2:        # MG_ERR_SHAPE: not a shape of the single-launch kernel
            return None
        check(rc)
        self._save_gen += 1
        ws_b._mg_busy = True
        ws_b._mg_gen = self._save_gen
        self.last_ws = ws_b
        return out_a, out_b, ws_b

    def has_cond_projection(self, packed=None):
        """True when the packs hold the all-layer conditioner projection (the fp32 inference packs, C = H = 256)."""
        packed = self.packed_weights() if packed is None else packed
        return (self.precision != "bf16x3" and self._packed_key is not None and bool(self._packed_key[0] & 4)
                and packed is self._packed)

    def cond_projection(self, cond, out=None, packed=None):
        """conditioner_projection(cond) of every residual layer (model/blocks.py:1150,1160) as one product:
        cond [B,H,L] -> [B, n_layers * C, L].  It depends on neither x_t nor t, so a T-step sampling loop
        (model/diffusion.py:133-147) computes it once and hands it to each p_sample(cproj=...)."""
        packed = self.packed_weights() if packed is None else packed
        if not self.has_cond_projection(packed):
            raise _lib.MixganHipError("cond_projection needs the fp32 inference packs (channels = cond channels = 256)")
        B, _, L = cond.shape
        if out is None:
            out = torch.empty(B, self._dims.n_layers * self._dims.channels, L, device=cond.device, dtype=torch.float32)
        check(_lib.lib().mg_denoiser_cond_project(ctypes.byref(self._dims), fptr(packed), fptr(cond), fptr(out), B, L,
                                                  stream_ptr()))
        return out

    def step_vectors(self, ts, spk, packed=None):
        """The step-dependent vectors of forward() -- diffusion_embedding -> mlp (model/modules.py:433-434), every
        layer's diffusion_projection (+ speaker_projection(spk), model/blocks.py:1159-1163) -- for the n steps of a
        sampling loop in one set of launches.  ts int64 [n, B]; spk [B, H] | None.  Returns an opaque tensor for
        p_sample(step_vectors=(tensor, index, n))."""
        n, B = ts.shape
        packed = self.packed_weights() if packed is None else packed
        L = _lib.lib()
        out = torch.empty(L.mg_denoiser_step_vectors_floats(ctypes.byref(self._dims), n, B), device=ts.device,
                          dtype=torch.float32)
        check(L.mg_denoiser_step_vectors(ctypes.byref(self._dims), fptr(packed), iptr(ts, torch.int64),
                                         fptr(spk, not self.multi_speaker), fptr(out), out.numel(), n, B, stream_ptr()))
        return out

    def p_sample(self, x_t, t, cond, spk, coef1, coef2, logvar, noise=None, clip=True, out=None, x0_out=None,
                 packed=None, ws=None, cproj=None, cproj_out=None, step_vectors=None):
        """One reverse step (model/diffusion.py:121-129) as one library call: x_0 = forward(x_t); clamp; posterior mean
        + sigma * noise.  x_t [B,M,L], cond [B,H,L]; coef1 / coef2 / logvar: the diffusion's posterior_mean_coef1 / 2 and
        posterior_log_variance_clipped buffers.  noise None = drawn in the kernel: Philox keyed by a seed taken once
        from torch's generator (so torch.manual_seed reproduces a run) mixed with the rank and device; the counter is
        (this workspace's process-wide number, launches on it so far -- kept on the device), fresh on every call, every
        shape, every re-allocated workspace and every replay of every captured graph.  cproj_out / cproj: a
        [B, n_layers * C, L] buffer the first step of a sampling loop fills with its conditioner projections and the
        following steps read instead of projecting (same result bit for bit, 11 % fewer multiply-adds per step).
        step_vectors: (self.step_vectors(ts, spk), index of this step in ts, n) -- the loop's vectors computed at once.
        Returns x_{t-1} [B,M,L] (a new tensor or `out`, never x_t itself)."""
        B, M, L = x_t.shape
        raise_if_failed((self,))
        if packed is None:
            packed = self.packed_weights()
        if ws is None:
            ws = self._workspace(B, L, False, x_t.device)
        if out is None:
            out = torch.empty_like(x_t)
        if getattr(self, "_rng_seed", None) is None:
            self._rng_seed = (int(torch.randint(0, 2 ** 62, (1,)).item()) ^ _rank_salt(x_t.device)) & (2 ** 64 - 1)
        mode = 2 if self.precision == "bf16x3" else 0
        if self._packed_key is not None and (self._packed_key[0] & 4) and packed is self._packed:
            mode |= 4
        loop = None
        if cproj is not None or cproj_out is not None or step_vectors is not None:
            sv = step_vectors or (None, 0, 0)
            loop = ctypes.byref(_lib.SamplingLoop(fptr(cproj, True).value, fptr(cproj_out, True).value,
                                                  fptr(sv[0], True).value, int(sv[1]), int(sv[2])))
        check(_lib.lib().mg_denoiser_psample(
            ctypes.byref(self._dims), fptr(packed), fptr(x_t), iptr(t, torch.int64), fptr(cond),
            fptr(spk, not self.multi_speaker), fptr(coef1), fptr(coef2), fptr(logvar), coef1.numel(), fptr(noise, True),
            self._rng_seed, self._noise_stream_of(ws), int(bool(clip)), fptr(out), fptr(x0_out, True),
            loop, fptr(ws), ws.numel(), B, L, mode, stream_ptr()))
        return out

    @staticmethod
    def _noise_stream_of(ws):
        n = getattr(ws, "_mg_noise_stream", None)
        if n is None:      # a caller-made workspace: number it on first use
            n = ws._mg_noise_stream = next(_NOISE_STREAMS)
        return n

    def persist_status(self, B, L, ws=None):
        """{ticket, error, launches, done} of the single-launch forward's counters for this shape (synchronises);
        error != 0: a neighbour hand-off timed out and that launch's output is invalid."""
        ws = self._workspace(B, L, False, next(self.parameters()).device) if ws is None else ws
        host = (ctypes.c_uint * 4)()
        check(_lib.lib().mg_denoiser_persist_status(ctypes.byref(self._dims), fptr(ws), B, L, host, stream_ptr()))
        return {"ticket": host[0], "error": host[1], "launches": host[2], "done": host[3]}

    def forward(self, mel, diffusion_step, conditioner, speaker_emb, mask=None):
        """mel [B,1,M,T], diffusion_step [B], conditioner [B,H,T], speaker_emb [B,H]|None -> [B,1,M,T]."""
        if not mel.is_cuda:
            raise _lib.MixganHipError("Denoiser.forward on %s: the HIP path has no CPU fallback" % mel.device)
        x = mel[:, 0].contiguous()
        t = diffusion_step.to(torch.int64).contiguous()
        cond = conditioner.contiguous()
        spk = speaker_emb.contiguous() if (self.multi_speaker and speaker_emb is not None) else None
        needs_grad = torch.is_grad_enabled() and (
            x.requires_grad or cond.requires_grad or (spk is not None and spk.requires_grad)
            or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            from .autograd import DenoiserFn
            return DenoiserFn.apply(self, x, t, cond, spk, None, *[p for p in self._weight_table() if p is not None])[:, None]
        return self.run(x, t, cond, spk)[:, None]

    # ------------------------------------------------------------------ backward (autograd.DenoiserFn)
    # ------------------------------------------------------------------ gradient placement
    def grad_order(self):
        """Parameters in the order the backward produces their gradients: the 8 head / tail tensors, then each
        per-layer kind for all layers (mg_denoiser_bwd writes every kind as ONE layer-major array).  A
        distributed.GradBucket laid out in this order lets the backward write straight into the all-reduce
        buffer: no per-parameter gather copy (148 small launches per step otherwise)."""
        table = self._weight_table()
        head = [p for p in table[:8] if p is not None]
        per_layer = [table[8 + 9 * l: 8 + 9 * (l + 1)] for l in range(len(self.residual_layers))]
        kinds = [[lay[j] for lay in per_layer] for j in range(9)]
        return head + [p for kind in kinds for p in kind if p is not None]

    def bind_grad_buffer(self, flat, offsets):
        """flat: fp32 buffer; offsets: {id(param): element offset}.  Used only when every kind is layer-contiguous."""
        table = self._weight_table()
        NL = len(self.residual_layers)
        for j in range(9):
            ps = [table[8 + 9 * l + j] for l in range(NL)]
            if ps[0] is None:
                continue
            for l in range(NL):
                if id(ps[l]) not in offsets or offsets[id(ps[l])] != offsets[id(ps[0])] + l * ps[0].numel():
                    raise _lib.MixganHipError("bind_grad_buffer: parameters are not laid out in grad_order()")
        self._grad_sink = (flat, dict(offsets))

    def _grad_targets(self, dev):
        """(head grads [8], per-kind layer-major arrays [9]) -- views of the bound bucket when every parameter's
        .grad is unset (autograd then adopts the returned views as .grad without a copy), fresh tensors otherwise
        (a second backward before zero_grad must ADD to .grad, which autograd does from a separate tensor)."""
        table = self._weight_table()
        NL = len(self.residual_layers)
        sink = self._grad_sink
        use_sink = (sink is not None and sink[0].device == dev
                    and all(p is None or (p.grad is None and id(p) in sink[1]) for p in table))
        self._sink_used = use_sink
        def alloc(p, lead=()):
            if p is None:
                return None
            if use_sink:
                off = sink[1][id(p)]
                n = p.numel() * (lead[0] if lead else 1)
                return sink[0][off:off + n].view(*lead, *p.shape)
            return torch.empty(*lead, *p.shape, device=dev, dtype=torch.float32)
        return [alloc(p) for p in table[:8]], [alloc(table[8 + j], (NL,)) for j in range(9)]

    def run_backward(self, g_out, x_t, t, cond, spk, ws, gen, want_dx, want_dcond, want_dspk):
        """Returns (d_x_t, d_cond, d_spk, [param grads in weight-table order, None entries skipped]).
        ws: the workspace the grad-enabled forward filled (kept alive by the autograd node)."""
        if getattr(ws, "_mg_gen", None) != gen:
            raise _lib.MixganHipError(
                "Denoiser backward: the saved activations were overwritten by a later forward into the same workspace")
        L_ = _lib.lib()
        B, M, L = x_t.shape
        dev = x_t.device
        d = self._dims
        NL = d.n_layers
        raise_if_failed((self,))
        packed = self.packed_weights(with_backward=True)
        k = (B, L, dev, torch.cuda.current_stream(dev).cuda_stream)
        bws = self._bws.get(k)
        if bws is None:
            # transient scratch of this call only (stream-ordered), so eviction is always safe
            # zero-filled once: the single-launch data-gradient kernel keeps its counters in here
            bws = torch.zeros(L_.mg_denoiser_bwd_workspace_floats(ctypes.byref(d), B, L), device=dev)
            self._bws[k] = bws
            while len(self._bws) > 4:
                self._bws.popitem(last=False)
        else:
            self._bws.move_to_end(k)
        head, kinds = self._grad_targets(dev)
        # per-layer gradients are slices of layer-major tensors: mg_denoiser_bwd computes them for all layers at once
        grads = list(head)
        for l in range(NL):
            grads += [None if a is None else a[l] for a in kinds]
        ptrs = (ctypes.c_void_p * len(grads))(*[None if g is None else g.data_ptr() for g in grads])
        d_x = torch.empty_like(x_t) if want_dx else None
        d_cond = torch.empty_like(cond) if want_dcond else None
        d_spk = torch.empty_like(spk) if (want_dspk and spk is not None) else None
        hook = self.after_conv3_grads if self._sink_used else None
        ev = None
        if hook is not None:
            ev = getattr(self, "_conv3_event", None)
            if ev is None:      # torch creates the hipEvent_t at the first record(): make the handle exist
                ev = self._conv3_event = torch.cuda.Event()
                ev.record()
        check(L_.mg_denoiser_bwd_staged(ctypes.byref(d), fptr(packed), fptr(g_out), fptr(x_t), fptr(cond),
                                        fptr(spk, not self.multi_speaker), fptr(ws), fptr(bws), bws.numel(), ptrs,
                                        fptr(d_x, True), fptr(d_cond, True), fptr(d_spk, True), B, L,
                                        ctypes.c_void_p(ev.cuda_event) if ev is not None else None, stream_ptr()))
        if hook is not None:
            hook(ev)
        ws._mg_busy = False
        return d_x, d_cond, d_spk, [g for g in grads if g is not None]
