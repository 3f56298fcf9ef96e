"""GaussianDiffusion (model/diffusion.py:38-235): same constructor, buffers, state_dict keys,
forward/sampling signatures and hidden `cond`/`spk_emb` stash as the reference; all tensor
math runs in the HIP library.

Differences a caller can see (all opt-in or documented in DESIGN.md):
  * `noise_fn` -- optional callable(shape)->tensor replacing `torch.randn` so tests can inject
    the reference's exact draws (GPU and CPU generators differ); `t_fn` likewise for randint;
  * `sampling(noise=None, keep_trace=True)`: keep_trace=False returns only the final mel
    instead of the reference's T+1 retained tensors;
  * `use_graph` (attribute, default False): the inference branch of forward() replays its T-step loop as one
    captured hipGraph;
  * no tqdm progress bars in the sampling loop (they force a host iteration per step);
  * `check_failures` (attribute, default True): `sampling()` waits for its last step and raises MixganHipError if a
    single-launch kernel reported a hand-off timeout (its output is NaN then; include/mixgan_hip.h, mg_persist_error)
    -- callers that pipeline further GPU work behind `sampling()` and check `denoise_fn.check()` themselves turn it off.
"""
import json
import os

import numpy as np
import torch
from torch import nn

from . import ops, _lib
from .denoiser import Denoiser
from .schedule import beta_schedule, diffusion_buffers, BUFFER_NAMES


class GaussianDiffusion(nn.Module):
    def __init__(self, args, preprocess_config, model_config, train_config):
        super().__init__()
        self.model = args.model
        self.denoise_fn = Denoiser(preprocess_config, model_config)
        self.mel_bins = preprocess_config["preprocessing"]["mel"]["n_mel_channels"]
        den = model_config["denoiser"]
        betas = beta_schedule(
            den["noise_schedule_naive"],
            den["timesteps" if self.model == "naive" else "shallow_timesteps"],
            min_beta=den["min_beta"], max_beta=den["max_beta"], s=den["s"])
        self.num_timesteps = int(betas.shape[0])
        self.loss_type = train_config["loss"]["noise_loss"]
        bufs = diffusion_buffers(betas)
        for k in BUFFER_NAMES:
            self.register_buffer(k, torch.tensor(bufs[k], dtype=torch.float32))
        with open(os.path.join(preprocess_config["path"]["preprocessed_path"], "stats.json")) as f:
            stats = json.load(f)
        keep = den["keep_bins"]
        self.register_buffer("spec_min", torch.FloatTensor(stats["spec_min"])[None, None, :keep])
        self.register_buffer("spec_max", torch.FloatTensor(stats["spec_max"])[None, None, :keep])
        self.noise_fn = None
        self.t_fn = None
        # inference through forward(): replay the T-step loop as one captured hipGraph (BASELINE configs[2]);
        # synthesize.py-style callers set `model.diffusion.use_graph = True` once
        self.use_graph = False
        self.check_failures = True
        self.cond = None
        self.spk_emb = None

    # ------------------------------------------------------------------ helpers
    def _buf(self):
        return {k: getattr(self, k) for k in BUFFER_NAMES + ("spec_min", "spec_max")}

    def _randn(self, shape, device):
        if self.noise_fn is not None:
            return self.noise_fn(tuple(shape)).to(device=device, dtype=torch.float32).contiguous()
        return torch.randn(shape, device=device)

    def _randint(self, B, device):
        if self.t_fn is not None:
            return self.t_fn((B,)).to(device=device, dtype=torch.int64).contiguous()
        return torch.randint(0, self.num_timesteps, (B,), device=device).long()

    @staticmethod
    def _bml(x4):
        """[B,1,M,L] view of the reference layout -> our contiguous [B,M,L]."""
        return x4[:, 0].contiguous()

    # ------------------------------------------------------------------ reference surface
    def norm_spec(self, x):
        from .autograd import spec_affine
        return spec_affine(x, self.spec_min, self.spec_max, True)

    def denorm_spec(self, x):
        from .autograd import spec_affine
        return spec_affine(x, self.spec_min, self.spec_max, False)

    def out2mel(self, x):
        return x

    def q_sample(self, x_start, t, noise=None):
        """x_start [B,1,M,L] already normalised (model/diffusion.py:147-153)."""
        x = self._bml(x_start)
        B, M, L = x.shape
        noise = self._randn((B, 1, M, L), x.device) if noise is None else noise
        mel_like = ops.transpose_bml(x, True)            # the kernel takes the [B,L,M] side
        ident = {**self._buf(), "spec_min": torch.full_like(self.spec_min, -1.0),
                 "spec_max": torch.full_like(self.spec_max, 1.0)}       # norm_spec == identity
        return ops.diffuse(mel_like, t.clamp(min=0).contiguous(), self._bml(noise), None, ident)[:, None]

    def diffuse_fn(self, x_start, t, noise=None, keep=None):
        """mel [B,L,M] -> x_t [B,1,M,L] (model/diffusion.py:177-185); t<0 rows return the clean mel."""
        B, L, M = x_start.shape
        noise = self._randn((B, 1, M, L), x_start.device) if noise is None else noise
        out = ops.diffuse(x_start.contiguous(), t.contiguous(), self._bml(noise), keep, self._buf())
        t[t < 0] = 0                                      # the reference mutates its argument
        return out[:, None]

    def q_posterior_sample(self, x_start, x_t, t, repeat_noise=False, keep=None, clip=False):
        """model/diffusion.py:113-119 on [B,1,M,L] tensors."""
        B, _, M, L = x_t.shape
        noise = self._randn((B, 1, M, L), x_t.device)
        return ops.posterior_sample(self._bml(x_start), self._bml(x_t), t.contiguous(), self._bml(noise), keep,
                                    self._buf(), clip=clip)[:, None]

    @torch.no_grad()
    def p_sample(self, x_t, t, cond, spk_emb, clip_denoised=True, repeat_noise=False):
        """model/diffusion.py:121-129; x_t [B,1,M,L], cond [B,H,L]."""
        x = self._bml(x_t)
        B, M, L = x.shape
        noise = self._bml(self._randn((B, 1, M, L), x.device)) if self.noise_fn is not None else None
        return self._p_sample_bml(x, t.contiguous(), cond.contiguous(), spk_emb, noise, clip_denoised)[:, None]

    def _p_sample_bml(self, x, t, cond, spk, noise, clip=True, out=None, packed=None, ws=None, cproj=None,
                      cproj_out=None, step_vectors=None):
        """Denoiser.forward + clamp + posterior sample on [B,M,L] tensors as ONE library call (one kernel launch on the
        fp32 inference path).  noise None: N(0,1) drawn inside the kernel.  cproj_out / cproj: _loop_cond_buffer(), written
        by the first step of a sampling loop and read by the others; step_vectors: (_loop_step_vectors(), index, T)."""
        return self.denoise_fn.p_sample(x, t, cond, spk, self.posterior_mean_coef1, self.posterior_mean_coef2,
                                        self.posterior_log_variance_clipped, noise, clip, out, None, packed, ws, cproj,
                                        cproj_out, step_vectors)

    # The T steps of a sampling loop (model/diffusion.py:133-147) see the same conditioner, and each residual layer's
    # conditioner_projection(cond) (model/blocks.py:1160) depends on neither x_t nor t: the first step of a loop leaves
    # its projections in a buffer and the steps behind it read them instead of projecting again (bit-identical results;
    # MG_COND_PREPROJECT=0 or cond_preproject = False: every step projects).
    cond_preproject = True
    cond_preproject_max_bytes = 32 << 30      # 20 KB per frame: B=16, L=1000 is 328 MB; beyond this the steps project

    def _preprojects(self, packed):
        return (self.cond_preproject and self.num_timesteps >= 2 and os.environ.get("MG_COND_PREPROJECT", "1") != "0"
                and self.denoise_fn.has_cond_projection(packed))

    def _loop_cond_buffer(self, cond, packed, new=False):
        """Where a sampling loop over `cond` keeps its conditioner projections ([B, n_layers * C, L]; None: the loop
        projects in every step).  Pass it as cproj_out to the loop's first step and as cproj to the others."""
        den = self.denoise_fn
        if not self._preprojects(packed):
            return None
        B, _, L = cond.shape
        shape = (B, den._dims.n_layers * den._dims.channels, L)
        if 4 * shape[0] * shape[1] * shape[2] > self.cond_preproject_max_bytes:
            return None
        if new:
            return torch.empty(shape, device=cond.device, dtype=torch.float32)
        key = (B, L, cond.device, torch.cuda.current_stream(cond.device).cuda_stream)
        if self._cproj_buf is None or self._cproj_buf[0] != key:
            self._cproj_buf = (key, torch.empty(shape, device=cond.device, dtype=torch.float32))
        return self._cproj_buf[1]

    _cproj_buf = None

    def _loop_ts(self, B, dev):
        """t of the T steps of a sampling loop, in loop order (T-1 .. 0): int64 [T, B]; row i is step i's t."""
        key = (B, dev, self.num_timesteps)
        if self._loop_ts_buf is None or self._loop_ts_buf[0] != key:
            ts = torch.arange(self.num_timesteps - 1, -1, -1, device=dev, dtype=torch.long)[:, None].expand(-1, B)
            self._loop_ts_buf = (key, ts.contiguous())
        return self._loop_ts_buf[1]

    _loop_ts_buf = None

    def _loop_step_vectors(self, ts, k, spk, packed, held=None):
        """A sampling loop knows its t values in advance: the step embedding, its MLP and the per-layer diffusion /
        speaker projections of many steps in one set of launches instead of one set per step.  Called before step k of
        the loop with what the previous call returned; returns (vectors, first step, count) covering step k -- a new
        chunk of up to ~1024 rows (steps x utterances: 50 MB) when k runs past the held one -- or None (each step
        computes its own)."""
        if not self._preprojects(packed):
            return None
        if held is not None and held[1] <= k < held[1] + held[2]:
            return held
        T, B = ts.shape
        n = min(T - k, max(1, 1024 // B))
        return (self.denoise_fn.step_vectors(ts[k:k + n], spk, packed), k, n)

    @torch.no_grad()
    def sampling(self, noise=None, keep_trace=True, use_graph=False, _final_keep=None):
        """Reverse process from the stashed cond/spk (model/diffusion.py:155-165).
        Returns the list of denormalised mels [B,L,M] (T+1 entries, or only the last).
        use_graph=True replays the whole T-step loop as one captured hipGraph (static shapes, t read
        from device memory, on-device RNG); it implies keep_trace=False and ignores `noise_fn`."""
        cond = self.cond
        B, _, L = cond.shape
        dev = cond.device
        M, T = self.mel_bins, self.num_timesteps
        if use_graph and self.noise_fn is None:
            x = self._bml(torch.randn((B, 1, M, L), device=dev) if noise is None else noise)
            x = self._sampling_graph(x, cond, self.spk_emb)
            res = [ops.transpose_bml(x, True, 2, self.spec_min, self.spec_max, _final_keep)]
            self._check_failed()
            return res
        buf = self._buf()
        den = self.denoise_fn
        packed = den.packed_weights()
        x = self._bml(self._randn((B, 1, M, L), dev) if noise is None else noise)
        xs = [x] if keep_trace else None
        cond = cond.contiguous()
        cproj = self._loop_cond_buffer(cond, packed)
        ts = self._loop_ts(B, dev)
        vecs = None
        for k in range(T):                      # step k of the loop: t = T-1-k
            vecs = self._loop_step_vectors(ts, k, self.spk_emb, packed, vecs)
            nz = self._bml(self._randn((B, 1, M, L), dev)) if self.noise_fn is not None else None
            first = k == 0
            x = self._p_sample_bml(x, ts[k], cond, self.spk_emb, nz, True, packed=packed,
                                   cproj=None if first else cproj, cproj_out=cproj if first else None,
                                   step_vectors=None if vecs is None else (vecs[0], k - vecs[1], vecs[2]))
            if keep_trace:
                xs.append(x)
        outs = xs if keep_trace else [x]
        res = [ops.transpose_bml(a, True, 2, self.spec_min, self.spec_max) for a in outs[:-1]]
        res = res + [ops.transpose_bml(outs[-1], True, 2, self.spec_min, self.spec_max, _final_keep)]
        self._check_failed()
        return res

    def _check_failed(self):
        """End of a sampling loop: one stream synchronisation, then the failure word (a host memory read)."""
        if self.check_failures and not torch.cuda.is_current_stream_capturing():
            try:
                self.denoise_fn.check(sync=True)
            except _lib.MixganHipError:
                self._graph = None       # the captured graph's own workspace carries the sticky error word
                raise

    def _sampling_graph(self, x_start, cond, spk):
        """The T-step p_sample loop as one hipGraph: captured once per (B, L, T, device) on a side
        stream, replayed with new contents in the static x / cond / spk buffers."""
        B, M, L = x_start.shape
        dev = x_start.device
        T = self.num_timesteps
        den = self.denoise_fn
        packed = den.packed_weights()
        key = (B, L, T, dev, packed.data_ptr(), den._packed_key)
        g = getattr(self, "_graph", None)
        if g is None or g["key"] != key:
            buf = self._buf()
            st = {"key": key, "x": [torch.empty_like(x_start), torch.empty_like(x_start)],
                  "cond": torch.empty_like(cond), "spk": None if spk is None else torch.empty_like(spk),
                  "ts": self._loop_ts(B, dev).clone(),
                  # the graph bakes in raw pointers: it owns its workspace and holds the packed weights it captured
                  "ws": den.new_workspace(B, L, False, dev), "packed": packed,
                  "cproj": self._loop_cond_buffer(cond, packed, new=True)}
            st["cond"].copy_(cond)
            if spk is not None:
                st["spk"].copy_(spk)
            st["x"][0].copy_(x_start)

            def loop():
                cur = 0
                vecs, st["vecs"] = None, []
                for k in range(T):
                    nxt = self._loop_step_vectors(st["ts"], k, st["spk"], packed, vecs)   # inside the graph: spk changes
                    if nxt is not vecs:
                        st["vecs"].append(nxt)         # the graph's kernels read these buffers on every replay
                    vecs = nxt
                    first = k == 0
                    self._p_sample_bml(st["x"][cur], st["ts"][k], st["cond"], st["spk"], None, True,
                                       out=st["x"][cur ^ 1], packed=packed, ws=st["ws"],
                                       cproj=None if first else st["cproj"], cproj_out=st["cproj"] if first else None,
                                       step_vectors=None if vecs is None else (vecs[0], k - vecs[1], vecs[2]))
                    cur ^= 1
                return cur

            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                loop()                                  # warm-up: workspaces allocated outside capture
            torch.cuda.current_stream(dev).wait_stream(side)
            st["x"][0].copy_(x_start)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                st["final"] = loop()
            st["graph"] = graph
            self._graph = g = st
        g["cond"].copy_(cond)
        if spk is not None:
            g["spk"].copy_(spk)
        g["x"][0].copy_(x_start)
        g["graph"].replay()
        return g["x"][g["final"]].clone()

    def diffuse_trace(self, x_start, mask):
        """aux only (model/diffusion.py:167-175): mask True = pad.  Differentiable w.r.t. x_start (the coarse mel
        of aux pre-training): the T+1 entries come out of one autograd node whose backward is mg_diffuse_trace_bwd."""
        if torch.is_grad_enabled() and x_start.requires_grad:
            return list(_DiffuseTraceFn.apply(self, x_start, mask))
        return self._diffuse_trace(x_start, mask)

    def _diffuse_trace(self, x_start, mask):
        B, L, M = x_start.shape
        keep = (~mask).to(torch.uint8).contiguous()
        first = self.norm_spec(x_start).clamp_(-1.0, 1.0) * (~mask).unsqueeze(-1)
        trace = [first]
        for i in range(self.num_timesteps):
            t = torch.full((B,), i, device=x_start.device, dtype=torch.long)
            x = self.diffuse_fn(x_start, t, keep=keep)
            trace.append(ops.transpose_bml(self._bml(x), True))
        return trace

    # HotPathTrainer sets this around its D-phase forward: the next (grad-enabled) forward is launched together with it
    # (Denoiser.run_pair).  Off by default: it draws that second forward's randomness early.  pair_inputs: the
    # (cond, spk_emb, coarse_mel) that grad-enabled forward will be called with (the reference's two model calls of a
    # step see different conditioners when the encoder has dropout, train.py:133,153); None = the same as this call's.
    pair_forward = False
    pair_inputs = None
    _pair_stash = None

    @staticmethod
    def _pair_key(mel, cond, spk, mel_mask, coarse_mel):
        return (mel.data_ptr(), mel._version, cond.data_ptr(), cond._version, mel_mask.data_ptr(), tuple(mel.shape),
                None if coarse_mel is None else coarse_mel.data_ptr(), None if spk is None else spk.data_ptr())

    def forward(self, mel, cond, spk_emb, mel_mask, coarse_mel=None, clip_denoised=True):
        """model/diffusion.py:187-226.  mel [B,L,M]|None, cond [B,L,H], mel_mask bool [B,L] True = pad."""
        B = cond.shape[0]
        dev = cond.device
        if dev.type != "cuda":
            raise _lib.MixganHipError("GaussianDiffusion.forward on %s: the HIP path has no CPU fallback" % dev)
        x_t = x_t_prev = x_t_prev_pred = t = None
        keep = (~mel_mask).to(torch.uint8).contiguous()
        spk = spk_emb.contiguous() if spk_emb is not None else None
        grad = torch.is_grad_enabled() and mel is not None and (
            cond.requires_grad or (spk is not None and spk.requires_grad)
            or any(p.requires_grad for p in self.denoise_fn.parameters()))
        if grad:
            from .autograd import transpose_to_bml
            cond_t = transpose_to_bml(cond.contiguous())
        else:
            cond_t = ops.transpose_bml(cond.detach().contiguous(), False)
        self.cond = cond_t.detach()
        self.spk_emb = spk.detach() if spk is not None else None
        buf = self._buf()
        if mel is None:
            if self.model != "shallow":
                noise = None
            else:
                t = torch.full((B,), self.num_timesteps - 1, device=dev, dtype=torch.long)
                noise = self.diffuse_fn(coarse_mel, t, keep=keep)
            # final mel: denorm + [B,M,L]->[B,L,M] + mask multiply (:164,:200) fused in one transpose kernel
            x_0_pred = self.sampling(noise=noise, keep_trace=False, use_graph=self.use_graph, _final_keep=keep)[-1]
            return x_0_pred, x_t, x_t_prev, x_t_prev_pred, t
        M, L = mel.shape[2], mel.shape[1]
        melc = mel.contiguous()

        def draw():   # the four random draws of one forward, in the reference's order (model/diffusion.py:203-209)
            t_ = self._randint(B, dev)
            x_t_ = ops.diffuse(melc, t_, self._bml(self._randn((B, 1, M, L), dev)), keep, buf)
            x_prev_ = ops.diffuse(melc, (t_ - 1).contiguous(), self._bml(self._randn((B, 1, M, L), dev)), keep, buf)
            return t_, x_t_, x_prev_, self._bml(self._randn((B, 1, M, L), dev))
        pair_key = self._pair_key(mel, cond, spk, mel_mask, coarse_mel)
        stash, self._pair_stash = self._pair_stash, None
        if grad:
            from .autograd import denoise_and_posterior
            pre = None
            if (stash is not None and stash["key"] == pair_key
                    and self.denoise_fn.packed_weights(with_backward=True) is stash["packed"]
                    and self.denoise_fn._packed_key == stash["packed_key"]):
                # this forward was already launched next to the previous no-grad one (pair_forward): adopt its draws,
                # its output and its saved activations
                t, x_t_b, x_prev_b, post_noise = stash["draws"]
                pre = (stash["x0"], stash["ws"]) if stash["ws"] is not None else None
            else:
                if stash is not None and stash["ws"] is not None:
                    stash["ws"]._mg_busy = False      # never used: hand the workspace back
                t, x_t_b, x_prev_b, post_noise = draw()
            x0c, xpp = denoise_and_posterior(self, x_t_b, t, cond_t, spk, post_noise, keep, clip_denoised,
                                             coarse_mel if self.model == "shallow" else None, pre=pre)
            from .autograd import transpose_to_blm
            return (transpose_to_blm(x0c), ops.transpose_bml(x_t_b, True), ops.transpose_bml(x_prev_b, True),
                    transpose_to_blm(xpp), t)
        if stash is not None and stash["ws"] is not None:
            stash["ws"]._mg_busy = False
        t, x_t_b, x_prev_b, post_noise = draw()
        x0 = None
        if self.pair_forward and self.denoise_fn.precision == "fp32":
            # The GAN step runs this forward twice on the same weights -- here without gradient (train.py:133), then
            # with (train.py:153), with fresh t / noise.  Draw the second set now (nothing else consumes randomness in
            # between) and run both in one launch; the grad-enabled call that follows picks its half up above.
            draws2 = draw()
            cond_g, spk_g, coarse_g = self.pair_inputs if self.pair_inputs is not None else (cond, spk_emb, coarse_mel)
            spk_g = spk_g.contiguous() if spk_g is not None else None
            cond_t2 = spk2 = None
            if cond_g is not cond:
                if cond_g.shape != cond.shape:
                    raise _lib.MixganHipError("pair_inputs: the two phases' conditioners differ in shape")
                cond_t2 = ops.transpose_bml(cond_g.detach().contiguous(), False)
            if spk_g is not None and spk is not None and spk_g.data_ptr() != spk.data_ptr():
                spk2 = spk_g.detach()
            both = self.denoise_fn.run_pair(x_t_b, t, draws2[1], draws2[0], cond_t, spk, cond_t2, spk2)
            if both is not None:
                x0 = both[0]
            # (not a single-launch shape: the second forward still uses these draws, so that the random stream is
            # consumed in the same order either way)
            pair_key = self._pair_key(mel, cond_g, spk_g, mel_mask, coarse_g)
            self._pair_stash = {"key": pair_key, "draws": draws2, "x0": None if both is None else both[1],
                                "ws": None if both is None else both[2],
                                "packed": self.denoise_fn._packed, "packed_key": self.denoise_fn._packed_key}
        if x0 is None:
            x0 = self.denoise_fn.run(x_t_b, t, cond_t, spk)
        if self.model != "shallow":
            xpp, x0c = ops.posterior_sample(x0, x_t_b, t, post_noise, keep, buf, clip=clip_denoised, want_x0c=True)
        else:
            # shallow: the posterior starts from the (detached) coarse mel, x0 is only masked+clamped
            start = ops.transpose_bml(coarse_mel.detach().contiguous(), False, 1, self.spec_min, self.spec_max)
            xpp = ops.posterior_sample(start, x_t_b, t, post_noise, keep, buf, clip=False)
            _, x0c = ops.posterior_sample(x0, x_t_b, t, post_noise, keep, buf, clip=clip_denoised, want_x0c=True)
        tb = lambda a: ops.transpose_bml(a, True)
        return tb(x0c), tb(x_t_b), tb(x_prev_b), tb(xpp), t


class _DiffuseTraceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, diff, x_start, mask):
        x = x_start.detach().contiguous()
        trace = diff._diffuse_trace(x, mask)
        ctx.diff = diff
        ctx.save_for_backward(x, (~mask).to(torch.uint8).contiguous())
        return tuple(trace)

    @staticmethod
    def backward(ctx, *gs):
        x, keep = ctx.saved_tensors
        diff = ctx.diff
        B, L, M = x.shape
        T = diff.num_timesteps
        g = torch.stack([gi if gi is not None else torch.zeros_like(x) for gi in gs]).contiguous()
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().mg_diffuse_trace_bwd(
            _lib.fptr(g), _lib.fptr(x), _lib.fptr(diff.spec_min.contiguous()), _lib.fptr(diff.spec_max.contiguous()),
            _lib.iptr(keep, torch.uint8), _lib.fptr(diff._buf()["sqrt_alphas_cumprod"].contiguous()), _lib.fptr(dx),
            T, B, L, M, _lib.stream_ptr()))
        return None, dx, None
