"""Warm-up / anneal learning-rate schedule of aux pre-training, wrapped around Adam.

Drop-in for the object `utils/model.py:36` builds as `optG_fs2` (reference class: `model/optimizer.py:5-56`) --
the surface `train.py` touches is kept: `step()` (returns the lr it just applied), `zero_grad()`,
`get_last_lr()`, `load_state_dict(adam_state)`, and the inner optimizer under `._optimizer`
(`train.py:257` checkpoints `optG_fs2._optimizer.state_dict()`).

Schedule, with s the 1-based count of `step()` calls (continuing from `current_step` on restore), d the encoder
width, w the warm-up length and (a_i) the anneal boundaries:

    lr(s) = d^-0.5 * min(s^-0.5, s * w^-1.5) * rate^(#{i : s > a_i})

i.e. linear warm-up to d^-0.5 * w^-0.5 at s = w, inverse-square-root decay after it, and a multiplicative drop at
every boundary passed.
"""
import torch


def noam_annealed_lr(step, d_model, warmup, anneal_steps=(), anneal_rate=1.0):
    """lr(s) of the docstring above for one step index (pure function; float64 like the reference's numpy)."""
    scale = min(float(step) ** -0.5, float(step) * float(warmup) ** -1.5)
    drops = sum(1 for boundary in anneal_steps if step > boundary)
    return float(d_model) ** -0.5 * scale * float(anneal_rate) ** drops if drops else float(d_model) ** -0.5 * scale


class ScheduledOptim:
    def __init__(self, model, train_config, model_config, current_step):
        hp = train_config["optimizer_fs2"]
        self._optimizer = torch.optim.Adam(model.parameters(), betas=hp["betas"], eps=hp["eps"],
                                           weight_decay=hp["weight_decay"])
        self._d_model = model_config["transformer"]["encoder_hidden"]
        self.n_warmup_steps = hp["warm_up_step"]
        self.anneal_steps = tuple(hp["anneal_steps"])
        self.anneal_rate = hp["anneal_rate"]
        self.current_step = current_step
        self.init_lr = float(self._d_model) ** -0.5
        self.last_lr = self.init_lr

    # -- schedule
    def lr_at(self, step):
        return noam_annealed_lr(step, self._d_model, self.n_warmup_steps, self.anneal_steps, self.anneal_rate)

    def get_last_lr(self):
        return self.last_lr

    # -- optimizer surface used by train.py
    def step(self, **inner):
        """inner: keyword arguments for the wrapped optimizer's step (FlatAdam's max_grad_norm)."""
        self.current_step += 1
        self.last_lr = self.lr_at(self.current_step)
        for group in self._optimizer.param_groups:
            group["lr"] = self.last_lr
        self._optimizer.step(**inner)
        return self.last_lr

    def use_flat(self, bucket):
        """Swap the inner torch.optim.Adam for a FlatAdam over `bucket` (same hyper-parameters, state taken over): the
        `optG_fs2._optimizer.state_dict()` that train.py:257 checkpoints keeps its layout and parameter order."""
        old = self._optimizer
        g = old.param_groups[0]
        order = [p for p in g["params"] if any(p is q for q in bucket.params)]
        flat = FlatAdam(bucket, lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"],
                        param_order=order if len(order) == len(bucket.params) else None)
        self._optimizer = flat.adopt(old)
        return self

    def zero_grad(self):
        self._optimizer.zero_grad()

    def state_dict(self):
        return self._optimizer.state_dict()

    def load_state_dict(self, adam_state):
        self._optimizer.load_state_dict(adam_state)


class FlatAdam(torch.optim.Adam):
    """torch.optim.Adam (utils/model.py:32-40) for parameters whose gradients sit in a `GradBucket`: the parameters
    and both moments are re-homed into flat buffers with the bucket's layout, so that `clip_grad_norm_` + `step()` of
    train.py:81-83 become two HIP launches (ops.grad_norm, ops.adam_flat) at 32 bytes of HBM traffic per parameter,
    instead of a norm, a scale and ~10 multi-tensor launches.  Same update rule, same `state_dict()` layout (per
    parameter `step`, `exp_avg`, `exp_avg_sq` -- here views of the flat moments), same `param_groups` (lr schedulers
    work unchanged).  Build it after the module is on its device; `.to()` afterwards would undo the aliasing.

    One deliberate difference from torch.optim.Adam: every parameter of the bucket is stepped on every call, with a
    zero gradient where autograd produced none (GradBucket.gather zero-fills those slices so that all ranks reduce the
    same buffer and cannot drift apart).  A stock Adam skips a parameter whose .grad is None -- its moments and step
    count stand still -- whereas here such a parameter's moments decay and it keeps moving along its first moment, and
    `state_dict()` reports the shared step count for it.  Every parameter of the hot path receives a gradient in every
    step of train.py:131-184 (the denoiser's speaker projections only in multi-speaker models, where they exist), so the
    two rules coincide on this path; a caller that steps conditionally-unused parameters through this class should know."""

    def __init__(self, bucket, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, param_order=None):
        """param_order: the parameter list a stock optimizer would have been built over (e.g. `model.parameters()`,
        utils/model.py:33) -- it fixes the index order of `state_dict()`, which is how checkpoints address the state;
        the flat layout follows the bucket either way.  Must contain every parameter of the bucket (extras, e.g. frozen
        parameters, keep their index but are never stepped)."""
        order = list(param_order) if param_order is not None else list(bucket.params)
        # a superset is fine: parameters outside the bucket (requires_grad=False -- the decoder's position table --
        # sit in the reference's Adam too, without ever getting state) only hold their index in state_dict()
        if not {id(p) for p in bucket.params} <= {id(p) for p in order} or len({id(p) for p in order}) != len(order):
            raise ValueError("param_order must list every parameter of the bucket, each once")
        super().__init__(order, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.bucket = bucket
        flat = bucket.flat
        if not flat.is_cuda:
            raise ValueError("FlatAdam runs on the HIP path; keep torch.optim.Adam for CPU parameters")
        self.flat_p = torch.empty_like(flat)
        self.flat_m = torch.zeros_like(flat)
        self.flat_v = torch.zeros_like(flat)
        self._norm = torch.zeros(2, device=flat.device, dtype=torch.float32)
        self._scratch = None
        self._steps = torch.tensor(0.0)                   # shared by every parameter's state (they step together)
        self._hyper = None                                # device pair read by the kernel (enable_device_hyper)
        for p in bucket.params:
            off, n = bucket.offsets[id(p)], p.numel()
            home = self.flat_p[off:off + n].view_as(p)
            home.copy_(p.data)
            p.data = home
            self.state[p] = {"step": self._steps, "exp_avg": self.flat_m[off:off + n].view_as(p),
                             "exp_avg_sq": self.flat_v[off:off + n].view_as(p)}

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        """Adam step from bucket.flat (call bucket.gather()/all_reduce_mean() first).  max_grad_norm: clip the global
        gradient norm first, like clip_grad_norm_(params, max_grad_norm) -- the factor is applied inside the update;
        bucket.flat itself is left unscaled.  Returns the device tensor (norm, factor) when clipping."""
        from . import ops
        if closure is not None:
            raise NotImplementedError("FlatAdam.step takes no closure")
        g = self.param_groups[0]
        if len(self.param_groups) != 1 or g.get("amsgrad") or g.get("maximize"):
            raise NotImplementedError("FlatAdam: one parameter group, no amsgrad / maximize")
        scale = None
        if max_grad_norm is not None:
            if self._scratch is None:
                self._scratch = torch.empty(ops._lib.lib().mg_grad_norm_scratch_floats(), device=self.flat_p.device)
            ops.grad_norm(self.bucket.flat, max_grad_norm, out=self._norm, scratch=self._scratch)
            scale = self._norm[1:]
        self._steps += 1
        if self._hyper is not None and not torch.cuda.is_current_stream_capturing():
            self.write_hyper()                        # (under capture the replaying side writes it before every replay)
        ops.adam_flat(self.flat_p, self.bucket.flat, self.flat_m, self.flat_v, g["lr"], g["betas"], g["eps"],
                      g["weight_decay"], int(self._steps.item()), scale, hyper=self._hyper)
        torch.autograd.graph.increment_version(self.bucket.params)    # derived caches (packed weights) key on it
        return self._norm if max_grad_norm is not None else None

    def enable_device_hyper(self):
        """From now on the update kernel reads lr / (1 - beta1^t) and 1 / sqrt(1 - beta2^t) from device memory (written
        by write_hyper) instead of taking them as launch arguments: a hipGraph that contains step() then replays with the
        step count and learning rate of the moment (HotPathTrainer.capture)."""
        if self._hyper is None:
            self._hyper = torch.zeros(2, device=self.flat_p.device, dtype=torch.float32)
            self._hyper_host = torch.zeros(2, dtype=torch.float32).pin_memory()
        return self

    def write_hyper(self, step=None):
        """Host -> device: the two scalars of the step about to run (step: 1-based count; default: the current one)."""
        g = self.param_groups[0]
        t = float(self._steps) if step is None else float(step)
        b1, b2 = g["betas"]
        self._hyper_host[0] = g["lr"] / (1.0 - b1 ** t)
        self._hyper_host[1] = 1.0 / (1.0 - b2 ** t) ** 0.5
        self._hyper.copy_(self._hyper_host, non_blocking=True)

    def adopt(self, adam):
        """Continue from a stock torch.optim.Adam over (a superset of) the same parameter objects -- e.g. the optG /
        optD `get_model(..., train=True)` returns with a checkpoint's state loaded (utils/model.py:41-46): moments and
        step count are copied by parameter identity, hyper-parameters from its first group."""
        g0 = adam.param_groups[0]
        for k in ("lr", "betas", "eps", "weight_decay"):
            self.param_groups[0][k] = g0[k]
        if "initial_lr" in g0:
            self.param_groups[0]["initial_lr"] = g0["initial_lr"]
        steps = 0.0
        with torch.no_grad():
            for p in self.bucket.params:
                st = adam.state.get(p)
                if st:
                    self.state[p]["exp_avg"].copy_(st["exp_avg"])
                    self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
                    steps = max(steps, float(st["step"]))
        self._steps.fill_(steps)
        return self

    def state_dict(self):
        """torch.optim.Adam's layout.  Every parameter gets its OWN `step` tensor: a stock Adam that loads this
        increments the step of each parameter separately, and must not find them aliased."""
        sd = super().state_dict()
        sd["state"] = {k: dict(st, step=st["step"].clone()) for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        """Accepts a torch.optim.Adam state dict (the reference's checkpoints, train.py:252-267): the moments are
        copied into the flat buffers and the state re-pointed at them."""
        super().load_state_dict(state_dict)
        steps = 0.0
        for p in self.bucket.params:
            st = self.state.get(p)
            off, n = self.bucket.offsets[id(p)], p.numel()
            m, v = self.flat_m[off:off + n].view_as(p), self.flat_v[off:off + n].view_as(p)
            if st:
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                steps = max(steps, float(st["step"]))
            else:
                m.zero_()
                v.zero_()
            self.state[p] = {"step": self._steps, "exp_avg": m, "exp_avg_sq": v}
        self._steps.fill_(steps)
