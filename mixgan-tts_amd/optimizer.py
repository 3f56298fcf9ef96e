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
    def step(self):
        self.current_step += 1
        self.last_lr = self.lr_at(self.current_step)
        for group in self._optimizer.param_groups:
            group["lr"] = self.last_lr
        self._optimizer.step()
        return self.last_lr

    def zero_grad(self):
        self._optimizer.zero_grad()

    def state_dict(self):
        return self._optimizer.state_dict()

    def load_state_dict(self, adam_state):
        self._optimizer.load_state_dict(adam_state)
