"""Learning-rate schedule wrapper of aux pre-training (`model/optimizer.py:5-56`, used by `train.py:75-85` as
`optG_fs2`): Adam whose lr follows  d_model^-0.5 * min(step^-0.5, step * warmup^-1.5) * anneal_rate^(#passed
anneal steps).  Same constructor, methods and checkpoint behaviour (`load_state_dict` takes the inner Adam's
state dict, `utils/model.py:24-31`)."""
import numpy as np
import torch


class ScheduledOptim:
    def __init__(self, model, train_config, model_config, current_step):
        cfg = train_config["optimizer_fs2"]
        self._optimizer = torch.optim.Adam(model.parameters(), betas=cfg["betas"], eps=cfg["eps"],
                                           weight_decay=cfg["weight_decay"])
        self.n_warmup_steps = cfg["warm_up_step"]
        self.anneal_steps = cfg["anneal_steps"]
        self.anneal_rate = cfg["anneal_rate"]
        self.current_step = current_step
        self.last_lr = self.init_lr = np.power(model_config["transformer"]["encoder_hidden"], -0.5)

    def get_last_lr(self):
        return self.last_lr

    def step(self):
        lr = self._update_learning_rate()
        self._optimizer.step()
        return lr

    def zero_grad(self):
        self._optimizer.zero_grad()

    def load_state_dict(self, state):
        self._optimizer.load_state_dict(state)

    def state_dict(self):
        return self._optimizer.state_dict()

    def _get_lr_scale(self):
        lr = np.min([np.power(self.current_step, -0.5), np.power(self.n_warmup_steps, -1.5) * self.current_step])
        for s in self.anneal_steps:
            if self.current_step > s:
                lr = lr * self.anneal_rate
        return lr

    def _update_learning_rate(self):
        self.current_step += 1
        self.last_lr = lr = self.init_lr * self._get_lr_scale()
        for group in self._optimizer.param_groups:
            group["lr"] = lr
        return lr
