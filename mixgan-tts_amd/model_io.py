"""Model / optimizer construction and checkpoint restore: the `get_model` surface of `utils/model.py:12-67`
(called by `train.py:42`, `synthesize.py:245`, `evaluate.py`), plus the matching save of `train.py:252-267`.

Checkpoint format (unchanged): `<train_config.path.ckpt_path>/<step>.pth.tar`, a dict with keys `epoch`, `G`, `D`,
`optG_fs2`, `optG`, `optD`, `sdlG`, `sdlD` (state dicts).  It is read with `torch.load(..., weights_only=True)`: the
file only holds tensors and plain containers, and nothing in it is ever executed.

The linguistic encoder is out of scope and injected (`linguistic_encoder=`): with it the generator's state dict has
the reference's full key set and `G` loads strictly; without it only the keys of the modules on the HIP path are
restored (and the call says which were skipped).
"""
import os

import torch

from .discriminator import JCUDiscriminator
from .mixgantts import MixGANTTS
from .optimizer import ScheduledOptim

CKPT_KEYS = ("epoch", "G", "D", "optG_fs2", "optG", "optD", "sdlG", "sdlD")


def checkpoint_path(train_config, step):
    return os.path.join(train_config["path"]["ckpt_path"], "{}.pth.tar".format(step))


def get_model(args, configs, device, train=False, linguistic_encoder=None):
    """-> model (eval) or (model, discriminator, optG_fs2, optG, optD, sdlG, sdlD, epoch) when `train`."""
    preprocess_config, model_config, train_config = configs
    epoch = 1
    model = MixGANTTS(args, preprocess_config, model_config, train_config, linguistic_encoder=linguistic_encoder).to(device)
    discriminator = JCUDiscriminator(preprocess_config, model_config, train_config).to(device)
    ckpt = None
    if args.restore_step:
        ckpt = torch.load(checkpoint_path(train_config, args.restore_step), map_location=device, weights_only=True)
        epoch = int(ckpt["epoch"])
        if linguistic_encoder is not None:
            model.load_state_dict(ckpt["G"])
        else:
            own = model.state_dict()
            skipped = sorted(k for k in ckpt["G"] if k not in own)
            model.load_state_dict({k: v for k, v in ckpt["G"].items() if k in own}, strict=True)
            model.skipped_checkpoint_keys = skipped       # the out-of-scope encoder's parameters
        discriminator.load_state_dict(ckpt["D"])
    if not train:
        model.eval()
        model.requires_grad_ = False     # (sic) utils/model.py:52 assigns the attribute instead of calling it
        return model
    oc = train_config["optimizer"]
    optG_fs2 = ScheduledOptim(model, train_config, model_config, args.restore_step)
    optG = torch.optim.Adam(model.parameters(), lr=oc["init_lr_G"], betas=oc["betas"])
    optD = torch.optim.Adam(discriminator.parameters(), lr=oc["init_lr_D"], betas=oc["betas"])
    sdlG = torch.optim.lr_scheduler.ExponentialLR(optG, oc["gamma"])
    sdlD = torch.optim.lr_scheduler.ExponentialLR(optD, oc["gamma"])
    # the aux -> shallow hand-over restarts the optimizers (utils/model.py:41)
    if ckpt is not None and args.restore_step != train_config["step"]["total_step_aux"]:
        optG_fs2.load_state_dict(ckpt["optG_fs2"])
        optG.load_state_dict(ckpt["optG"])
        optD.load_state_dict(ckpt["optD"])
        sdlG.load_state_dict(ckpt["sdlG"])
        sdlD.load_state_dict(ckpt["sdlD"])
    model.train()
    discriminator.train()
    return model, discriminator, optG_fs2, optG, optD, sdlG, sdlD, epoch


def save_checkpoint(train_config, step, epoch, model, discriminator, optG_fs2, optG, optD, sdlG, sdlD):
    """train.py:252-267 (without the DataParallel `.module` hop: one process per GPU, rank 0 saves)."""
    path = checkpoint_path(train_config, step)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"epoch": epoch, "G": model.state_dict(), "D": discriminator.state_dict(),
                "optG_fs2": optG_fs2._optimizer.state_dict(), "optG": optG.state_dict(), "optD": optD.state_dict(),
                "sdlG": sdlG.state_dict(), "sdlD": sdlD.state_dict()}, path)
    return path


def get_param_num(model):
    return sum(p.numel() for p in model.parameters())
