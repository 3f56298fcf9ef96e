"""Shared helpers for the parity tests (oracle side)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle import weights as WR  # noqa: E402


def golden(name):
    return {k: v for k, v in np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False).items()}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def seeded(manifest, name, seed, prefix="", requires_grad=False):
    """Draw the weights of fixture module `name` exactly as tests/golden/make_golden.py did."""
    w = WR.draw(manifest[name]["seeded"], seed)
    out = {}
    for k, a in w.items():
        t = torch.from_numpy(a)
        if requires_grad and not (k.endswith("running_mean") or k.endswith("running_var")):
            t.requires_grad_()
        out[prefix + k] = t
    return out, WR.checksum(w)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def assert_close(a, b, tol, what=""):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    e = rel_err(a, b)
    assert e <= tol, "%s: max-abs err / max-abs ref = %.3e > %.1e" % (what, e, tol)


def digest(g):
    g = g.detach().double().cpu()
    corner = g[tuple(slice(0, min(4, s)) for s in g.shape)].float().numpy()
    return np.array([g.sum().item(), g.abs().sum().item()]), corner


def assert_digest(g, gold, key, tol):
    """Compare a gradient against the (sum, abs-sum, corner) digest stored in the fixture."""
    d, corner = digest(g)
    ref = gold["dw_sum/" + key]
    scale = abs(ref[1]) + 1e-30
    assert abs(d[0] - ref[0]) / scale <= tol, (key, d, ref)
    assert abs(d[1] - ref[1]) / scale <= tol, (key, d, ref)
    if ("dw_corner/" + key) in gold:
        c = gold["dw_corner/" + key]
        # corner tolerance is relative to the tensor's mean magnitude
        mean_mag = ref[1] / max(1, g.numel())
        assert np.abs(corner - c).max() <= tol * max(np.abs(c).max(), mean_mag) * 4, (key,)


# ----------------------------------------------------------------------------------------------
# Hot-path configuration keys (SURVEY.md section 5 "Config / flags"), LJSpeech values.
def hot_path_configs(model="naive", T=4, mode="vpsde", multi_speaker=False, stats_dir=None, max_seq_len=1000):
    import types
    pre = {"preprocessing": {"mel": {"n_mel_channels": 80}, "speaker_embedder": "none"},
           "path": {"preprocessed_path": stats_dir}}
    mc = {
        "transformer": {"encoder_hidden": 256, "decoder_layer": 6, "decoder_head": 2, "decoder_hidden": 256,
                        "conv_filter_size": 1024, "conv_kernel_size": 9, "decoder_dropout": 0.2},
        "denoiser": {"denoiser_hidden": 512, "denoiser_dropout": 0.2, "residual_layers": 20, "residual_channels": 256,
                     "noise_schedule_naive": mode, "timesteps": T, "shallow_timesteps": T, "min_beta": 0.1,
                     "max_beta": 40, "s": 0.008, "keep_bins": 80},
        "discriminator": {"n_layer": 3, "n_uncond_layer": 2, "n_cond_layer": 2, "n_channels": [64, 128, 512, 128, 1],
                          "kernel_sizes": [3, 5, 5, 5, 3], "strides": [1, 2, 2, 1, 1]},
        "multi_speaker": multi_speaker, "max_seq_len": max_seq_len,
    }
    tr = {"loss": {"noise_loss": "l1", "adv_loss_mode": "lsgan", "lambda_fm": 10.0, "lambda_fm_shallow": 0.001},
          "optimizer": {"batch_size": 8, "betas": [0.5, 0.9], "gamma": 0.999, "grad_clip_thresh": 1, "grad_acc_step": 1,
                        "init_lr_G": 0.0001, "init_lr_D": 0.0002}}
    return types.SimpleNamespace(model=model), pre, mc, tr


def write_stats(tmpdir, spec_min, spec_max, n_speakers=0):
    import json
    with open(os.path.join(str(tmpdir), "stats.json"), "w") as f:
        json.dump({"pitch": [-2.0, 8.0, 0.0, 1.0], "energy": [-1.5, 7.0, 0.0, 1.0],
                   "spec_min": [float(v) for v in spec_min], "spec_max": [float(v) for v in spec_max]}, f)
    if n_speakers:
        with open(os.path.join(str(tmpdir), "speakers.json"), "w") as f:
            json.dump({"spk%d" % i: i for i in range(n_speakers)}, f)
    return str(tmpdir)


def load_seeded(module, manifest, name, seed, prefix=""):
    """Seed a product module with the fixture recipe through load_state_dict (checks key parity)."""
    w = WR.draw(manifest[name]["seeded"], seed)
    sd = module.state_dict()
    for k, a in w.items():
        assert prefix + k in sd, "missing state_dict key " + prefix + k
        assert tuple(sd[prefix + k].shape) == a.shape, (k, tuple(sd[prefix + k].shape), a.shape)
        sd[prefix + k] = torch.from_numpy(a)
    module.load_state_dict(sd)
    return WR.checksum(w)


class Tape:
    """noise_fn / t_fn stand-in replaying fixture tensors in call order."""

    def __init__(self, items):
        self.items = [torch.from_numpy(np.ascontiguousarray(a)) for a in items]
        self.i = 0

    def __call__(self, shape):
        x = self.items[self.i]
        self.i += 1
        assert tuple(x.shape) == tuple(shape), (tuple(x.shape), tuple(shape))
        return x


# ----------------------------------------------------------------------------------------------
# (a16) MixGANTTS.forward fixtures (tests/golden/mixgantts_*.npz): the recorded outputs of the reference's
# LinguisticEncoder, the 16 slots + p_targets + coarse_mels with their None-ness / requires_grad flags.
MIXGANTTS_CASES = [("naive", 0, True), ("naive", 1, True), ("naive", 0, False), ("shallow", 0, True),
                   ("shallow", 0, False), ("aux", 0, True)]


def mixgantts_case_name(model, ms, train):
    return "mixgantts_%s_ms%d_%s" % (model, ms, "train" if train else "infer")


def mixgantts_encoder_outputs(g, train, device=None):
    """The nine outputs of model/linguistic_encoder.py:373-383 as recorded; in training the float ones require grad,
    as they do coming out of the reference's encoder."""
    def one(key):
        t = T(g[key])
        if device is not None:
            t = t.to(device)
        if train and t.is_floating_point():
            t.requires_grad_()
        return t
    out = []
    for i in range(9):
        if ("enc/%d" % i) in g:
            out.append(one("enc/%d" % i))
        else:
            n = len([k for k in g if k.startswith("enc/%d/" % i)])
            out.append([one("enc/%d/%d" % (i, j)) for j in range(n)])
    return tuple(out)


def mixgantts_leaves(out, p_targets, coarse):
    """name -> tensor|None for every leaf of MixGANTTS.forward's return value (names as in the fixture)."""
    d = {}
    for i, o in enumerate(out):
        if isinstance(o, (list, tuple)):
            for j, oo in enumerate(o):
                d["slot%02d/%d" % (i, j)] = oo
        else:
            d["slot%02d" % i] = o
    d["p_targets"], d["coarse_mels"] = p_targets, coarse
    return d


def assert_mixgantts_slots(leaves, g, tol, check_flags):
    """Values, None-ness and (in training) requires_grad of every leaf against the fixture."""
    flags = dict(zip([str(k) for k in g["flag_names"]], [int(v) for v in g["flags"]]))
    assert sorted(leaves) == sorted(flags), (sorted(leaves), sorted(flags))
    for k, v in leaves.items():
        if flags[k] < 0:
            assert v is None, k + " should be None"
            continue
        assert v is not None, k + " is None"
        ref = g[k]
        a = v.detach().cpu().numpy()
        assert a.shape == ref.shape and str(a.dtype) == str(ref.dtype), (k, a.shape, a.dtype, ref.shape, ref.dtype)
        if a.dtype.kind == "f":
            fin = np.isfinite(ref)                 # encoder pass-throughs hold log(0) = -inf for padded words
            assert np.array_equal(a[~fin], ref[~fin], equal_nan=True), k
            assert_close(np.where(fin, a, 0), np.where(fin, ref, 0), tol, k)
        else:
            assert (a == ref).all(), k
        if check_flags:
            assert int(v.requires_grad) == flags[k], "%s.requires_grad = %s, reference %d" % (k, v.requires_grad, flags[k])


def mixgantts_tapes(g):
    """(rng draws in call order, dropout keep-masks in call order) of the fixture."""
    n = len([k for k in g if k.startswith("rng")])
    m = len([k for k in g if k.startswith("mask")])
    return [g["rng%d" % i] for i in range(n)], [g["mask%d" % i] for i in range(m)]
