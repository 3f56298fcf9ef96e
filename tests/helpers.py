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
    """Draw the weights of fixture module `name` exactly as tools/make_golden.py did."""
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
