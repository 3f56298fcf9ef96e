"""Replayable weight recipe shared by tests/golden/make_golden.py (reference side) and the tests.

The reference ships no checkpoints, so parity fixtures use seeded random
weights.  To keep the fixtures small the weights themselves are never committed: both sides
draw them from the same numpy stream.  The recipe is a function of the *sorted*
`{state_dict key: shape}` manifest only, so it doubles as a check that our modules expose
the reference's exact checkpoint keys ("Checkpoint / resume").

Test infrastructure (see oracle/__init__.py).
"""
import numpy as np


def _kind(key):
    if key.endswith("running_var"):
        return "var"
    if key.endswith("running_mean"):
        return "bias"
    if "layer_norm.weight" in key or (".1.weight" in key and "convolutions" in key):
        return "gain"       # LayerNorm / BatchNorm scale
    return "w"


def draw(manifest, seed):
    """manifest: {key: shape tuple}.  Returns {key: float32 ndarray}.

    dim >= 2 : N(0, 1/fan_in) with fan_in = prod(shape[1:])  (keeps activations O(1) over
               the 20 residual layers, and makes the zero-initialised output projection of
               model/modules.py:418 non-trivial);
    1-D bias : 0.1 * N(0,1);  norm gains: 1 + 0.1 * N(0,1);  running_var: 0.5 + |N(0,1)|.
    """
    rng = np.random.default_rng(seed)
    out = {}
    for key in sorted(manifest):
        shape = tuple(int(s) for s in manifest[key])
        z = rng.standard_normal(shape)
        kind = _kind(key)
        if kind == "var":
            a = 0.5 + np.abs(z)
        elif kind == "gain":
            a = 1.0 + 0.1 * z
        elif len(shape) >= 2:
            a = z / np.sqrt(float(np.prod(shape[1:])))
        else:
            a = 0.1 * z
        out[key] = a.astype(np.float32)
    return out


def checksum(weights):
    """fp64 (sum, abs-sum) over all tensors in sorted-key order."""
    s = 0.0
    a = 0.0
    for k in sorted(weights):
        w = np.asarray(weights[k], dtype=np.float64)
        s += float(w.sum())
        a += float(np.abs(w).sum())
    return np.array([s, a], dtype=np.float64)
