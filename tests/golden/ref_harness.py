"""Import harness for the read-only reference at /root/reference (THIS CONTAINER ONLY).

Used only by tests/golden/make_golden.py to generate the committed fixtures under tests/golden/.
Nothing under tests/, bench.py or the package imports this file: /root/reference does not
exist on the GPU box.

The reference's `model/__init__.py` eagerly imports the whole product, including
third-party modules that are absent here and that the hot path never touches
(SURVEY.md §8c).  They are replaced by empty stub modules before import.
"""
import os
import sys
import types
import json
import tempfile

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    sys.dont_write_bytecode = True
    _stub("unidecode", unidecode=lambda s: s)
    _stub("inflect", engine=lambda: None)
    _stub("librosa")
    _stub("parselmouth")
    pw = _stub("pycwt")
    pw.wavelet = _stub("pycwt.wavelet")
    ds = _stub("deepspeaker")
    ds.embedding = _stub("deepspeaker.embedding")
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load_configs(dataset="LJSpeech"):
    import yaml
    out = []
    for n in ("preprocess", "model", "train"):
        with open(os.path.join(REF, "config", dataset, n + ".yaml")) as f:
            out.append(yaml.load(f, Loader=yaml.SafeLoader))
    return out


def make_stats_dir(spec_min, spec_max, n_speakers=0):
    d = tempfile.mkdtemp(prefix="mg_stats_")
    with open(os.path.join(d, "stats.json"), "w") as f:
        json.dump({"pitch": [-2.0, 8.0, 0.0, 1.0], "energy": [-1.5, 7.0, 0.0, 1.0],
                   "spec_min": list(map(float, spec_min)), "spec_max": list(map(float, spec_max))}, f)
    if n_speakers:
        with open(os.path.join(d, "speakers.json"), "w") as f:
            json.dump({"spk%d" % i: i for i in range(n_speakers)}, f)
    return d
