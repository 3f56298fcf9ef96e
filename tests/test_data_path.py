"""SURVEY.md section 8 f2 -- the data path (dataset.py:13-272, utils/tools.py:33-110,334-371) against a run of
the reference on a synthetic preprocessed_data tree (fixture dataset.npz, made by tests/golden/make_golden.py).
Bit-exact: values, shapes and dtypes of every batch slot.  CPU only."""
import json
import os
import types

import numpy as np
import pytest
import torch

from helpers import golden

import mixgan_tts_amd  # noqa: F401  (alias module)
from mixgan_tts_amd import data as D

N_ITEMS = 11
KINDS = ("mel", "pitch", "energy", "duration", "phones_per_word", "attn_prior")


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    g = golden("dataset")
    d = str(tmp_path_factory.mktemp("preprocessed"))
    spks = ["spkA", "spkB", "spkC"]
    with open(os.path.join(d, "speakers.json"), "w") as f:
        json.dump({s: i for i, s in enumerate(spks)}, f)
    for k in KINDS + ("spker_embed",):
        os.makedirs(os.path.join(d, k))
    for s in spks:
        np.save(os.path.join(d, "spker_embed", "%s-spker_embed.npy" % s), g["spk/" + s])
    lines = [str(x) for x in g["meta_lines"]]
    ids = {}
    for i, ln in enumerate(lines):
        base, spk, text, _ = ln.split("|")
        for k in KINDS:
            np.save(os.path.join(d, k, "%s-%s-%s.npy" % (spk, k, base)), g["item%02d/%s" % (i, k)])
        ids[text] = g["item%02d/phone_ids" % i]
    with open(os.path.join(d, "train.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")
    t2s = lambda text, cleaners: ids[text].tolist()  # noqa: E731  (the text front-end is out of scope: injected)
    pre = {"dataset": "Synth", "path": {"preprocessed_path": d},
           "preprocessing": {"text": {"text_cleaners": ["english_cleaners"]}, "speaker_embedder": "none"}}
    train = {"optimizer": {"batch_size": 4, "batch_size_shallow": 3}}
    return g, d, pre, train, t2s


def _check(g, prefix, batchs, torch_side=False):
    assert len(batchs) == int(g[prefix + "/n_batches"])
    for b, tup in enumerate(batchs):
        for j, x in enumerate(tup):
            key = "%s/b%d/s%02d" % (prefix, b, j)
            if x is None:
                assert key not in g, key
                continue
            ref = g[key]
            if torch.is_tensor(x):
                assert str(x.dtype) == str(g[key + "_torch_dtype"]), (key, x.dtype)
                x = x.cpu().numpy()
            elif torch_side:
                assert key + "_torch_dtype" not in g, key
            x = np.asarray(x)
            assert x.shape == ref.shape, (key, x.shape, ref.shape)
            if ref.dtype.kind in "US":
                assert [str(v) for v in x.ravel()] == [str(v) for v in ref.ravel()], key
            else:
                assert x.dtype == ref.dtype, (key, x.dtype, ref.dtype)
                assert np.array_equal(x, ref), key


def test_sorted_drop_last_collation_matches_reference(tree):
    g, d, pre, train, t2s = tree
    ds = D.Dataset("train.txt", types.SimpleNamespace(model="naive"), pre, {"multi_speaker": False}, train,
                   sort=True, drop_last=True, text_to_sequence=t2s)
    assert len(ds) == N_ITEMS
    _check(g, "sorted_drop", ds.collate_fn([ds[i] for i in range(N_ITEMS)]))


def test_unsorted_keep_tail_speaker_embeddings_and_to_device(tree):
    g, d, pre, train, t2s = tree
    pre2 = json.loads(json.dumps(pre))
    pre2["preprocessing"]["speaker_embedder"] = "DeepSpeaker"
    ds = D.Dataset("train.txt", types.SimpleNamespace(model="shallow"), pre2, {"multi_speaker": True}, train,
                   sort=False, drop_last=False, text_to_sequence=t2s, mmap=False)
    batchs = ds.collate_fn([ds[int(i)] for i in g["plain_keep/order"]])
    _check(g, "plain_keep", batchs)
    _check(g, "plain_keep_dev", [D.to_device(b, torch.device("cpu")) for b in batchs], torch_side=True)


def test_text_dataset(tree):
    g, d, pre, train, t2s = tree
    pre2 = json.loads(json.dumps(pre))
    pre2["preprocessing"]["speaker_embedder"] = "DeepSpeaker"
    tds = D.TextDataset(os.path.join(d, "train.txt"), pre2, {"multi_speaker": True}, text_to_sequence=t2s)
    b = tds.collate_fn([tds[int(i)] for i in g["text/order"]])
    _check(g, "text", [b])
    _check(g, "text_dev", [D.to_device(b, torch.device("cpu"))], torch_side=True)
    with pytest.raises(ValueError):
        D.to_device(b[:5], torch.device("cpu"))


def test_pad_edge_cases():
    assert D.pad_1D([np.array([1, 2, 3]), np.array([], dtype=np.int64)]).tolist() == [[1, 2, 3], [0, 0, 0]]
    with pytest.raises(ValueError):
        D.pad_2D([np.zeros((5, 80))], maxlen=4)
    with pytest.raises(ValueError):
        D.pad_2D([np.zeros((5, 80)), np.zeros((5, 81))])
    assert D.pad_3D([np.ones((2, 3))], 2, 4, 5).sum() == 6


def test_rank_shards_are_disjoint_equal_and_epoch_seeded():
    n, group, world = 103, 8, 4
    per_rank = []
    for r in range(world):
        s = D.RankShardSampler(n, group, r, world, seed=7)
        s.set_epoch(3)
        gs = list(s)
        assert len(gs) == len(s) == (n // group) // world
        assert all(len(x) == group for x in gs)
        per_rank.append([i for x in gs for i in x])
    flat = [i for r in per_rank for i in r]
    assert len(set(flat)) == len(flat)                       # disjoint
    assert len({len(r) for r in per_rank}) == 1              # same number of steps on every rank
    s0 = D.RankShardSampler(n, group, 0, world, seed=7)
    s0.set_epoch(4)
    assert [i for x in s0 for i in x] != per_rank[0]         # reshuffled per epoch
    s0.set_epoch(3)
    assert [i for x in s0 for i in x] == per_rank[0]         # ... deterministically
    with pytest.raises(ValueError):
        D.RankShardSampler(n, group, 4, 4)


@pytest.mark.parametrize("workers,depth", [(1, 1), (3, 2)])
def test_prefetch_loader_yields_the_collated_groups_in_order(tree, workers, depth):
    g, d, pre, train, t2s = tree
    ds = D.Dataset("train.txt", types.SimpleNamespace(model="naive"), pre, {"multi_speaker": False}, train,
                   sort=True, drop_last=True, text_to_sequence=t2s)
    smp = D.RankShardSampler(len(ds), 5, rank=1, world=2, seed=11)
    want = [[D.to_device(b, "cpu") for b in ds.collate_fn([ds[i] for i in idxs])] for idxs in smp]
    got = list(D.PrefetchLoader(ds, smp, "cpu", depth=depth, workers=workers))
    assert len(got) == len(want) == 1
    for gb, wb in zip(got, want):
        assert len(gb) == len(wb)
        for a, b in zip(gb, wb):
            for x, y in zip(a, b):
                if torch.is_tensor(y):
                    assert x.dtype == y.dtype and torch.equal(x, y)
                else:
                    assert np.all(np.asarray(x) == np.asarray(y))


def test_prefetch_loader_surfaces_worker_errors(tree):
    g, d, pre, train, t2s = tree
    ds = D.Dataset("train.txt", types.SimpleNamespace(model="naive"), pre, {"multi_speaker": False}, train,
                   text_to_sequence=lambda text, cleaners: (_ for _ in ()).throw(KeyError("no such symbol")))
    with pytest.raises(KeyError):
        list(D.PrefetchLoader(ds, D.RankShardSampler(len(ds), 4, shuffle=False), "cpu"))
