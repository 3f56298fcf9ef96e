"""PrefetchLoader on the GPU (pinned staging + copies on a side stream): same tensors as the synchronous
to_device path."""
import types

import numpy as np
import pytest
import torch

from test_data_path import tree  # noqa: F401  (fixture: rebuilds the synthetic preprocessed tree from dataset.npz)
from mixgan_tts_amd import data as D

pytestmark = pytest.mark.gpu


def test_prefetch_loader_device_batches(tree):  # noqa: F811
    g, d, pre, train, t2s = tree
    ds = D.Dataset("train.txt", types.SimpleNamespace(model="naive"), pre, {"multi_speaker": False}, train,
                   sort=True, drop_last=False, text_to_sequence=t2s)
    smp = D.RankShardSampler(len(ds), 5, rank=0, world=1, seed=3)
    want = [[D.to_device(b, "cpu") for b in ds.collate_fn([ds[i] for i in idxs])] for idxs in smp]
    got = list(D.PrefetchLoader(ds, smp, "cuda:0", depth=2, workers=2))
    assert len(got) == len(want) == 2
    n = 0
    for gb, wb in zip(got, want):
        assert len(gb) == len(wb)
        for a, b in zip(gb, wb):
            for x, y in zip(a, b):
                if torch.is_tensor(y):
                    assert x.is_cuda and x.dtype == y.dtype and torch.equal(x.cpu(), y)
                    n += 1
                else:
                    assert np.all(np.asarray(x) == np.asarray(y))
    assert n > 20
