"""Input side, CPU part: the oracle's mask-preprocessing restatement against the reference's golden vectors,
and the P x K sampler (host logic of the product) against the reference's index sequences for fixed seeds."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import data as OD
from bpbreid_amd.data import RandomIdentitySampler, MaskPreprocessor


def _groups(z, key):
    if 'goff/' + key not in z.files:
        return None, 'max'
    off, ch = z['goff/' + key], z['gch/' + key]
    return [list(map(int, ch[off[i]:off[i + 1]])) for i in range(len(off) - 1)], ('sum' if int(z['sum/' + key]) else 'max')


def test_oracle_mask_preprocessing_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, 'masks_pre.npz'))
    raw = torch.from_numpy(z['raw'])
    assert len(z['cases']) >= 4
    for key in z['cases']:
        name, strat, sw, thr, scale = str(key).split('|')
        groups, mode = _groups(z, key)
        got = OD.preprocess_masks(raw, raw.shape[2], raw.shape[3], int(scale), groups, mode, strat, float(sw), float(thr))
        ref = z['out/' + key]
        assert got.shape == ref.shape, key
        assert np.allclose(got.numpy(), ref, atol=1e-7, equal_nan=True), key


def test_sampler_reproduces_reference_sequences(golden_dir):
    z = np.load(os.path.join(golden_dir, 'sampler.npz'))
    source = [{'pid': int(p)} for p in z['pids']]
    for bs, ni in ((16, 4), (8, 2), (12, 3)):
        random.seed(1234)
        np.random.seed(4321)
        s = RandomIdentitySampler(source, bs, ni)
        seq = np.array(list(iter(s)), dtype=np.int64)
        assert len(s) == int(z['len/%d/%d' % (bs, ni)])
        assert np.array_equal(seq, z['seq/%d/%d' % (bs, ni)]), (bs, ni)
        pids = z['pids'][seq].reshape(-1, bs // ni, ni) if len(seq) % bs == 0 else None
        if pids is not None:                      # every batch: P identities x K instances
            assert (pids == pids[:, :, :1]).all()
            assert all(len(set(b[:, 0])) == bs // ni for b in pids)
    with pytest.raises(ValueError):
        RandomIdentitySampler(source, 2, 4)


def test_mask_preprocessor_argument_checks():
    with pytest.raises(ValueError):
        MaskPreprocessor(256, 128, background_computation_strategy='nope')
    mp = MaskPreprocessor(256, 128, 4, parts_grouping={'a': ['x', 'y'], 'b': ['z']}, parts_map={'x': 0, 'y': 2, 'z': 1})
    assert mp.parts_num == 2 and mp.size == (64, 32) and mp.groups == [[0, 2], [1]]
    from bpbreid_amd import native as nv
    with pytest.raises(nv.NativeError):
        mp(torch.zeros(1, 3, 8, 8))               # no CPU fallback
