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


def test_re_ranking_native_and_oracle_match_reference(golden_dir):
    """k-reciprocal re-ranking (utils/rerank.py): the oracle restatement and the native threaded implementation against the
    reference's output on three small cases (default k1=20/k2=6, small k, and k2=1 = no query expansion)."""
    from oracle import metrics as OM
    from bpbreid_amd.metrics import re_ranking
    z = np.load(os.path.join(golden_dir, 'rerank.npz'))
    for tag in ('a', 'b', 'c'):
        k1, k2, lam = z[tag + '/params']
        qg, qq, gg, ref = z[tag + '/qg'], z[tag + '/qq'], z[tag + '/gg'], z[tag + '/out']
        got_o = OM.re_ranking(qg, qq, gg, int(k1), int(k2), float(lam))
        assert np.allclose(got_o, ref, atol=2e-6), (tag, np.abs(got_o - ref).max())
        for nth in (1, 4):
            got = re_ranking(qg, qq, gg, int(k1), int(k2), float(lam), nthreads=nth)
            assert got.dtype == np.float32 and got.shape == ref.shape
            assert np.allclose(got, ref, atol=2e-6), (tag, nth, np.abs(got - ref).max())
            assert np.array_equal(np.argsort(got, axis=1, kind='stable')[:, :5], np.argsort(ref, axis=1, kind='stable')[:, :5])
    with pytest.raises(ValueError):
        re_ranking(np.zeros((3, 4)), np.zeros((2, 2)), np.zeros((4, 4)))
    from bpbreid_amd import native as nv
    with pytest.raises(nv.NativeError):
        re_ranking(np.zeros((2, 3), np.float32), np.zeros((2, 2), np.float32), np.zeros((3, 3), np.float32), k1=20)   # k1 + 1 > Q + G


def test_re_ranking_medium_size_against_oracle():
    from oracle import metrics as OM
    from bpbreid_amd.metrics import re_ranking
    g = torch.Generator().manual_seed(11)
    nq, ng, dim = 40, 300, 32
    cent = torch.randn(25, dim, generator=g)
    qf = torch.nn.functional.normalize(cent[torch.randint(0, 25, (nq,), generator=g)] + 0.4 * torch.randn(nq, dim, generator=g), dim=1)
    gf = torch.nn.functional.normalize(cent[torch.randint(0, 25, (ng,), generator=g)] + 0.4 * torch.randn(ng, dim, generator=g), dim=1)
    d = lambda a, b: torch.cdist(a, b).numpy().astype(np.float32)
    qg, qq, gg = d(qf, gf), d(qf, qf), d(gf, gf)
    ref = OM.re_ranking(qg, qq, gg)
    got = re_ranking(qg, qq, gg)
    assert np.allclose(got, ref, atol=3e-6), np.abs(got - ref).max()
