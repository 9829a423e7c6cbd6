"""Golden vectors for the input side (mask preprocessing chain + P x K sampler) from the REAL reference.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_data.py          (development container only)
Writes tests/golden/masks_pre.npz and tests/golden/sampler.npz.
"""
import os
import random
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE]
import numpy as np                                             # noqa: E402
import torch                                                   # noqa: E402
from _ref_loader import load_reference                        # noqa: E402

load_reference()
from torchreid.data.masks_transforms import masks_preprocess_all, AddBackgroundMask, ResizeMasks   # noqa: E402
from torchreid.data.sampler import RandomIdentitySampler      # noqa: E402

out = {}
g = torch.Generator().manual_seed(99)
N, C, H, W = 3, 36, 64, 32
raw = torch.rand(N, C, H, W, generator=g) ** 3                 # confidence-like: mostly small, a few strong responses
raw[0, :, :8] = 0                                              # an all-zero region (background by every strategy)
out['raw'] = raw.numpy()
cases = [('five_v', 'threshold', 15.0, 0.5, 4), ('eight', 'sum', 0.0, 0.3, 4), ('six', 'diff_from_max', 10.0, 0.3, 2),
         ('none', 'threshold', 15.0, 0.5, 4), ('two_v', 'sum', 15.0, 0.5, 4)]
names = []
for name, strat, sw, thr, scale in cases:
    if name != 'none' and name not in masks_preprocess_all:
        continue
    t = masks_preprocess_all[name]() if name != 'none' else None
    res = []
    for m in raw:
        x = t.apply_to_mask(m) if t is not None else m
        x = AddBackgroundMask(strat, sw, thr).apply_to_mask(x)
        x = ResizeMasks(H, W, scale).apply_to_mask(x)
        res.append(x)
    key = '%s|%s|%g|%g|%d' % (name, strat, sw, thr, scale)
    names.append(key)
    out['out/' + key] = torch.stack(res).numpy()
    if t is not None:
        groups = [[t.parts_map[k] for k in t.parts_grouping[p]] for p in t.parts_names]
        out['goff/' + key] = np.cumsum([0] + [len(gr) for gr in groups]).astype(np.int32)
        out['gch/' + key] = np.concatenate([np.asarray(gr, dtype=np.int32) for gr in groups])
        out['sum/' + key] = np.int32(1 if t.combine_mode == 'sum' else 0)
out['cases'] = np.array(names)
np.savez_compressed(os.path.join(HERE, 'masks_pre.npz'), **out)
print('masks_pre.npz:', names)

# sampler: identities with 1..9 images (some below num_instances -> over-sampling path)
rng = np.random.RandomState(5)
pids = np.repeat(np.arange(23), rng.randint(1, 10, size=23))
rng.shuffle(pids)
source = [{'pid': int(p)} for p in pids]
seqs = {}
for bs, ni in ((16, 4), (8, 2), (12, 3)):
    random.seed(1234)
    np.random.seed(4321)
    s = RandomIdentitySampler(source, bs, ni)
    seqs['seq/%d/%d' % (bs, ni)] = np.array(list(iter(s)), dtype=np.int64)
    seqs['len/%d/%d' % (bs, ni)] = np.int64(len(s))
np.savez_compressed(os.path.join(HERE, 'sampler.npz'), pids=pids.astype(np.int64), **seqs)
print('sampler.npz:', {k: (v.shape if hasattr(v, 'shape') else v) for k, v in seqs.items()})

# k-reciprocal re-ranking (utils/rerank.py): small random feature sets, euclidean distances like engine.py:431-437
from torchreid.utils.rerank import re_ranking                  # noqa: E402
from torchreid import metrics as RM                            # noqa: E402
rr = {}
g = torch.Generator().manual_seed(77)
for tag, (nq, ng, dim, k1, k2, lam) in {'a': (12, 40, 16, 20, 6, 0.3), 'b': (7, 30, 8, 6, 3, 0.5), 'c': (5, 25, 8, 5, 1, 0.3)}.items():
    cent = torch.randn(8, dim, generator=g)
    qf = torch.nn.functional.normalize(cent[torch.randint(0, 8, (nq,), generator=g)] + 0.3 * torch.randn(nq, dim, generator=g), dim=1)
    gf = torch.nn.functional.normalize(cent[torch.randint(0, 8, (ng,), generator=g)] + 0.3 * torch.randn(ng, dim, generator=g), dim=1)
    qg = RM.compute_distance_matrix(qf, gf, 'euclidean').numpy()
    qq = RM.compute_distance_matrix(qf, qf, 'euclidean').numpy()
    gg = RM.compute_distance_matrix(gf, gf, 'euclidean').numpy()
    rr['%s/qg' % tag], rr['%s/qq' % tag], rr['%s/gg' % tag] = qg, qq, gg
    rr['%s/params' % tag] = np.array([k1, k2, lam], dtype=np.float64)
    rr['%s/out' % tag] = re_ranking(qg, qq, gg, k1=k1, k2=k2, lambda_value=lam)
np.savez_compressed(os.path.join(HERE, 'rerank.npz'), **rr)
print('rerank.npz:', {k: v.shape for k, v in rr.items() if k.endswith('/out')})
