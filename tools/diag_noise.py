"""GPU diagnostic: distribution of |gpu - ref64| against the reference's own |ref32 - ref64| for every output of a golden
model fixture (max, rms, 99.9th percentile).  Usage: python tools/diag_noise.py hr32_k5_full [train|eval]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tests')]
import common as Cm                                           # noqa: E402
from bpbreid_amd.model import bpbreid                        # noqa: E402
from test_gpu_model import MODEL_CASES                       # noqa: E402


def stats(tag, got, r32, r64):
    got, r32, r64 = [np.asarray(a, dtype=np.float64).ravel() for a in (got, r32, r64)]
    e, n = np.abs(got - r64), np.abs(r32 - r64)
    sc = np.abs(r64).max()
    print('%-22s scale %.3e | max err %.3e noise %.3e (x%.1f) | rms err %.3e noise %.3e (x%.1f) | p99.9 err %.3e noise %.3e'
          % (tag, sc, e.max(), n.max(), e.max() / max(n.max(), 1e-30), np.sqrt((e ** 2).mean()), np.sqrt((n ** 2).mean()),
             np.sqrt((e ** 2).mean()) / max(np.sqrt((n ** 2).mean()), 1e-30), np.quantile(e, 0.999), np.quantile(n, 0.999)))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'hr32_k5_full'
    modes = sys.argv[2:] or ['train', 'eval']
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'model_%s.npz' % name))
    backbone, extra = MODEL_CASES[name]
    k, d, n, h, w, ncls = [int(x) for x in z['meta']]
    dev = torch.device('cuda', 0)
    model = Cm.fill_state_dict_(bpbreid(ncls, config=Cm.make_cfg(backbone, k, d, **extra), pretrained=False)).to(dev)
    imgs, masks, pids = Cm.synth_batch(n, h, w, k, ncls)
    for mode in modes:
        model.train(mode == 'train')
        with torch.no_grad():
            emb, vis, ids, pix, sp, mk = model(imgs.to(dev), external_parts_masks=masks.to(dev))
        t32, t64 = 'f32/' + mode, 'f64/' + mode
        print('==', name, mode)
        stats('spatial', Cm.to_np(Cm.subsample(sp.contiguous())), z[t32 + '/sp_sub'], z[t64 + '/sp_sub'])
        stats('pix', Cm.to_np(pix), z[t32 + '/pix'], z[t64 + '/pix'])
        stats('masks', Cm.to_np(mk['parts']), z[t32 + '/mask_parts'], z[t64 + '/mask_parts'])
        for kk, v in emb.items():
            stats('emb ' + kk, Cm.to_np(v), z['%s/emb/%s' % (t32, kk)], z['%s/emb/%s' % (t64, kk)])
        for kk, v in ids.items():
            stats('ids ' + kk, Cm.to_np(v), z['%s/ids/%s' % (t32, kk)], z['%s/ids/%s' % (t64, kk)])


if __name__ == '__main__':
    main()
