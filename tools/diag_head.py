"""Diagnostic (GPU box): where does the model-level gradient deviate from the oracle?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import torch
import common as Cm
from bpbreid_amd.model import bpbreid
from bpbreid_amd.engine import ImagePartBasedEngine
from oracle.bpbreid import BPBreID as OracleModel
from oracle import losses as OL

dev = torch.device('cuda', 0)
W = {'globl': {'id': 1., 'tr': 0.}, 'foreg': {'id': 1., 'tr': 1.}, 'conct': {'id': 1., 'tr': 0.}, 'parts': {'id': 0., 'tr': 1.},
     'pixls': {'ce': 0.35}}


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


for backbone, k, d, n, h, w, ncls in (('hrnet_w8', 5, 64, 8, 64, 32, 16), ('resnet50', 2, 512, 8, 128, 64, 16)):
    print('=====', backbone)
    cfg = Cm.make_cfg(backbone, k, d)
    model = Cm.fill_state_dict_(bpbreid(ncls, config=cfg, pretrained=False)).to(dev).train()
    om = Cm.fill_state_dict_(OracleModel(ncls, cfg)).double().train()
    imgs, masks, pids = Cm.synth_batch(n, h, w, k, ncls)
    eng = ImagePartBasedEngine(model, losses_weights=W, mask_filtering_training=True)
    out = model(imgs.to(dev), external_parts_masks=masks.to(dev))
    loss, _ = eng.combine_losses(out[1], out[0], out[2], pids.to(dev), out[3], masks.to(dev), bpa_weight=0.35)
    for t in out[0].values():
        t.retain_grad()
    out[3].retain_grad()
    loss.backward()
    torch.cuda.synchronize()
    oo = om(imgs.double(), masks.double())
    f32 = lambda dct: {kk: (v.float() if v.is_floating_point() else v) for kk, v in dct.items()}
    ol = (f32(oo[0]), f32(oo[1]), f32(oo[2]), oo[3].float(), oo[4], oo[5])
    rloss, _ = OL.combined_loss(ol, pids, masks, W, 0.35, use_visibility=True)
    for t in oo[0].values():
        t.retain_grad()
    oo[3].retain_grad()
    oo[4].retain_grad()
    rloss.backward()
    print('loss', float(loss), float(rloss))
    plan = next(iter(model._plans.values()))
    for kk in oo[0]:
        if oo[0][kk].grad is not None and out[0][kk].grad is not None:
            print('  emb grad %-9s rel %.3e' % (kk, rel(out[0][kk].grad, oo[0][kk].grad)))
    print('  dlogit (total)     rel %.3e' % rel(plan.dlogit.view_as(oo[3]), oo[3].grad))
    print('  pix ext grad       rel %.3e  (vs total)' % rel(out[3].grad, oo[3].grad))
    dx = plan.feats.grad.permute(0, 3, 1, 2)
    print('  d spatial feats    rel %.3e' % rel(dx, oo[4].grad))
    c = dx.shape[1]
    for lo in range(0, c, max(1, c // 8)):
        hi = min(c, lo + max(1, c // 8))
        print('     channels %4d:%4d rel %.3e' % (lo, hi, rel(dx[:, lo:hi], oo[4].grad[:, lo:hi])))
    rp = dict(om.named_parameters())
    rows = []
    for name, p in model.named_parameters():
        if p.grad is not None and rp[name].grad is not None:
            rows.append((rel(p.grad, rp[name].grad), name))
    rows.sort(reverse=True)
    head = [r for r in rows if not r[1].startswith('backbone')]
    print('  worst head params:', [(round(e, 6), nme) for e, nme in head[:6]])
    print('  worst backbone params:', [(round(e, 6), nme) for e, nme in rows[:6]])
    bb = [r for r in rows if r[1].startswith('backbone')]
    print('  best backbone params:', [(round(e, 6), nme) for e, nme in bb[-4:]])
