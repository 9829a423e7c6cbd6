"""Diagnostic (GPU box): head backward intermediates vs an fp64 torch re-evaluation with exposed intermediates."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import torch
import torch.nn.functional as F
import common as Cm
from bpbreid_amd.model import bpbreid
from bpbreid_amd.engine import ImagePartBasedEngine
from oracle.bpbreid import BPBreID as OracleModel, _masked_pool
from oracle import losses as OL

dev = torch.device('cuda', 0)
W = {'globl': {'id': 1., 'tr': 0.}, 'foreg': {'id': 1., 'tr': 1.}, 'conct': {'id': 1., 'tr': 0.}, 'parts': {'id': 0., 'tr': 1.},
     'pixls': {'ce': 0.35}}


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


for backbone, k, d, n, h, w, ncls in (('hrnet_w8', 5, 64, 8, 64, 32, 16), ('hrnet_w8', 2, 64, 8, 64, 32, 16), ('hrnet_w8', 5, 512, 8, 64, 32, 16)):
    print('=====', backbone, 'K', k, 'D', d)
    cfg = Cm.make_cfg(backbone, k, d)
    model = Cm.fill_state_dict_(bpbreid(ncls, config=cfg, pretrained=False)).to(dev).train()
    om = Cm.fill_state_dict_(OracleModel(ncls, cfg)).double().train()
    imgs, masks, pids = Cm.synth_batch(n, h, w, k, ncls)
    eng = ImagePartBasedEngine(model, losses_weights=W, mask_filtering_training=True)
    out = model(imgs.to(dev), external_parts_masks=masks.to(dev))
    loss, _ = eng.combine_losses(out[1], out[0], out[2], pids.to(dev), out[3], masks.to(dev), bpa_weight=0.35)
    loss.backward()
    torch.cuda.synchronize()
    plan = next(iter(model._plans.values()))
    # ---- oracle with exposed intermediates (same math as oracle.bpbreid.BPBreID.forward)
    feats = om.backbone_appearance_feature_extractor(imgs.double())
    feats.retain_grad()
    pix = om.pixel_classifier(feats)
    pix.retain_grad()
    probs = F.softmax(pix, dim=1)
    probs.retain_grad()
    bg, parts = probs[:, 0], probs[:, 1:]
    fg = parts.max(dim=1)[0]
    onehot = F.one_hot(probs.argmax(dim=1), k + 1).permute(0, 3, 1, 2)
    vis = onehot.amax(dim=(2, 3)).to(torch.bool)
    g0 = feats.mean(dim=(2, 3))
    f0 = _masked_pool(feats, fg.unsqueeze(1), False).flatten(1, 2)
    b0 = _masked_pool(feats, bg.unsqueeze(1), False).flatten(1, 2)
    p0 = _masked_pool(feats, parts, True)
    for t in (g0, f0, b0, p0):
        t.retain_grad()
    g = om.global_after_pooling_dim_reduce(g0)
    f = om.foreground_after_pooling_dim_reduce(f0)
    b = om.background_after_pooling_dim_reduce(b0)
    p = om.parts_after_pooling_dim_reduce(p0)
    for t in (g, f, p):
        t.retain_grad()
    c = p.flatten(1, 2)
    bn_g, s_g = om.global_identity_classifier(g)
    bn_f, s_f = om.foreground_identity_classifier(f)
    bn_c, s_c = om.concat_parts_identity_classifier(c)
    emb = {'globl': g.float(), 'foreg': f.float(), 'conct': c.float(), 'parts': p.float()}
    fgv = vis.amax(1)
    visd = {'globl': torch.ones_like(fgv), 'foreg': fgv, 'conct': fgv, 'parts': vis[:, 1:]}
    ids = {'globl': s_g.float(), 'foreg': s_f.float(), 'conct': s_c.float(), 'parts': None}
    l1, _ = OL.gilt(emb, visd, ids, pids, W, use_visibility=True)
    bpa, _ = OL.body_part_attention(pix.float(), masks)
    rloss = l1 + 0.35 * bpa
    rloss.backward()
    print('loss', float(loss), float(rloss))
    gp_ref = torch.cat([g0.grad.unsqueeze(1), f0.grad.unsqueeze(1), b0.grad.unsqueeze(1) if b0.grad is not None else torch.zeros_like(g0).unsqueeze(1), p0.grad], 1)
    for j, nm in enumerate(['global', 'fg', 'bg'] + ['part%d' % i for i in range(k)]):
        print('  gpool row %-7s rel %.3e  (scale %.3e)' % (nm, rel(plan.g_pooled[:, j], gp_ref[:, j]) if gp_ref[:, j].abs().max() > 0 else -1, float(gp_ref[:, j].abs().max())))
    print('  pooled fwd         rel %.3e' % rel(plan.pooled, torch.cat([g0.unsqueeze(1), f0.unsqueeze(1), b0.unsqueeze(1), p0], 1)))
    x = feats.detach().permute(0, 2, 3, 1).reshape(n, -1, feats.shape[1])
    Dref = torch.einsum('njc,npc->npj', gp_ref[:, 1:], x)
    print('  D                  rel %.3e' % rel(plan.Dd, Dref))
    gpr = (gp_ref * torch.cat([g0.unsqueeze(1), f0.unsqueeze(1), b0.unsqueeze(1), p0], 1).detach()).sum(-1)
    print('  gp                 rel %.3e' % rel(plan.gp, gpr))
    print('  dprob->dlogit tot  rel %.3e' % rel(plan.dlogit.view_as(pix), pix.grad))
    e = (plan.dlogit.view_as(pix).double().cpu() - pix.grad).abs()
    idx = (e == e.max()).nonzero()[0].tolist()
    print('     worst at', idx, 'mine', float(plan.dlogit.view_as(pix)[tuple(idx)]), 'ref', float(pix.grad[tuple(idx)]),
          'argpart', int(plan.argpart.view(n, -1)[idx[0], idx[2] * feats.shape[3] + idx[3]]), 'probs', probs[idx[0], :, idx[2], idx[3]].tolist())
    print('  zinv', plan.zinv[idx[0]].tolist())
    print('  d spatial feats    rel %.3e' % rel(plan.feats.grad.permute(0, 3, 1, 2), feats.grad))
