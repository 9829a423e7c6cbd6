"""Pure-torch CPU restatement of the GiLt objective (TEST INFRASTRUCTURE).

Restates torchreid/losses/cross_entropy_loss.py:34-56, part_averaged_triplet_loss.py:35-224,
part_{max,min,max_min,individual}_triplet_loss.py, utils/tensortools.py:3-21,
GiLt_loss.py:45-119, body_part_attention_loss.py:45-52 and the BPA part of
engine/image/part_based_engine.py:114-128.  `part_random_max_min` draws torch.rand and is
therefore restated but only pinned with a fixed RNG state.
"""
import torch
import torch.nn.functional as F

GLOBAL, FOREGROUND, CONCAT_PARTS, PARTS, PIXELS = 'globl', 'foreg', 'conct', 'parts', 'pixls'
FMAX = torch.finfo(torch.float32).max


def label_smooth_ce(logits, targets, weights=None, eps=0.1):
    """cross_entropy_loss.py:34-56."""
    logp = F.log_softmax(logits, dim=1)
    c = logits.shape[1]
    t = torch.zeros_like(logp).scatter_(1, targets.unsqueeze(1), 1.0)
    t = (1 - eps) * t + eps / c
    if weights is not None:
        per = (-t * logp).sum(dim=1)
        return (per * F.normalize(weights.to(per.dtype), p=1, dim=0)).sum()
    return (-t * logp).mean(0).sum()


def masked_mean(x, mask):
    """tensortools.py:12-21: mean over dim 0 weighted by mask; -1 where no valid entry."""
    w = mask.sum(0)
    out = (x * mask).sum(0) / (w + (w == 0))
    invalid = (mask.sum(0) == 0)
    return out * (~invalid) + invalid * (-1.0)


def part_pairwise_dist(emb_knd, eps=1e-16):
    """part_averaged_triplet_loss.py:77-93, emb [K,N,D] -> [K,N,N] (non-squared)."""
    dot = emb_knd @ emb_knd.transpose(1, 2)
    sq = dot.diagonal(dim1=1, dim2=2)
    d = F.relu(sq.unsqueeze(2) - 2 * dot + sq.unsqueeze(1))
    zero = (d == 0).to(d.dtype)
    return torch.sqrt(d + zero * eps) * (1 - zero)


def combine_part_dists(strategy, dist, mask, labels):
    """_combine_part_based_dist_matrices of the six variants.  dist [K,N,N], mask None|bool|float."""
    same = labels.unsqueeze(0) == labels.unsqueeze(1)
    if strategy == 'part_averaged_triplet_loss':
        return dist.mean(0) if mask is None else masked_mean(dist, mask)
    if strategy == 'intra_parts_triplet_loss':                 # part_individual_triplet_loss.py:22-32
        return dist if mask is None else dist * mask + (~mask) * (-1.0)
    if strategy == 'part_random_max_min_triplet_loss':         # part_random_max_min_triplet_loss.py:15-44
        if mask is None:
            mask = torch.ones_like(dist, dtype=torch.bool)
        mask = mask * (torch.rand(mask.shape) > 0.5)
    for_max = dist if mask is None else dist * mask + (~mask) * (-1.0)
    for_min = dist if mask is None else dist * mask + (~mask) * FMAX
    if strategy == 'part_max_triplet_loss':                    # part_max_triplet_loss.py:12-26
        return for_max.max(0)[0]
    mn = for_min.min(0)[0]
    if strategy == 'part_min_triplet_loss':                    # part_min_triplet_loss.py:14-32
        out = mn
    elif strategy in ('part_max_min_triplet_loss', 'part_random_max_min_triplet_loss'):
        out = for_max.max(0)[0] * same + mn * ~same             # part_max_min_triplet_loss.py:34-36
    else:
        raise ValueError(strategy)
    if mask is not None:
        inv = mask.sum(0) == 0
        out = out * (~inv) + inv * (-1.0)
    return out


def batch_hard(dist, labels, margin):
    """part_averaged_triplet_loss.py:95-224.  dist [N,N] or [K,N,N] with -1 = invalid pair.
    Returns (loss, trivial_ratio, valid_ratio) or None when no valid triplet exists (:161-163)."""
    if dist.dim() == 2:
        dist = dist.unsqueeze(0)
    n = labels.shape[0]
    valid = dist != -1.0
    same = labels.unsqueeze(0) == labels.unsqueeze(1)
    pos = (same & ~torch.eye(n, dtype=torch.bool)).unsqueeze(0) & valid
    neg = (~same).unsqueeze(0) & valid
    dp = (dist * pos - (~pos).to(dist.dtype)).max(-1)[0]
    dn = (dist * neg + (~neg).to(dist.dtype) * FMAX).min(-1)[0]
    ok = (dp != -1) & (dn != FMAX)
    if not ok.any():
        return None
    p, q = dp[ok], dn[ok]
    hinge = F.relu(p - q + (margin if margin > 0 else 0.3))
    trivial = (hinge == 0).sum() / hinge.numel()
    valid_ratio = ok.sum() / ok.numel()
    if margin > 0:
        return hinge.mean(), trivial, valid_ratio
    soft = F.soft_margin_loss(q - p, torch.ones_like(p))      # :197-211
    if soft == float('inf'):
        return hinge.mean(), trivial, valid_ratio
    return soft, trivial, valid_ratio


def part_triplet(strategy, emb_nkd, labels, vis_nk=None, margin=0.3):
    """PartAveragedTripletLoss.forward (:35-65) for any registered combination strategy."""
    dist = part_pairwise_dist(emb_nkd.transpose(0, 1))
    mask = None
    if vis_nk is not None:
        v = vis_nk.t()
        mask = v.unsqueeze(1) * v.unsqueeze(2)
        if mask.dtype is not torch.bool:
            mask = torch.sqrt(mask)
    return batch_hard(combine_part_dists(strategy, dist, mask, labels), labels, margin)


DEFAULT_WEIGHTS = {GLOBAL: {'id': 1., 'tr': 0.}, FOREGROUND: {'id': 1., 'tr': 0.},
                   CONCAT_PARTS: {'id': 1., 'tr': 0.}, PARTS: {'id': 0., 'tr': 1.}}


def gilt(emb, vis, scores, pids, weights=DEFAULT_WEIGHTS, use_visibility=False, margin=0.3,
         strategy='part_averaged_triplet_loss'):
    """GiLt_loss.py:45-119 -> (loss, summary)."""
    terms, summary = [], {}
    keys = [GLOBAL, FOREGROUND, CONCAT_PARTS, PARTS]
    for k in keys:
        info = {}
        if weights[k]['id'] > 0:
            s, v, y = scores[k], vis[k], pids
            if s.dim() == 3:
                m = s.shape[1]
                s, y, v = s.flatten(0, 1), pids.unsqueeze(1).expand(-1, m).flatten(0, 1), v.flatten(0, 1)
            w = None
            if use_visibility and v.dtype is torch.bool:
                s, y = s[v], y[v]
            elif use_visibility:
                w = v
            ce = label_smooth_ce(s, y, w)
            terms.append(weights[k]['id'] * ce)
            info['c'] = ce
            info['a'] = (s.argmax(1) == y).float().mean()
        summary[k] = info
    for k in keys:
        if weights[k]['tr'] > 0:
            e = emb[k] if emb[k].dim() == 3 else emb[k].unsqueeze(1)
            v = None
            if use_visibility:
                v = vis[k] if vis[k].dim() == 2 else vis[k].unsqueeze(1)
            t, tt, vt = part_triplet(strategy, e, pids, v, margin)
            terms.append(weights[k]['tr'] * t)
            summary[k].update({'t': t, 'tt': tt, 'vt': vt})
    if not terms:
        return torch.tensor(0.), summary
    return torch.stack(terms).sum(), summary


def body_part_attention(pix_scores, target_masks, label_smoothing=0.1):
    """part_based_engine.py:114-128 + body_part_attention_loss.py:45-52 -> (loss, accuracy)."""
    tm = F.interpolate(target_masks, pix_scores.shape[2:], mode='bilinear', align_corners=True)
    tgt = tm.argmax(dim=1).flatten()
    s = pix_scores.permute(0, 2, 3, 1).flatten(0, 2)
    return F.cross_entropy(s, tgt, label_smoothing=label_smoothing), (s.argmax(1) == tgt).float().mean()


def combined_loss(model_out, pids, target_masks, weights=DEFAULT_WEIGHTS, pixel_weight=0.35,
                  use_visibility=False, margin=0.3, strategy='part_averaged_triplet_loss'):
    """ImagePartBasedEngine.combine_losses (part_based_engine.py:107-130)."""
    emb, vis, scores, pix, _, _ = model_out
    loss, summary = gilt(emb, vis, scores, pids, weights, use_visibility, margin, strategy)
    if pix is not None and target_masks is not None and pixel_weight > 0:
        bpa, acc = body_part_attention(pix, target_masks)
        loss = loss + pixel_weight * bpa
        summary[PIXELS] = {'c': bpa, 'a': acc}
    return loss, summary
