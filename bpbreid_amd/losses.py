"""GiLt objective with the reference's loss-plug-in surface, computed by the HIP kernels of csrc/losses.hip.

Same class names, constructor and call signatures as the reference so an engine can use them unchanged:
  GiLtLoss(losses_weights, use_visibility_scores, triplet_margin, loss_name, use_gpu, writer)   GiLt_loss.py:11
      (embeddings_dict, visibility_scores_dict, id_cls_scores_dict, pids) -> (loss, summary)       GiLt_loss.py:45
  init_part_based_triplet_loss(name, margin=..., writer=...)                                     losses/__init__.py:24
      loss(part_based_embeddings[N,K,D], labels[N], parts_visibility=[N,K]|None) -> (loss, trivial, valid)
  CrossEntropyLoss(eps, label_smooth)(inputs, targets, weights=None)                              cross_entropy_loss.py:6
  BodyPartAttentionLoss(loss_type, label_smoothing, use_gpu)(pixels_cls_scores, targets)           body_part_attention_loss.py:11
Differences by design: no host synchronisation (no boolean indexing, no .item()), the writer is optional and
only absorbs calls, and scalar results stay on the device.
"""
import ctypes as C
from collections import OrderedDict

import torch
import torch.nn as nn

from . import native as nv

GLOBAL, FOREGROUND, CONCAT_PARTS, PARTS, PIXELS = 'globl', 'foreg', 'conct', 'parts', 'pixls'

_STRATEGY = {'part_averaged_triplet_loss': 0, 'part_max_triplet_loss': 1, 'part_min_triplet_loss': 2,
             'part_max_min_triplet_loss': 3, 'intra_parts_triplet_loss': 4, 'part_random_max_min_triplet_loss': 3}


def _need_cuda(t, what):
    nv.same_device(t, what)


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, weights, target_div, acc_on_selected, eps):
        _need_cuda(logits, 'CrossEntropyLoss')
        logits = logits.contiguous()
        r, c = logits.shape
        dev = logits.device
        row = torch.empty(2, r, device=dev, dtype=torch.float32)
        dl = torch.empty_like(logits)
        out = torch.empty(2, device=dev, dtype=torch.float32)
        w = weights.to(torch.float32).contiguous() if weights is not None else None
        nv.call('bpb_ce_label_smooth', logits.data_ptr(), c, targets.data_ptr(), target_div, nv.ptr(w), acc_on_selected, r, c,
                eps, row[0].data_ptr(), row[1].data_ptr(), dl.data_ptr(), c, out.data_ptr(), nv.stream())
        # continuous row weights that require a gradient (visibility scores, bpbreid.py:186-189): keep what d loss / d w needs
        ctx.wgrad = weights is not None and weights.dtype is not torch.bool and ctx.needs_input_grad[2]
        ctx.save_for_backward(dl, row if ctx.wgrad else None, w if ctx.wgrad else None)
        ctx.wshape = weights.shape if ctx.wgrad else None
        ctx.mark_non_differentiable(out)
        loss = out[0].clone()
        return loss, out

    @staticmethod
    def backward(ctx, gloss, _gout):
        dl, row, w = ctx.saved_tensors
        g = torch.empty_like(dl)
        gl = gloss.reshape(1).to(torch.float32).contiguous()
        nv.call('bpb_scale', dl.data_ptr(), gl.data_ptr(), 1.0, g.data_ptr(), dl.numel(), 0, nv.stream())
        gw = None
        if ctx.wgrad:
            gw = torch.empty_like(w)
            nv.call('bpb_ce_weight_grad', row[0].data_ptr(), w.data_ptr(), gl.data_ptr(), w.numel(), gw.data_ptr(), nv.stream())
            gw = gw.view(ctx.wshape)
        return g, None, gw, None, None, None


class CrossEntropyLoss(nn.Module):
    """Label-smoothed CE (cross_entropy_loss.py:6-56).  weights: None | float [N] | bool [N] (row filter)."""

    def __init__(self, eps=0.1, label_smooth=True):
        super().__init__()
        self.eps = eps if label_smooth else 0.0

    def forward(self, inputs, targets, weights=None, target_div=1, return_accuracy=False):
        sel = 1 if (weights is not None and weights.dtype is torch.bool) else 0
        loss, out = _CEFn.apply(inputs, targets.to(torch.int64).contiguous(), weights, target_div, sel, float(self.eps))
        if return_accuracy:
            return loss, out[1]
        return loss


class _PixelCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, masks, eps):
        _need_cuda(scores, 'BodyPartAttentionLoss')
        scores = scores.contiguous()
        n, k1, h, w = scores.shape
        dev = scores.device
        if masks.is_floating_point():          # float masks [N,K+1,Hm,Wm]: resize + arg-max inside the kernel
            masks = masks.to(device=dev, dtype=torch.float32).contiguous()
            if masks.dim() != 4 or masks.shape[0] != n or masks.shape[1] != k1:
                raise ValueError('BodyPartAttentionLoss: float target masks must be [N, K+1, Hm, Wm], got %s' % (tuple(masks.shape),))
            hm, wm = masks.shape[2:]
            mptr, tptr = masks.data_ptr(), None
        else:                                  # the reference engine's form: integer part index per pixel [N,Hf,Wf]
            masks = masks.to(device=dev, dtype=torch.int64).contiguous()
            if tuple(masks.shape) != (n, h, w):
                raise ValueError('BodyPartAttentionLoss: integer targets must be [N, Hf, Wf] = %s, got %s'
                                 % ((n, h, w), tuple(masks.shape)))
            hm, wm = h, w
            mptr, tptr = None, masks.data_ptr()
        ds = torch.empty_like(scores)
        nblocks = max(1, min(1024, n * h * w // 256))
        partial = torch.empty(nblocks * 2, device=dev, dtype=torch.float64)
        out = torch.empty(2, device=dev, dtype=torch.float32)
        nv.call('bpb_pixel_ce', scores.data_ptr(), mptr, tptr, n, k1, h, w, hm, wm, eps, ds.data_ptr(), partial.data_ptr(),
                nblocks, out.data_ptr(), nv.stream())
        ctx.save_for_backward(ds)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, gloss, _g):
        (ds,) = ctx.saved_tensors
        g = torch.empty_like(ds)
        gl = gloss.reshape(1).to(torch.float32).contiguous()
        nv.call('bpb_scale', ds.data_ptr(), gl.data_ptr(), 1.0, g.data_ptr(), ds.numel(), 0, nv.stream())
        return g, None, None


class BodyPartAttentionLoss(nn.Module):
    """Pixel-wise part CE (body_part_attention_loss.py:11-52).  Two call forms:
        loss(pixels_cls_scores[N,K+1,Hf,Wf], targets=int64 [N,Hf,Wf])           -- exactly what the reference engine passes
            after its own interpolate + argmax (part_based_engine.py:118-126);
        loss(pixels_cls_scores[N,K+1,Hf,Wf], target_masks=float [N,K+1,Hm,Wm])   -- the masks the engine holds; the bilinear
            resize + argmax of part_based_engine.py:118-124 then happens inside the kernel (one launch, no temporaries).
    returns (loss, {'pixls': {'c': loss, 'a': accuracy}}) with device scalars (the reference calls .item())."""

    def __init__(self, loss_type='cl', label_smoothing=0.1, use_gpu=True):
        super().__init__()
        if loss_type != 'cl':
            raise ValueError('Loss {} for part prediction is not supported'.format(loss_type))
        self.label_smoothing = label_smoothing

    def forward(self, pixels_cls_scores, targets):
        loss, out = _PixelCEFn.apply(pixels_cls_scores, targets, float(self.label_smoothing))
        summary = {PIXELS: OrderedDict(c=loss, a=out[1])}
        return loss, summary


class _TripletFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, labels, vis, drop, strategy, margin, epsilon):
        _need_cuda(emb, 'part triplet loss')
        n, k, d = emb.shape
        if emb.stride(2) != 1:
            emb = emb.contiguous()
        dev = emb.device
        vis_is_bool = 0
        visf = None
        if vis is not None:
            vis_is_bool = 1 if vis.dtype is torch.bool else 0
            visf = vis.to(torch.float32).contiguous()
        dist = torch.empty(k, n, n, device=dev, dtype=torch.float32)
        pair = torch.empty(k, n, n, device=dev, dtype=torch.float32)
        pair_part = torch.empty(n * n + 4 * k * n, device=dev, dtype=torch.int32)
        gsq = torch.empty(k, n, n, device=dev, dtype=torch.float32)
        out = torch.empty(4, device=dev, dtype=torch.float32)
        # continuous visibility scores that require a gradient: the mining kernel also returns d loss / d vis (part-averaged only)
        vgrad = vis is not None and not vis_is_bool and ctx.needs_input_grad[2] and strategy == 0
        gvis = torch.zeros(n, k, device=dev, dtype=torch.float32) if vgrad else None
        nv.call('bpb_part_triplet', emb.data_ptr(), emb.stride(0), emb.stride(1), labels.data_ptr(), nv.ptr(visf), vis_is_bool,
                nv.ptr(drop), n, k, d, strategy, margin, epsilon, dist.data_ptr(), pair.data_ptr(), pair_part.data_ptr(),
                gsq.data_ptr(), out.data_ptr(), nv.ptr(gvis), nv.stream())
        ctx.save_for_backward(emb, gsq, gvis)
        ctx.vshape = vis.shape if vgrad else None
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, gloss, _g):
        emb, gsq, gvis = ctx.saved_tensors
        n, k, d = emb.shape
        g = torch.empty(n, k, d, device=emb.device, dtype=torch.float32)
        gl = gloss.reshape(1).to(torch.float32).contiguous()
        nv.call('bpb_part_triplet_bwd', emb.data_ptr(), emb.stride(0), emb.stride(1), gsq.data_ptr(), gl.data_ptr(), 1.0, n, k, d,
                g.data_ptr(), k * d, d, 0, nv.stream())
        gv = None
        if gvis is not None:
            gv = torch.empty_like(gvis)
            nv.call('bpb_scale', gvis.data_ptr(), gl.data_ptr(), 1.0, gv.data_ptr(), gvis.numel(), 0, nv.stream())
            gv = gv.view(ctx.vshape)
        return g, None, gv, None, None, None, None


class PartAveragedTripletLoss(nn.Module):
    """Part-based batch-hard triplet loss (part_averaged_triplet_loss.py:10-224).  `name` selects how the K
    part-to-part distance matrices are combined (the reference's sub-classes)."""

    name = 'part_averaged_triplet_loss'

    def __init__(self, margin=0.3, epsilon=1e-16, writer=None, **kwargs):
        super().__init__()
        self.margin, self.epsilon, self.writer = margin, epsilon, writer

    def forward(self, part_based_embeddings, labels, parts_visibility=None):
        if parts_visibility is not None and parts_visibility.dtype is not torch.bool and self.name != 'part_averaged_triplet_loss':
            raise TypeError('continuous visibility scores are only supported by part_averaged_triplet_loss '
                            '(the reference applies ~ to a float mask and fails the same way)')
        drop = None
        if self.name == 'part_random_max_min_triplet_loss':        # part_random_max_min_triplet_loss.py:19
            n, k, _ = part_based_embeddings.shape
            drop = self._dropout_mask(k, n, part_based_embeddings.device).to(torch.uint8).contiguous()
        loss, out = _TripletFn.apply(part_based_embeddings, labels.to(torch.int64).contiguous(), parts_visibility, drop,
                                     _STRATEGY[self.name], float(self.margin), float(self.epsilon))
        # out = [loss, trivial ratio, valid ratio, #valid triplets]; no host sync: "no valid triplet" shows as out[3] == 0
        return loss, out[1], out[2]


    @staticmethod
    def _dropout_mask(k, n, device):
        """Keep-mask of part_random_max_min_triplet_loss.py:19 (`torch.rand(size=[K,N,N]) > 0.5`, drawn on the labels' device).
        Overridable so that a test can feed the kernel the very mask the reference drew."""
        return torch.rand(k, n, n, device=device) > 0.5


def _variant(n):
    return type(''.join(w.capitalize() for w in n.split('_')), (PartAveragedTripletLoss,), {'name': n})


__body_parts_losses = {n: (PartAveragedTripletLoss if n == 'part_averaged_triplet_loss' else _variant(n)) for n in _STRATEGY}


def init_part_based_triplet_loss(name, **kwargs):
    """losses/__init__.py:24-32.  'inter_parts_triplet_loss' is broken upstream and not offered."""
    if name not in __body_parts_losses:
        raise ValueError('Invalid loss name. Received "{}", but expected to be one of {}'.format(
            name, list(__body_parts_losses.keys())))
    return __body_parts_losses[name](**kwargs)


class _WeightedSumFn(torch.autograd.Function):
    """loss = sum_i w_i * term_i over device scalars in ONE launch (the reference: `loss += weight * term` per term,
    GiLt_loss.py:45-76 and part_based_engine.py:126); backward: one launch for the n scalars gloss * w_i."""

    @staticmethod
    def forward(ctx, weights, *terms):
        _need_cuda(terms[0], 'GiLtLoss')
        ts = [t.detach().reshape(1).to(torch.float32) for t in terms]
        out = torch.empty(1, device=ts[0].device, dtype=torch.float32)
        ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        ws = (C.c_float * len(ts))(*[float(w) for w in weights])
        nv.call('bpb_weighted_sum', ptrs, ws, len(ts), out.data_ptr(), nv.stream())
        ctx.weights, ctx.shapes = ws, [t.shape for t in terms]
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        n = len(ctx.shapes)
        g = torch.empty(n, device=gloss.device, dtype=torch.float32)
        gl = gloss.reshape(1).to(torch.float32).contiguous()
        nv.call('bpb_scalar_fanout', gl.data_ptr(), ctx.weights, n, g.data_ptr(), nv.stream())
        return (None,) + tuple(g[i].reshape(sh) for i, sh in enumerate(ctx.shapes))


def weighted_sum(weights, terms):
    """sum_i weights[i] * terms[i] (device scalars), chunks of 8 per launch."""
    terms, weights = list(terms), list(weights)
    while len(terms) > 8:
        terms = [_WeightedSumFn.apply(tuple(weights[:8]), *terms[:8])] + terms[8:]
        weights = [1.0] + weights[8:]
    return _WeightedSumFn.apply(tuple(weights), *terms)


class GiLtLoss(nn.Module):
    """Global-identity Local-triplet loss (GiLt_loss.py:11-119)."""

    default_losses_weights = {GLOBAL: {'id': 1., 'tr': 0.}, FOREGROUND: {'id': 1., 'tr': 0.},
                              CONCAT_PARTS: {'id': 1., 'tr': 0.}, PARTS: {'id': 0., 'tr': 1.}}

    def __init__(self, losses_weights=None, use_visibility_scores=False, triplet_margin=0.3,
                 loss_name='part_averaged_triplet_loss', use_gpu=True, writer=None):
        super().__init__()
        self.losses_weights = losses_weights if losses_weights is not None else self.default_losses_weights
        self.part_triplet_loss = init_part_based_triplet_loss(loss_name, margin=triplet_margin, writer=writer)
        self.identity_loss = CrossEntropyLoss(label_smooth=True)
        self.use_visibility_scores = use_visibility_scores

    def forward(self, embeddings_dict, visibility_scores_dict, id_cls_scores_dict, pids):
        weights, terms, loss_summary = self.weighted_terms(embeddings_dict, visibility_scores_dict, id_cls_scores_dict, pids)
        if not terms:
            return torch.zeros((), device=pids.device), loss_summary
        return weighted_sum(weights, terms), loss_summary

    def weighted_terms(self, embeddings_dict, visibility_scores_dict, id_cls_scores_dict, pids):
        """-> (weights, loss terms, summary): the weighted sum itself is one launch (weighted_sum); the engine appends its
        body-part-attention term to the same sum (part_based_engine.py:126)."""
        loss_summary, terms, weights = {}, [], []
        keys = [GLOBAL, FOREGROUND, CONCAT_PARTS, PARTS]
        for key in keys:
            info = OrderedDict()
            w = self.losses_weights[key]['id']
            if w > 0:
                c, a = self.compute_id_cls_loss(id_cls_scores_dict[key], visibility_scores_dict[key], pids)
                terms.append(c)
                weights.append(w)
                info['c'], info['a'] = c, a
            loss_summary[key] = info
        for key in keys:
            w = self.losses_weights[key]['tr']
            if w > 0:
                t, tt, vt = self.compute_triplet_loss(embeddings_dict[key], visibility_scores_dict[key], pids)
                terms.append(t)
                weights.append(w)
                loss_summary[key].update(t=t, tt=tt, vt=vt)
        return weights, terms, loss_summary

    def compute_triplet_loss(self, embeddings, visibility_scores, pids):
        vis = None
        if self.use_visibility_scores:
            vis = visibility_scores if visibility_scores.dim() == 2 else visibility_scores.unsqueeze(1)
        emb = embeddings if embeddings.dim() == 3 else embeddings.unsqueeze(1)
        return self.part_triplet_loss(emb, pids, parts_visibility=vis)

    def compute_id_cls_loss(self, id_cls_scores, visibility_scores, pids):
        div = 1
        if id_cls_scores.dim() == 3:                       # [N, K, classes]: row (n, k) is labelled pids[n]
            div = id_cls_scores.shape[1]
            id_cls_scores = id_cls_scores.flatten(0, 1)
            visibility_scores = visibility_scores.flatten(0, 1)
        weights = visibility_scores if self.use_visibility_scores else None   # bool -> row filter, float -> weights
        return self.identity_loss(id_cls_scores, pids, weights, target_div=div, return_accuracy=True)
