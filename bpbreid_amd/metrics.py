"""Eval-time part-based distance and ranking with the reference's metric call signatures.

  compute_distance_matrix_using_bp_features(qf, gf, qf_parts_visibility, gf_parts_visibility, dist_combine_strat,
        batch_size_pairwise_dist_matrix, use_gpu, metric) -> (distmat[Q,G], body_parts_distmat[P,Q,G])   distance.py:87
  evaluate_rank(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50, eval_metric='default') -> {'cmc','mAP'}   rank.py:173
The distance runs as one MFMA kernel per gallery shard (csrc/distance.hip); the ranking is the native evaluator of
csrc/rank.cpp (the wired-up counterpart of the reference's dead Cython module).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import native as nv


def _vis_mode(qv, gv):
    if qv is None or gv is None:
        return 0
    return 1 if (qv.dtype is torch.bool and gv.dtype is torch.bool) else 2


def _check_args(dist_combine_strat, metric):
    if dist_combine_strat not in ('mean', 'max'):
        raise ValueError('Body parts distance combination strategy "{}" not supported'.format(dist_combine_strat))
    if metric not in ('euclidean', 'cosine'):
        raise ValueError('Unknown distance metric: {}. Please choose either "euclidean" or "cosine"'.format(metric))


def part_distance_raw(qf, gf, qf_parts_visibility=None, gf_parts_visibility=None, dist_combine_strat='mean',
                      metric='euclidean', device=None, finalize=False, want_parts=True):
    """One launch of the distance kernel over (all queries) x (this gallery shard).  Returns device tensors
    (dist [Q,G], parts [P,Q,G], vmax [1] fp32): with `finalize=False` pairs without a shared visible part are still marked
    -1 and `vmax` holds the largest valid per-part distance of THIS shard -- a caller that shards the gallery reduces
    `vmax` (max) over the shards and then calls `fill_invalid` (distance.py:171-176, :214-216 take the max over the whole
    gallery).  `want_parts=False`: the [P,Q,G] per-part matrix is neither allocated nor written (`parts` is None) -- the
    q-q / g-g calls of the re-ranking and callers that only rank (9.6 GB at G = 20 000, P = 6 for the g-g matrix)."""
    _check_args(dist_combine_strat, metric)
    dev = device or (qf.device if qf.device.type == 'cuda' else torch.device('cuda', torch.cuda.current_device()))
    nv.init_device()
    mode = _vis_mode(qf_parts_visibility, gf_parts_visibility)
    qd = qf.to(dev, torch.float32).contiguous()
    gd = gf.to(dev, torch.float32).contiguous()
    q, p, d = qd.shape
    g = gd.shape[0]
    qv = qf_parts_visibility.to(dev, torch.float32).contiguous() if mode else None
    gv = gf_parts_visibility.to(dev, torch.float32).contiguous() if mode else None
    strat = 1 if (dist_combine_strat == 'max' and mode != 2) else 0      # continuous visibility: mean only (distance.py:200)
    parts = torch.empty(p, q, g, device=dev, dtype=torch.float32) if want_parts else None
    dist = torch.empty(q, g, device=dev, dtype=torch.float32)
    qsq = torch.empty(q * p, device=dev, dtype=torch.float32)
    gsq = torch.empty(g * p, device=dev, dtype=torch.float32)
    mx = torch.zeros(1, device=dev, dtype=torch.int32)
    nv.call('bpb_part_distance', qd.data_ptr(), gd.data_ptr(), nv.ptr(qv), nv.ptr(gv), q, g, p, d, mode, strat,
            1 if metric == 'cosine' else 0, qsq.data_ptr(), gsq.data_ptr(), mx.data_ptr(), nv.ptr(parts), dist.data_ptr(),
            1 if finalize else 0, nv.stream())
    return dist, parts, mx.view(torch.float32), mode


def fill_invalid(x, vmax):
    """-1 -> vmax + 1, in place (device tensor; `vmax` a one-element fp32 device tensor)."""
    nv.call('bpb_part_distance_fill', x.data_ptr(), x.numel(), vmax.data_ptr(), nv.stream())
    return x


def compute_distance_matrix_using_bp_features(qf, gf, qf_parts_visibility=None, gf_parts_visibility=None,
                                              dist_combine_strat='mean', batch_size_pairwise_dist_matrix=5000, use_gpu=True,
                                              metric='euclidean', device=None, return_device_tensors=False, want_parts=True):
    """`batch_size_pairwise_dist_matrix` is accepted for signature compatibility; 288 GB of HBM holds the whole
    [P,Q,G] result so the gallery is not chunked (results are identical: the reference's chunking only bounds memory).
    `want_parts=False` (extension): the per-part matrix is not produced and None is returned in its place."""
    dist, parts, _, _ = part_distance_raw(qf, gf, qf_parts_visibility, gf_parts_visibility, dist_combine_strat, metric, device,
                                          finalize=True, want_parts=want_parts)
    if return_device_tensors:
        return dist, parts
    return dist.cpu(), (parts.cpu() if parts is not None else None)


def evaluate_rank(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50, eval_metric='default', q_anns=None, g_anns=None,
                  use_cython=True, return_indices=False, nthreads=None):
    """market1501 protocol (rank.py:97-159) in native code, cuhk03 protocol (rank.py:17-94) on top of the native ranking.  Ties
    are broken by the lower gallery index (stable)."""
    if eval_metric == 'cuhk03':
        if isinstance(distmat, torch.Tensor):
            distmat = distmat.cpu().numpy()
        return _evaluate_cuhk03(distmat, q_pids, g_pids, q_camids, g_camids, max_rank, nthreads)
    if eval_metric != 'default':
        raise ValueError("Incorrect eval_metric value '{}'".format(eval_metric))
    if isinstance(distmat, torch.Tensor) and distmat.is_cuda:
        res = _evaluate_rank_gpu(distmat, q_pids, g_pids, q_camids, g_camids, max_rank)
        if res is not None:
            if return_indices:               # the index matrix of rank.py:110 from the GPU sort, handed out like the host one
                res['indices'] = argsort_rows_gpu(distmat).cpu().numpy()
            return res
    if isinstance(distmat, torch.Tensor):
        distmat = distmat.cpu().numpy()
    dm = np.ascontiguousarray(np.asarray(distmat, dtype=np.float32))
    nq, ng = dm.shape
    arr = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.int64))
    qp, gp, qc, gc = arr(q_pids), arr(g_pids), arr(q_camids), arr(g_camids)
    if ng < max_rank:
        max_rank = ng
    cmc = np.zeros(max_rank, dtype=np.float32)
    mAP = C.c_double(0.0)
    nvalid = C.c_int(0)
    idx = np.empty((nq, ng), dtype=np.int32) if return_indices else None
    pt = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    nthreads = nthreads or min(64, os.cpu_count() or 1)
    rc = nv.lib().bpb_eval_rank(pt(dm), pt(qp), pt(gp), pt(qc), pt(gc), nq, ng, max_rank, nthreads, pt(cmc), C.byref(mAP),
                                C.byref(nvalid), pt(idx))
    if rc == -2:
        raise AssertionError('Error: all query identities do not appear in gallery')
    nv.check(rc)
    res = {'cmc': cmc, 'mAP': float(mAP.value)}
    if return_indices:
        res['indices'] = idx
    return res


def _evaluate_rank_gpu(distmat, q_pids, g_pids, q_camids, g_camids, max_rank):
    """market1501 protocol for a distance matrix that lives in HBM (csrc/rank_gpu.hip).  Returns None when a query has more
    matching gallery entries than the kernel's LDS table holds (2048): the host routine then serves the call."""
    dm = distmat.to(torch.float32).contiguous()
    nq, ng = dm.shape
    if -(-ng // 64) * 8 > 32 * 1024:         # the kernel's LDS bit table holds 262 144 gallery entries: host routine beyond that
        return None
    dev = dm.device
    ids = [torch.as_tensor(np.asarray(a), dtype=torch.int64).to(dev) for a in (q_pids, g_pids, q_camids, g_camids)]
    max_rank = min(max_rank, ng)
    work = torch.empty(nq, device=dev, dtype=torch.float64)
    iwork = torch.empty(nq + 2, device=dev, dtype=torch.int32)
    cmc = torch.empty(max_rank, device=dev, dtype=torch.float32)
    mAP = torch.empty(1, device=dev, dtype=torch.float64)
    nv.call('bpb_eval_rank_gpu', dm.data_ptr(), ids[0].data_ptr(), ids[1].data_ptr(), ids[2].data_ptr(), ids[3].data_ptr(), nq, ng,
            max_rank, work.data_ptr(), iwork.data_ptr(), cmc.data_ptr(), mAP.data_ptr(), nv.stream())
    nvalid, overflow = [int(v) for v in iwork[nq:].tolist()]         # the one host sync of the evaluation
    if overflow:
        return None
    if nvalid == 0:
        raise AssertionError('Error: all query identities do not appear in gallery')
    return {'cmc': cmc.cpu().numpy(), 'mAP': float(mAP.item())}


def argsort_rows_gpu(distmat):
    """Row-wise stable argsort of a CUDA fp32 matrix (csrc/argsort_gpu.hip): int32 CUDA tensor [Q, G], ties by the lower index --
    np.argsort(distmat, axis=1, kind='stable') without leaving HBM (rank.py:110)."""
    dm = distmat.to(torch.float32).contiguous()
    nv.same_device(dm, 'argsort_rows_gpu')
    nq, ng = dm.shape
    need = C.c_long(0)
    nv.call('bpb_argsort_rows_gpu_workspace', nq, ng, C.byref(need))
    ws = torch.empty(need.value, device=dm.device, dtype=torch.uint8)
    idx = torch.empty(nq, ng, device=dm.device, dtype=torch.int32)
    nv.call('bpb_argsort_rows_gpu', dm.data_ptr(), nq, ng, idx.data_ptr(), ws.data_ptr(), need.value, nv.stream())
    return idx


def _native_argsort(dm, nthreads=None):
    """Row-wise stable argsort of a float32 matrix by the native ranking routine (csrc/rank.cpp)."""
    nq, ng = dm.shape
    idx = np.empty((nq, ng), dtype=np.int32)
    zeros = np.zeros(max(nq, ng), dtype=np.int64)
    cmc = np.zeros(1, dtype=np.float32)
    mAP, nvalid = C.c_double(0.0), C.c_int(0)
    pt = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = nv.lib().bpb_eval_rank(pt(dm), pt(zeros), pt(zeros), pt(zeros), pt(zeros + 1), nq, ng, 1, nthreads or min(64, os.cpu_count() or 1),
                                pt(cmc), C.byref(mAP), C.byref(nvalid), pt(idx))
    nv.check(rc)
    return idx


def _evaluate_cuhk03(distmat, q_pids, g_pids, q_camids, g_camids, max_rank, nthreads=None, repeats=10):
    """Single-gallery-shot protocol of rank.py:17-94.  The ranking comes from the native sort; per valid query, `repeats` times,
    one gallery image per identity is drawn from the GLOBAL numpy RNG with exactly the reference's call sequence
    (np.random.choice over the identity's positions, identities in order of first appearance in the ranking), so a seeded run
    reproduces the reference's numbers (golden vector 'cuhk03/*').  AP on the full ranking."""
    dm = np.ascontiguousarray(np.asarray(distmat, dtype=np.float32))
    nq, ng = dm.shape
    qp, gp = np.asarray(q_pids), np.asarray(g_pids)
    qc, gc = np.asarray(q_camids), np.asarray(g_camids)
    max_rank = min(max_rank, ng)
    order = _native_argsort(dm, nthreads)
    curve_sum = np.zeros(max_rank, dtype=np.float32)
    aps = []
    for i in range(nq):
        o = order[i]
        pid_ranked, cam_ranked = gp[o], gc[o]
        keep = ~((pid_ranked == qp[i]) & (cam_ranked == qc[i]))
        pids = pid_ranked[keep]
        hit = pids == qp[i]
        if not hit.any():
            continue
        # positions of every gallery identity (ascending), identities ordered by their first position in the ranking
        uniq, first, inverse = np.unique(pids, return_index=True, return_inverse=True)
        by_identity = np.argsort(inverse, kind='stable')
        counts = np.bincount(inverse, minlength=len(uniq))
        starts = np.concatenate(([0], np.cumsum(counts)[:-1]))
        if len(uniq) < max_rank:
            raise ValueError('cuhk03 protocol: query %d sees %d gallery identities, fewer than max_rank=%d (the reference fails '
                             'on the ragged CMC rows too)' % (i, len(uniq), max_rank))
        curve = np.zeros(max_rank, dtype=np.float32)
        for _ in range(repeats):
            chosen = np.empty(len(uniq), dtype=np.int64)
            for u in np.argsort(first, kind='stable'):
                chosen[u] = by_identity[starts[u] + np.random.choice(int(counts[u]))]
            chosen.sort()
            curve += np.minimum(np.cumsum(hit[chosen]), 1)[:max_rank].astype(np.float32)
        curve_sum += curve / repeats
        h = hit.astype(np.float64)
        aps.append(float((np.cumsum(h) / (np.arange(len(h)) + 1.0) * h).sum() / h.sum()))
    if not aps:
        raise AssertionError('Error: all query identities do not appear in gallery')
    return {'cmc': curve_sum / np.float32(len(aps)), 'mAP': float(np.mean(aps))}


def re_ranking_gpu(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3):
    """k-reciprocal re-ranking on the GPU (csrc/rerank_gpu.hip): CUDA float32 tensors in (the distance kernel's outputs stay in
    HBM), re-ranked [num_query, num_gallery] CUDA tensor out.  Dense (Q+G)^2 work matrices: 3 x 1.9 GB at Q + G = 22 048."""
    qg, qq, gg = [t.to(torch.float32).contiguous() for t in (q_g_dist, q_q_dist, g_g_dist)]
    if not (qg.is_cuda and qq.is_cuda and gg.is_cuda):
        raise ValueError('re_ranking_gpu: the three distance matrices must be CUDA tensors')
    nq, ng = qg.shape
    if qq.shape != (nq, nq) or gg.shape != (ng, ng):
        raise ValueError('re_ranking: expected q_q %s and g_g %s, got %s and %s' % ((nq, nq), (ng, ng), tuple(qq.shape), tuple(gg.shape)))
    fw, iw = C.c_long(0), C.c_long(0)
    nv.call('bpb_re_ranking_gpu_workspace', nq, ng, int(k1), int(k2), C.byref(fw), C.byref(iw))
    fwork = torch.empty(fw.value, device=qg.device, dtype=torch.float32)
    iwork = torch.empty(iw.value, device=qg.device, dtype=torch.int32)
    out = torch.empty(nq, ng, device=qg.device, dtype=torch.float32)
    nv.call('bpb_re_ranking_gpu', qg.data_ptr(), qq.data_ptr(), gg.data_ptr(), nq, ng, int(k1), int(k2), float(lambda_value),
            fwork.data_ptr(), iwork.data_ptr(), out.data_ptr(), nv.stream())
    return out


def re_ranking(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3, nthreads=None):
    """k-reciprocal re-ranking with the reference's signature (torchreid/utils/rerank.py:30; engine.py:433-437): numpy
    distance matrices in, re-ranked [num_query, num_gallery] float32 matrix out -- native threaded host routine
    (csrc/rerank.cpp).  CUDA tensors in -> the GPU kernels (re_ranking_gpu), CUDA tensor out."""
    if isinstance(q_g_dist, torch.Tensor) and q_g_dist.is_cuda:
        return re_ranking_gpu(q_g_dist, q_q_dist, g_g_dist, k1, k2, lambda_value)
    qg = np.ascontiguousarray(np.asarray(q_g_dist, dtype=np.float32))
    qq = np.ascontiguousarray(np.asarray(q_q_dist, dtype=np.float32))
    gg = np.ascontiguousarray(np.asarray(g_g_dist, dtype=np.float32))
    nq, ng = qg.shape
    if qq.shape != (nq, nq) or gg.shape != (ng, ng):
        raise ValueError('re_ranking: expected q_q %s and g_g %s, got %s and %s' % ((nq, nq), (ng, ng), qq.shape, gg.shape))
    out = np.empty((nq, ng), dtype=np.float32)
    pt = lambda a: a.ctypes.data_as(C.c_void_p)
    nthreads = nthreads or min(64, os.cpu_count() or 1)
    nv.check(nv.lib().bpb_re_ranking(pt(qg), pt(qq), pt(gg), nq, ng, int(k1), int(k2), float(lambda_value), int(nthreads), pt(out)))
    return out
