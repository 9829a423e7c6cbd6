"""Gallery-sharded evaluation through the engine (launched by test_gpu_model.py under torch.distributed.run, 2 ranks sharing
the GPU over gloo, or nccl with one rank per GPU).

Every rank runs `engine.feature_extraction(query_batches, shard=True, gather=True)` (its share of the query batches, rows
all-gathered) and `engine.feature_extraction(gallery_batches, shard=True)` (its share of the gallery, rows stay local), then
`engine.evaluate(..., gallery_sharded=True)`: [Q, G_r] distance blocks, ONE scalar all-reduce for the fill value, all-gather
of the blocks and labels, ranking of the full matrix on every rank (part_based_engine.py:168-240 with the gallery sharded,
SURVEY.md section 8e).  Rank 0 repeats the whole evaluation in a single process without collectives: distance matrix, CMC and
mAP must be identical (the kernels are deterministic; an image's eval embedding does not depend on its batch).

Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', default='gloo')
    ap.add_argument('--backbone', default='hrnet_w8')
    ap.add_argument('--rerank', type=int, default=0)
    args = ap.parse_args()
    world, rank = int(os.environ['WORLD_SIZE']), int(os.environ['RANK'])
    local = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if args.backend == 'nccl':
        dist.init_process_group('nccl', device_id=dev)
    else:
        dist.init_process_group(args.backend)
    import common as Cm
    from bpbreid_amd.model import bpbreid
    from bpbreid_amd.engine import ImagePartBasedEngine
    from bpbreid_amd.optim import FusedAdam
    from bpbreid_amd.distributed import broadcast_parameters
    K, H, W, classes, B = 5, 64, 32, 12, 8
    model = Cm.fill_state_dict_(bpbreid(classes, config=Cm.make_cfg(args.backbone, K, 64), pretrained=False)).to(dev)
    arena = model.arena()
    broadcast_parameters([arena['param'], arena['fbuf']])

    def batches(nb, seed):       # the last batch is ragged: 5 gallery batches over 2 ranks = 3 + 2, 37 rows = 24 + 13
        out = []
        for b in range(nb):
            n = B if b + 1 < nb else B - 3
            imgs, masks, pids = Cm.synth_batch(n, H, W, K, classes, seed=seed + b)
            g = torch.Generator().manual_seed(seed + 100 + b)
            out.append({'image': imgs, 'mask': masks, 'pid': pids.tolist(), 'camid': torch.randint(0, 3, (n,), generator=g).tolist()})
        return out

    qb, gb = batches(3, 500), batches(5, 900)
    eng = ImagePartBasedEngine(model, optimizer=FusedAdam(model), distributed=True)
    qf, qv, qp, qc = eng.feature_extraction(qb, shard=True, gather=True)
    gf, gv, gp, gc = eng.feature_extraction(gb, shard=True)
    cmc, mAP, dm, _ = eng.evaluate(qf, gf, qv, gv, qp, gp, qc, gc, max_rank=10, rerank=bool(args.rerank), gallery_sharded=True)
    info = {'world': world, 'q_rows': int(qf.shape[0]), 'g_rows_local': int(gf.shape[0]), 'dm_shape': list(dm.shape)}
    # every rank must hold the same result
    t = torch.tensor([float(mAP), float(np.asarray(cmc).sum()), float(dm.double().sum())], dtype=torch.float64,
                     device=dev if args.backend == 'nccl' else 'cpu')
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    info['ranks_agree'] = bool(torch.equal(lo, hi))
    dist.barrier()
    if rank == 0:
        one = ImagePartBasedEngine(model, optimizer=FusedAdam(model), distributed=False)
        qf1, qv1, qp1, qc1 = one.feature_extraction(qb)
        gf1, gv1, gp1, gc1 = one.feature_extraction(gb)
        cmc1, mAP1, dm1, _ = one.evaluate(qf1, gf1, qv1, gv1, qp1, gp1, qc1, gc1, max_rank=10, rerank=bool(args.rerank))
        info.update(q_rows_single=int(qf1.shape[0]), g_rows_single=int(gf1.shape[0]),
                    features_equal=bool(torch.equal(qf, qf1)), labels_equal=bool(list(qp) == list(qp1) and list(qc) == list(qc1)),
                    dist_max_abs_diff=float((dm - dm1).abs().max()), dist_equal=bool(torch.equal(dm, dm1)),
                    cmc_equal=bool(np.array_equal(np.asarray(cmc), np.asarray(cmc1))), map_diff=abs(float(mAP) - float(mAP1)), mAP=float(mAP1))
        print(json.dumps(info))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
