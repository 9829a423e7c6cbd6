"""Data parallelism: one process per GPU, gradients averaged with an all-reduce over the flat gradient arena.

The reference's only multi-GPU mechanism is single-process nn.DataParallel (torchreid/scripts/main.py:257):
replicate + scatter + gather every step, loss on device 0.  On MI355X each rank owns its 64-image batch (its
own PxK sample, local BatchNorm statistics, per-rank triplet mining) and the ONE exchange
per step is the gradient all-reduce: torch.distributed backend "nccl" = RCCL over xGMI.  Because the gradients
already live in one contiguous fp32 arena there is no bucketing-by-parameter: the arena is cut into a few large
buckets (default 16 MiB, big enough to run the links at full rate, small enough to pipeline) issued
asynchronously on RCCL's stream AS SOON AS the backward plan has enqueued the last launch that writes into a bucket
(model._ModelPlan.bucket_schedule): the exchange of the head / stage-4 gradients runs under the backward of stages 3..1.
The 1/world_size factor is folded into the fused Adam launch.
"""
import os

import torch
import torch.distributed as dist


def broadcast_parameters(arena_tensors, src=0, group=None):
    """Identical initial weights / buffers on every rank (what DataParallel's replicate achieves)."""
    for t in arena_tensors:
        dist.broadcast(t, src=src, group=group)


class GradAllReducer:
    def __init__(self, flat_grad, group=None, bucket_bytes=16 << 20):
        self.flat, self.group = flat_grad, group
        n = flat_grad.numel()
        per = max(1, bucket_bytes // flat_grad.element_size())
        self.buckets = [(o, min(per, n - o)) for o in range(0, n, per)]
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a world of one rank has nothing to exchange; BPB_EXCHANGE_WORLD1=1 runs the collectives anyway (functional check of the
        # RCCL call sequence on a one-GPU box: bench.py --force-dist)
        self.skip = self.world == 1 and not (dist.is_initialized() and os.environ.get('BPB_EXCHANGE_WORLD1', '0') == '1')
        self._work = []

    def begin(self):
        """Start of a backward pass: nothing is in flight, no bucket has been handed over yet."""
        self._work = []
        self._started = set()
        self.early_buckets = 0       # buckets handed over by the backward plan before its last launch (the overlapped ones)

    def ready(self, bucket_ids, early=True):
        """Buckets whose gradients are complete on the current stream: launch their all-reduce (sum) now.  async_op=True makes
        the collective's stream wait for everything enqueued so far on the current stream and returns immediately, so the
        remaining backward launches overlap with the exchange (over xGMI the 146 MB of an HRNet-W32 take ~1-1.5 ms)."""
        if self.skip:
            return
        for b in bucket_ids:
            if b in self._started:
                continue
            self._started.add(b)
            self.early_buckets += bool(early)
            off, n = self.buckets[b]
            self._work.append(dist.all_reduce(self.flat[off:off + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def start(self):
        """Launch the all-reduce (sum) of every bucket that has not been started by ready(); returns immediately."""
        if not hasattr(self, '_started'):
            self.begin()
        self.ready(range(len(self.buckets)), early=False)

    def finish(self):
        """Wait for the buckets; returns the scale (1/world) the optimizer must apply to the summed gradient."""
        for w in self._work:
            w.wait()
        self._work = []
        self._started = set()
        return 1.0 / self.world


# ---------------------------------------------------------------------------------------------------------------------
# Evaluation: the gallery is embarrassingly parallel.  Rank r holds gallery rows
# [gallery_shard(G, world, r)), computes the [Q, G_r] block of the part-based distance matrix against all queries, the
# "no shared visible part" fill value (global max + 1, distance.py:171) is agreed with ONE scalar all-reduce, and the
# blocks are all-gathered along the gallery axis.  No other exchange; the [P,Q,G_r] per-part blocks stay local.
def gallery_shard(n, world, rank):
    """Contiguous, balanced [begin, end) of rank `rank` (the first n % world ranks get one more row)."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def sharded_part_distance(qf, gf_local, qf_parts_visibility=None, gf_local_parts_visibility=None, dist_combine_strat='mean',
                          metric='euclidean', group=None, local_fn=None):
    """-> (distmat [Q, G] identical on every rank, local per-part block [P, Q, G_local]).

    `local_fn(qf, gf_local, qv, gv_local, strat, metric) -> (dist, parts, vmax[1] fp32, mode)` computes one shard with
    invalid pairs marked -1; default = the MI355X kernel (metrics.part_distance_raw).  The function is backend agnostic
    (RCCL on the GPU box, gloo in the CPU tests)."""
    if local_fn is None:
        from .metrics import part_distance_raw as local_fn
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    d, parts, vmax, mode = local_fn(qf, gf_local, qf_parts_visibility, gf_local_parts_visibility, dist_combine_strat, metric)
    if mode != 0:
        if world > 1:
            dist.all_reduce(vmax, op=dist.ReduceOp.MAX, group=group)
        if d.is_cuda:
            from .metrics import fill_invalid
            fill_invalid(d, vmax)
            if mode == 1 and parts is not None:
                fill_invalid(parts, vmax)
        else:
            d[d == -1] = vmax + 1
            if mode == 1 and parts is not None:
                parts[parts == -1] = vmax + 1
    if world == 1:
        return d, parts
    return all_gather_cat(d, dim=1, group=group), parts


def all_gather_cat(t, dim=0, group=None):
    """Concatenation along `dim` of the (differently sized along `dim`) tensors `t` of all ranks, identical on every rank.
    One size all-reduce + one all-gather of blocks padded to the largest shard.  RCCL moves device tensors directly; gloo
    (CPU tests, functional checks with ranks sharing a GPU) has no device all-gather: the block is staged through the host."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return t
    staged = t.is_cuda and dist.get_backend(group) == 'gloo'
    x = t.movedim(dim, 0).contiguous()
    x = x.cpu() if staged else x
    sizes = torch.zeros(world, dtype=torch.int64, device=x.device)
    sizes[dist.get_rank(group)] = x.shape[0]
    dist.all_reduce(sizes, group=group)
    sizes = sizes.tolist()
    pad = torch.zeros((max(sizes),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)     # all_gather needs equal shapes
    pad[:x.shape[0]] = x
    blocks = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(blocks, pad, group=group)
    full = torch.cat([b[:int(n)] for b, n in zip(blocks, sizes)], dim=0)
    return full.to(t.device).movedim(0, dim).contiguous()
