"""Data parallelism: one process per GPU, gradients averaged with an all-reduce over the flat gradient arena.

The reference's only multi-GPU mechanism is single-process nn.DataParallel (torchreid/scripts/main.py:257):
replicate + scatter + gather every step, loss on device 0.  On MI355X each rank owns its 64-image batch (its
own PxK sample, local BatchNorm statistics, per-rank triplet mining) and the ONE exchange
per step is the gradient all-reduce: torch.distributed backend "nccl" = RCCL over xGMI.  Because the gradients
already live in one contiguous fp32 arena there is no bucketing-by-parameter: the arena is cut into a few large
buckets (default 32 MiB, big enough to run the links at full rate, small enough to pipeline) issued
asynchronously on RCCL's stream.  The 1/world_size factor is folded into the fused Adam launch.
"""
import torch
import torch.distributed as dist


def broadcast_parameters(arena_tensors, src=0, group=None):
    """Identical initial weights / buffers on every rank (what DataParallel's replicate achieves)."""
    for t in arena_tensors:
        dist.broadcast(t, src=src, group=group)


class GradAllReducer:
    def __init__(self, flat_grad, group=None, bucket_bytes=32 << 20):
        self.flat, self.group = flat_grad, group
        n = flat_grad.numel()
        per = max(1, bucket_bytes // flat_grad.element_size())
        self.buckets = [(o, min(per, n - o)) for o in range(0, n, per)]
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._work = []

    def start(self):
        """Launch the all-reduce (sum) of every bucket; returns immediately."""
        self._work = []
        if self.world == 1:
            return
        for off, n in self.buckets:
            self._work.append(dist.all_reduce(self.flat[off:off + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for the buckets; returns the scale (1/world) the optimizer must apply to the summed gradient."""
        for w in self._work:
            w.wait()
        self._work = []
        return 1.0 / self.world
