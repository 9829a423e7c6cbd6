"""World-size-2 gloo tests (CPU) of the data-parallel exchange: broadcast of the arenas and the bucketed gradient
all-reduce that the engine uses with RCCL on the GPU box (bpbreid_amd/distributed.py is backend agnostic)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from bpbreid_amd.distributed import GradAllReducer, broadcast_parameters
    torch.manual_seed(100 + rank)
    params = torch.randn(10007)
    broadcast_parameters([params])
    g = torch.Generator().manual_seed(7 + rank)
    grad = torch.randn(10007, generator=g)
    red = GradAllReducer(grad, bucket_bytes=4096 * 4)        # several buckets + a ragged tail
    assert len(red.buckets) == 3 and sum(n for _, n in red.buckets) == 10007
    red.start()
    scale = red.finish()
    q.put((rank, params.numpy().copy(), grad.numpy().copy(), scale))     # numpy: no shared-memory handles across exit
    dist.destroy_process_group()


def test_gradient_allreduce_and_broadcast_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, g0, s0), (_, p1, g1, s1) = res
    p0, g0, p1, g1 = [torch.from_numpy(a) for a in (p0, g0, p1, g1)]
    assert torch.equal(p0, p1)                                         # identical replicas after the broadcast
    expect = torch.randn(10007, generator=torch.Generator().manual_seed(7)) + \
        torch.randn(10007, generator=torch.Generator().manual_seed(8))
    assert torch.allclose(g0, expect) and torch.equal(g0, g1)          # summed gradient, identical on both ranks
    assert s0 == s1 == 0.5                                             # the optimizer applies 1/world


def test_single_process_reducer_is_a_noop():
    from bpbreid_amd.distributed import GradAllReducer
    g = torch.arange(10.)
    red = GradAllReducer(g)
    red.start()
    assert red.finish() == 1.0 and torch.equal(g, torch.arange(10.))
