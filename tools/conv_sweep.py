"""GPU: forward-conv time for every (tile shape, channel chunk) of the main HRNet shapes vs the heuristic's choice."""
import os, sys, ctypes as C, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from bpbreid_amd import native as nv
from bpbreid_amd.graph import Net, Act

dev = torch.device('cuda', 0)
nv.init_device()
N = 64
SHAPES = [(64, 32, 32, 32, 3, 1), (32, 16, 64, 64, 3, 1), (16, 8, 128, 128, 3, 1), (8, 4, 256, 256, 3, 1), (64, 32, 64, 64, 3, 1),
          (64, 32, 64, 256, 1, 1), (64, 32, 256, 64, 1, 1), (128, 64, 64, 64, 3, 2), (64, 32, 32, 64, 3, 2), (32, 16, 64, 128, 3, 2),
          (16, 8, 128, 512, 1, 1), (8, 4, 256, 1024, 1, 1), (32, 16, 64, 32, 1, 1), (16, 8, 128, 64, 1, 1)]
TILES = [None, (2, 0, 2), (2, 0, 1), (1, 0, 2), (1, 0, 1), (1, 1, 1)]
CKS = [None, 32, 16, 8]


def run(h, w, cin, cout, k, stride, tile, ck, reps=20):
    net = Net(dev)
    net.force_tile, net.force_ck = tile, ck
    x = Act(net, N, h, w, cin)
    x.buf.normal_()
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wt.grad = torch.zeros_like(wt)
    net.conv(x, wt, stride, k // 2)
    net.finalize(False)
    p = net.debug_convs[0][0]
    ops = [i for i, m in enumerate(net.plan_train[2]) if m['label'].startswith('conv_fwd')]
    one = (nv.PlanOp * 1)(net.plan_train[0][ops[0]])
    net.run(net.plan_train)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        nv.call('bpb_plan_run', C.cast(one, C.c_void_p), 1, nv.stream())
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps, (p.mt_r, p.lwn, p.nt, p.CK, p.dma)


for (h, w, cin, cout, k, st) in SHAPES:
    res = {}
    for tile, ck in itertools.product(TILES, CKS):
        try:
            us, cfg = run(h, w, cin, cout, k, st, tile, ck)
        except AssertionError:
            continue
        if (tile is None) != (ck is None) and not (tile is None and ck is None):
            pass
        res.setdefault(cfg, us)
        if tile is None and ck is None:
            default = (cfg, us)
    best = min(res.items(), key=lambda kv: kv[1])
    flops = 2.0 * N * (h // st) * (w // st) * k * k * cin * cout
    print('%3dx%-3d %4d->%-4d k%d s%d  default %s %6.1f us (%5.1f TF) | best %s %6.1f us (%5.1f TF) | %s' % (
        h, w, cin, cout, k, st, default[0], default[1], flops / default[1] * 1e-6, best[0], best[1], flops / best[1] * 1e-6,
        ' '.join('%s:%.0f' % (''.join(map(str, c_)), u) for c_, u in sorted(res.items(), key=lambda kv: kv[1])[:6])), flush=True)
