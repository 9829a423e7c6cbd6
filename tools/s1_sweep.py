"""GPU: the lean stride-1 conv kernel (csrc/conv_s1.hip) over every (wave tile, channel chunk) variant of the main HRNet /
ResNet shapes, each launch alone on the GPU, and the four-branch module step as ONE grouped launch.  Prints us and TFLOP/s."""
import os, sys, ctypes as C, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from bpbreid_amd import native as nv
from bpbreid_amd.graph import Net, Act

dev = torch.device('cuda', 0)
nv.init_device()
N = 64
SHAPES = [(64, 32, 32, 32, 3), (32, 16, 64, 64, 3), (16, 8, 128, 128, 3), (8, 4, 256, 256, 3), (64, 32, 64, 64, 3),
          (64, 32, 64, 256, 1), (64, 32, 256, 64, 1), (16, 8, 512, 2048, 1), (16, 8, 2048, 512, 1), (16, 8, 512, 512, 3)]
if len(sys.argv) > 1 and sys.argv[1] == '1x1':      # the 1x1 shapes of layer 1 (HRNet, ResNet-50) and of ResNet-50's layers 2-4, forward orientation
    SHAPES = [(64, 32, 64, 64, 1), (64, 32, 64, 256, 1), (64, 32, 256, 64, 1), (32, 16, 256, 128, 1), (32, 16, 128, 512, 1), (32, 16, 512, 128, 1),
              (16, 8, 512, 256, 1), (16, 8, 256, 1024, 1), (16, 8, 1024, 256, 1), (16, 8, 1024, 512, 1), (16, 8, 512, 2048, 1), (16, 8, 2048, 512, 1)]
TILES = [(1, 0, 1), (2, 0, 1), (1, 1, 1), (2, 1, 1), (1, 0, 2), (2, 0, 2), (1, 1, 2), (2, 1, 2)]
CKS = [32, 16, 8]


def timed(net, reps=20):
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
    return s.elapsed_time(e) * 1e3 / reps


def build(shapes, tile, ck, grouped):
    net = Net(dev)
    net.force_tile, net.force_ck = tile, ck
    if grouped:
        net.fork(len(shapes))
    for i, (h, w, cin, cout, k) in enumerate(shapes):
        if grouped:
            net.set_slot(i)
        x = Act(net, N, h, w, cin)
        x.buf.normal_()
        wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
        wt.grad = torch.zeros_like(wt)
        net.conv(x, wt, 1, k // 2)
    if grouped:
        net.set_slot(0)
        net.join(len(shapes))
    net.finalize(False)
    return net


for sh in SHAPES:
    h, w, cin, cout, k = sh
    flops = 2.0 * N * h * w * k * k * cin * cout
    res = {}
    for tile, ck in itertools.product(TILES, CKS):
        try:
            net = build([sh], tile, ck, False)
        except AssertionError:
            continue
        p = net.debug_convs[0][0]
        if not isinstance(p, nv.ConvS1Prob):
            continue
        cfg = (p.mt_r, p.lwn, p.nt, p.CK)
        if cfg in res:
            continue
        res[cfg] = timed(net)
    net = build([sh], None, None, False)
    p = net.debug_convs[0][0]
    dcfg = (p.mt_r, p.lwn, p.nt, p.CK)
    dus = timed(net)
    best = min(res.items(), key=lambda kv: kv[1])
    print('%3dx%-3d %4d->%-4d k%d  default %s %6.1f us (%5.1f TF) | best %s %6.1f us (%5.1f TF) | %s' % (
        h, w, cin, cout, k, dcfg, dus, flops / dus * 1e-6, best[0], best[1], flops / best[1] * 1e-6,
        ' '.join('%s:%.0f' % (''.join(map(str, c_)), u) for c_, u in sorted(res.items(), key=lambda kv: kv[1])[:(24 if len(sys.argv) > 1 else 8)])), flush=True)

if len(sys.argv) > 1:
    sys.exit(0)
# the four-branch module step as one grouped launch, uniform wave tile
MODULE = SHAPES[:4]
flops = sum(2.0 * N * h * w * k * k * cin * cout for (h, w, cin, cout, k) in MODULE)
for tile, ck in itertools.product([(1, 0, 1), (1, 1, 1), (2, 0, 1), (2, 1, 1)], [16, 8]):
    try:
        net = build(MODULE, tile, ck, True)
    except AssertionError as ex:
        print('grouped', tile, ck, 'infeasible', ex)
        continue
    cfgs = [(p.mt_r, p.lwn, p.nt, p.CK) for p, *_ in net.debug_convs]
    if len(net.plan_groups['train']) != 1 + 1:      # pack + one grouped conv
        print('grouped', tile, ck, 'not one launch:', [len(g) for g in net.plan_groups['train']], cfgs)
        continue
    us = timed(net)
    print('module step x4 grouped  tile %s ck %s -> %s  %6.1f us (%5.1f TF)' % (tile, ck, cfgs, us, flops / us * 1e-6), flush=True)
