"""GPU: weight-gradient kernels (csrc/wgrad16.hip second generation vs bpb_conv_wgrad_kernel of conv_igemm.hip) on the main 3x3
stride-1 HRNet shapes, each launch alone and the four-branch module step as ONE grouped launch, over tile / split settings."""
import os, sys, ctypes as C, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from bpbreid_amd import native as nv
from bpbreid_amd.graph import Net, Act
from bpbreid_amd import graph as G

dev = torch.device('cuda', 0)
nv.init_device()
N = 64
SHAPES = [(64, 32, 32, 32), (32, 16, 64, 64), (16, 8, 128, 128), (8, 4, 256, 256), (64, 32, 64, 64), (16, 8, 512, 512)]


def build(shapes, grouped, env):
    # lower-case keys are entries of graph.TUNE (tile / split constants), upper-case ones environment switches
    tune = {k: v for k, v in env.items() if k.islower()}
    env = {k: v for k, v in env.items() if not k.islower()}
    old = {k: os.environ.get(k) for k in env}
    old_tune = {k: G.TUNE[k] for k in tune}
    os.environ.update(env)
    G.TUNE.update({k: int(v) for k, v in tune.items()})
    try:
        net = Net(dev)
        net.use_wgrad16 = env.get('USE_WGRAD16', '1') != '0'      # (a plan attribute since round 5, no longer an environment switch)
        if grouped:
            net.fork(len(shapes))
        outs = []
        for i, (h, w, cin, cout) in enumerate(shapes):
            if grouped:
                net.set_slot(i)
            x = Act(net, N, h, w, cin)
            x.buf.normal_()
            wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
            wt.grad = torch.zeros_like(wt)
            node = net.conv(x, wt, 1, 1)
            node.y.ensure_grad(net).normal_()
            outs.append(node)
        if grouped:
            net.set_slot(0)
            net.join(len(shapes))
        # a consumer is needed for the backward walk: identity fuse per conv output
        net.finalize(True)
        return net
    finally:
        G.TUNE.update(old_tune)
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def timed(net, prefix, reps=20):
    arr, n, meta = net.plan_bwd
    ops = [i for i, m in enumerate(meta) if m['label'].startswith(prefix)]
    sel = (nv.PlanOp * len(ops))(*[arr[i] for i in ops])
    nv.call('bpb_plan_run', C.cast(sel, C.c_void_p), len(ops), nv.stream())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        nv.call('bpb_plan_run', C.cast(sel, C.c_void_p), len(ops), nv.stream())
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


CONFIGS = [('gen1', {'USE_WGRAD16': '0'}), ('gen1 blk256', {'USE_WGRAD16': '0', 'wgrad_blocks': '256'})]
for blk, tpb in itertools.product(('512', '256', '128', '64'), ('2', '4', '8')):
    CONFIGS.append(('g2 blk%s tpb%s' % (blk, tpb), {'wgrad16_blocks': blk, 'wgrad16_tpb': tpb}))

for group in [[s] for s in SHAPES] + [SHAPES[:4]]:
    flops = sum(2.0 * N * h * w * 9 * cin * cout for (h, w, cin, cout) in group)
    res = []
    for name, env in CONFIGS:
        try:
            net = build(group, len(group) > 1, env)
        except AssertionError as ex:
            continue
        wg = timed(net, 'conv_wgrad')
        rd = timed(net, 'wgrad_reduce')
        nsp = [p.nsplit for p, _ in net.debug_wgrads]
        res.append((wg + rd, wg, rd, name, nsp))
    res.sort()
    tag = ' + '.join('%dx%d %d->%d' % s for s in group)
    print('== %s   (%.2f GFLOP)' % (tag, flops / 1e9))
    for tot, wg, rd, name, nsp in res[:6] + [r for r in res if r[3].startswith('gen1')]:
        print('   %-26s wgrad %6.1f us (%5.1f TF)  reduce %5.1f us  total %6.1f us (%5.1f TF)  nsplit %s' % (name, wg, flops / wg * 1e-6, rd, tot, flops / tot * 1e-6, nsp), flush=True)
