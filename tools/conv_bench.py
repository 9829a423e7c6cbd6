"""Micro-benchmark (GPU box): per-shape throughput of the conv kernels (forward, dgrad, wgrad) at N=64."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from bpbreid_amd import native as nv
from bpbreid_amd.graph import Net, Act

dev = torch.device('cuda', 0)
nv.init_device()
N = int(os.environ.get('NB', 64))
SHAPES = [  # H, W, Cin, Cout, k, stride
    (64, 32, 32, 32, 3, 1), (32, 16, 64, 64, 3, 1), (16, 8, 128, 128, 3, 1), (8, 4, 256, 256, 3, 1),
    (64, 32, 64, 64, 3, 1), (64, 32, 64, 256, 1, 1), (64, 32, 256, 64, 1, 1), (128, 64, 64, 64, 3, 2), (256, 128, 3, 64, 3, 2),
    (64, 32, 32, 64, 3, 2), (32, 16, 64, 32, 1, 1), (8, 4, 256, 1024, 1, 1), (16, 8, 128, 512, 1, 1),
]
if os.environ.get('ONLY') == '1x1':      # the stand-alone pointwise shapes (layer 1 of both backbones, ResNet-50 layer 2): BPB_CONV_PW=0/1
    SHAPES = [(64, 32, 64, 64, 1, 1), (64, 32, 64, 256, 1, 1), (64, 32, 256, 64, 1, 1), (64, 32, 256, 128, 1, 1), (32, 16, 128, 512, 1, 1),
              (32, 16, 256, 128, 1, 1), (32, 16, 128, 128, 1, 1)]
print('%-28s %9s %9s %9s   (TFLOP/s; us)' % ('shape', 'fwd', 'dgrad', 'wgrad'))
for (h, w, cin, cout, k, st) in SHAPES:
    net = Net(dev)
    cp = 4 if cin == 3 else cin
    x = Act(net, N, h, w, cp)
    x.needs_grad = cin != 3
    x.buf.normal_()
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wt.grad = torch.zeros_like(wt)
    g, b = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    g.grad, b.grad = torch.zeros_like(g), torch.zeros_like(b)
    node = net.conv(x, wt, st, k // 2, bn=(g, b, torch.zeros(cout, device=dev), torch.ones(cout, device=dev)))
    out = net.fuse([(node, 0)], relu=True)
    net.finalize(True)
    out.grad.normal_()
    net.run(net.plan_train); net.run(net.plan_bwd)
    torch.cuda.synchronize()
    acc = {}
    for rep in range(10):
        for plan in (net.plan_train, net.plan_bwd, net.plan_eval):
            for meta, ms in net.run_timed(plan):
                key = meta['label'].split(' ')[0] + ('_eval' if plan is net.plan_eval else '')
                a = acc.setdefault(key, [0.0, 0.0])
                a[0] += ms; a[1] += meta['flops'] if rep == 0 else 0
    def tf(key):
        if key not in acc: return '   -    '
        ms = acc[key][0] / 10
        return '%5.1f/%4.0f' % (acc[key][1] / (ms * 1e-3) / 1e12, ms * 1e3)
    extra = ' '.join('%s=%.0fus' % (k2, acc[k2][0] / 10 * 1e3) for k2 in ('fuse_fwd', 'bn_finalize', 'bn_bwd_reduce', 'bn_bwd_apply', 'wgrad_reduce', 'bn_bwd_finalize') if k2 in acc)
    extra += ' eval_fwd=' + tf('conv_fwd_eval').strip()
    print('%-28s %9s %9s %9s   %s' % ('%dx%d %d->%d k%d s%d' % (h, w, cin, cout, k, st), tf('conv_fwd'), tf('conv_dgrad'), tf('conv_wgrad'), extra))
