"""GPU probe: round-off of the F(2,3) form of bpb_conv_s1 next to the direct form, forward and data gradient of one 3x3 stride-1
convolution + BatchNorm against fp64 (max and rms error relative to the largest reference value).
    python tools/wino_err.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import torch.nn.functional as F
from bpbreid_amd import native as nv
from bpbreid_amd.graph import Net, Act

dev = torch.device('cuda', 0)
nv.init_device()
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
nchw = lambda t: t.permute(0, 3, 1, 2)


def run(n, h, w, cin, cout, wino, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(n, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    gr = torch.randn(n, h, w, cout, generator=g)
    net = Net(dev)
    net.use_wino = wino
    xa = Act(net, n, h, w, cin)
    xa.needs_grad = True
    xa.buf.copy_(nhwc(x))
    wp = wt.to(dev)
    wp.grad = torch.zeros_like(wp)
    gamma, beta = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    gamma.grad, beta.grad = torch.zeros_like(gamma), torch.zeros_like(beta)
    node = net.conv(xa, wp, 1, 1, bn=(gamma, beta, torch.zeros(cout, device=dev), torch.ones(cout, device=dev)))
    out = net.fuse([(node, 0)], relu=False)
    net.finalize(train_backward=True)
    kinds = [m['label'] for m in net.plan_train[2] if m['label'].startswith('conv_fwd')] + [m['label'] for m in net.plan_bwd[2] if m['label'].startswith('conv_dgrad')]
    net.run(net.plan_train)
    out.grad.copy_(gr)
    net.run(net.plan_bwd)
    torch.cuda.synchronize()
    xr = x.double().requires_grad_(True)
    wr = wt.double()
    yr = F.conv2d(xr, wr, padding=1)
    o2 = F.batch_norm(yr, None, None, training=True, eps=1e-5)
    o2.backward(nchw(gr).double())
    err = lambda got, ref: ((got.double().cpu() - ref).abs().max() / ref.abs().max(), (got.double().cpu() - ref).pow(2).mean().sqrt() / ref.abs().max())
    return err(nchw(node.y.buf), yr.detach()), err(nchw(xa.grad), xr.grad), kinds


for shp in ((16, 32, 16, 32, 32), (16, 16, 8, 64, 64), (16, 8, 4, 128, 128), (16, 4, 2, 256, 256), (64, 64, 32, 32, 32), (64, 8, 4, 256, 256),
            (8, 2, 1, 64, 64), (8, 4, 2, 32, 32), (8, 3, 5, 64, 64), (24, 7, 3, 32, 32)):
    res = {}
    for wino in (False, True):
        res[wino] = run(*shp, wino)
    (fd, gd, kd), (fw, gw, kw) = res[False], res[True]
    print('%-22s fwd max %.2e rms %.2e -> %.2e %.2e (x%.2f, x%.2f) | dgrad max %.2e rms %.2e -> %.2e %.2e (x%.2f, x%.2f)  %s' % (
        shp, fd[0], fd[1], fw[0], fw[1], fw[0] / fd[0], fw[1] / fd[1], gd[0], gd[1], gw[0], gw[1], gw[0] / gd[0], gw[1] / gd[1],
        'F(2,3)' if any(',true>' in k for k in kw) else 'NOT the F(2,3) form: ' + kw[0]), flush=True)
