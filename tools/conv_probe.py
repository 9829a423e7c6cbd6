"""Probe (GPU box): where does the time of the b1-shape conv go?  Variants of the same launch with parts disabled."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from bpbreid_amd import native as nv
from bpbreid_amd.graph import Net, Act

dev = torch.device('cuda', 0)
nv.init_device()
N = 64


def build(h, w, cin, cout, k, tile=None, dma=True):
    net = Net(dev)
    net.force_tile = tile
    net.use_dma = dma
    x = Act(net, N, h, w, cin)
    x.buf.normal_()
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wt.grad = torch.zeros_like(wt)
    node = net.conv(x, wt, 1, k // 2)
    net.finalize(False)
    return net


def time_conv(net, mutate=None, reps=30):
    prob, xb, wb, yb = net.debug_convs[0]
    import copy
    saved = bytes(C.string_at(C.addressof(prob), C.sizeof(prob)))
    if mutate:
        mutate(prob)
    # re-upload descriptor
    dev_t = [t for t in net.keep if isinstance(t, torch.Tensor) and t.dtype == torch.uint8 and t.numel() == C.sizeof(prob)][-1]
    dev_t.copy_(torch.frombuffer(bytearray(C.string_at(C.addressof(prob), C.sizeof(prob))), dtype=torch.uint8))
    conv_ops = [k for k, m in enumerate(net.plan_train[2]) if m['label'].startswith('conv_fwd')]
    arr = net.plan_train[0]
    op = arr[conv_ops[0]]
    one = (nv.PlanOp * 1)(op)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        nv.call('bpb_plan_run', C.cast(one, C.c_void_p), 1, nv.stream())
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        nv.call('bpb_plan_run', C.cast(one, C.c_void_p), 1, nv.stream())
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    C.memmove(C.addressof(prob), saved, len(saved))
    dev_t.copy_(torch.frombuffer(bytearray(saved), dtype=torch.uint8))
    return us, prob


for (h, w, cin, cout, k) in ((64, 32, 32, 32, 3), (64, 32, 64, 64, 3), (32, 16, 64, 64, 3)):
    for tile in (None, (2, 0, 1), (1, 0, 1), (2, 0, 2), (1, 0, 2), (1, 1, 1)):
        for dma in (True, False):
            try:
                net = build(h, w, cin, cout, k, tile, dma)
            except AssertionError as ex:
                continue
            p = net.debug_convs[0][0]
            if tile is not None and (p.mt_r, p.lwn, p.nt) != tile:
                continue
            full, _ = time_conv(net)
            notaps, _ = time_conv(net, lambda q: (setattr(q, 'Rt', 0), setattr(q, 'St', 0)))
            flops = 2.0 * N * h * w * k * k * cin * cout
            print('%dx%d %d->%d k%d tile(mt=%d,lwn=%d,nt=%d) CK=%d dma=%d blocks=%d: full %.1f us (%.1f TF)  no-mfma-loop %.1f us'
                  % (h, w, cin, cout, k, p.mt_r, p.lwn, p.nt, p.CK, p.dma, p.n_mtiles * p.n_ntiles, full, flops / full / 1e6, notaps))
