"""GPU probe: one stride-1 conv shape on the lean kernel (csrc/conv_s1.hip), the default configuration or a forced (tile, chunk),
timed with events; run under `rocprofv3 --pmc ...` to get the SQ counters of exactly these launches.
    python tools/conv_pmc.py H W CIN COUT K [reps]        (CONV_PMC_TILE=mt,lwn,nt  CONV_PMC_CK=8|16|32 optional)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from bpbreid_amd import native as nv
from bpbreid_amd.graph import Net, Act

dev = torch.device('cuda', 0)
nv.init_device()
BR = [(64, 32, 32, 32, 3), (32, 16, 64, 64, 3), (16, 8, 128, 128, 3), (8, 4, 256, 256, 3)]       # the HRNet-W32 module step
if sys.argv[1] in ('x2', 'x3', 'x4'):       # python tools/conv_pmc.py x4 [reps]: the grouped launch of the first 2..4 branches
    shapes = BR[:int(sys.argv[1][1])]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
else:
    shapes = [tuple(int(a) for a in sys.argv[1:6])]
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
h, w, cin, cout, k = shapes[0]
N = 64
flops = sum(2.0 * N * h_ * w_ * k_ * k_ * ci_ * co_ for h_, w_, ci_, co_, k_ in shapes)
net = Net(dev)
if os.environ.get('CONV_PMC_NOPAD') == '0':    # every halo padded (LD = CK + 4): the bank-conflict counter's control
    net.s1_nopad = False
if os.environ.get('CONV_PMC_TILE'):
    net.force_tile = tuple(int(v) for v in os.environ['CONV_PMC_TILE'].split(','))
if os.environ.get('CONV_PMC_CK'):
    net.force_ck = int(os.environ['CONV_PMC_CK'])
standalone = os.environ.get('CONV_PMC_STANDALONE') == '1'      # a launch of its own (layer 1: 1x1 shapes then take bpb_conv_pw)
if not standalone:
    net.fork(max(2, len(shapes)))          # inside a fork region: the tile policy of the grouped module steps
for i, (h_, w_, ci_, co_, k_) in enumerate(shapes):
    if not standalone:
        net.set_slot(i)
    x = Act(net, N, h_, w_, ci_)
    x.buf.normal_()
    wt = torch.randn(co_, ci_, k_, k_, device=dev) * 0.05
    wt.grad = torch.zeros_like(wt)
    net.conv(x, wt, 1, k_ // 2)
if not standalone:
    net.set_slot(0)
    net.join(max(2, len(shapes)))
net.finalize(False)
p = (net.debug_convs or net.debug_pw)[0][0]
ops = [i for i, m in enumerate(net.plan_train[2]) if m['label'].startswith('conv_fwd')]
one = (nv.PlanOp * 1)(net.plan_train[0][ops[0]])
net.run(net.plan_train)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
if os.environ.get('CONV_PMC_COLD') == '1':
    # cache-cold timing: a 1 GB fill between the launches pushes the operands out of the 256 MB Infinity Cache (back-to-back
    # repetitions of one launch otherwise find a 134 MB input on die, which the launches of a real plan never do)
    scratch = torch.empty(1 << 28, device=dev, dtype=torch.float32)
    tot = 0.0
    for _ in range(reps):
        scratch.fill_(1.0)
        s.record()
        nv.call('bpb_plan_run', C.cast(one, C.c_void_p), 1, nv.stream())
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    us = tot * 1e3 / reps
else:
    s.record()
    for _ in range(reps):
        nv.call('bpb_plan_run', C.cast(one, C.c_void_p), 1, nv.stream())
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
if isinstance(p, nv.ConvPwProb):
    print('ConvPwProb NTC=%d column blocks=%d pixel groups=%d : %6.1f us  %5.1f TF' % (p.NTC, 1 << p.l_ntiles, p.n_mtiles, us, flops / us * 1e-6), flush=True)
else:
    print('%s tile mt=%d lwn=%d nt=%d CK=%d mtiles=%d ntiles=%d : %6.1f us  %5.1f TF' % (
        type(p).__name__, p.mt_r, p.lwn, p.nt, p.CK, p.n_mtiles, p.n_ntiles, us, flops / us * 1e-6), flush=True)
