"""GPU: phase timeline of the lean conv kernel inside ONE grouped launch (the four-branch HRNet module step), from a
measurement build of csrc/conv_s1.hip (-DBPB_S1_TRACE: every wave stamps s_memtime at entry / prologue done / first chunk
landed / MFMA loop done / exit and records the SIMD it ran on).  Answers: where does a wave's lifetime go, how many waves of
a SIMD are in their MFMA loop at the same time, how long is the launch's tail.

    python tools/s1_trace.py [x4|x3|x2|b0|b1|b2|b3] [tile mt,lwn,nt] [ck]
"""
import os, sys, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import torch
from bpbreid_amd import native as nv, build as B

objdir = os.path.join(os.path.dirname(B.LIB), 'build')
ABL = os.environ.get('S1_ABL', '0')          # ablation bit mask / S1_NA of the measurement build (csrc/conv_s1.hip)
NA = os.environ.get('S1_NA', '1')
tag = '' if (ABL, NA) == ('0', '1') else '_abl%s_na%s' % (ABL, NA)
trace_lib = os.path.join(objdir, 'libbpbreid_hip_trace%s.so' % tag)
trace_obj = os.path.join(objdir, 'conv_s1_trace%s.o' % tag)
src = os.path.join(B.CSRC, 'conv_s1.hip')
B.build()
if not os.path.exists(trace_obj) or os.path.getmtime(trace_obj) < os.path.getmtime(src):
    subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip',
                           '-DBPB_S1_TRACE', '-DS1_ABL=' + ABL, '-DS1_NA=' + NA, '-c', src, '-o', trace_obj, '-I', B.CSRC, '-Wno-unused-value'])
objs = [os.path.join(objdir, s.rsplit('.', 1)[0] + '.o') for s in B.SOURCES if s != 'conv_s1.hip'] + [trace_obj]
if not os.path.exists(trace_lib) or any(os.path.getmtime(o) > os.path.getmtime(trace_lib) for o in objs):
    subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', trace_lib] + objs)
nv.LIB_PATH = trace_lib
from bpbreid_amd.graph import Net, Act               # noqa: E402

dev = torch.device('cuda', 0)
nv.init_device()
N = 64
BR = [(64, 32, 32, 32, 3), (32, 16, 64, 64, 3), (16, 8, 128, 128, 3), (8, 4, 256, 256, 3)]
what = sys.argv[1] if len(sys.argv) > 1 else 'x4'
shapes = {'x4': BR, 'x3': BR[:3], 'x2': BR[:2], 'b0': BR[:1], 'b1': BR[1:2], 'b2': BR[2:3], 'b3': BR[3:4]}[what]
net = Net(dev)
if len(sys.argv) > 2 and sys.argv[2] != '-':
    net.force_tile = tuple(int(v) for v in sys.argv[2].split(','))
if len(sys.argv) > 3:
    net.force_ck = int(sys.argv[3])
net.fork(max(2, len(shapes)))
for i, (h, w, cin, cout, k) in enumerate(shapes):
    net.set_slot(i)
    x = Act(net, N, h, w, cin)
    x.buf.normal_()
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wt.grad = torch.zeros_like(wt)
    net.conv(x, wt, 1, k // 2)
net.set_slot(0)
net.join(max(2, len(shapes)))
net.finalize(False)
ops = [i for i, m in enumerate(net.plan_eval[2]) if m['label'].startswith('conv_fwd')]
assert len(ops) == 1, [m['label'] for m in net.plan_eval[2]]
one = (nv.PlanOp * 1)(net.plan_eval[0][ops[0]])
probs = [p for p, *_ in net.debug_convs][:len(shapes)]
nblk = sum(p.n_mtiles * p.n_ntiles for p in probs)
flops = sum(2.0 * N * h * w * k * k * cin * cout for (h, w, cin, cout, k) in shapes)
net.run(net.plan_eval)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    nv.call('bpb_plan_run', C.cast(one, C.c_void_p), 1, nv.stream())
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / 20
buf = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device=dev)
lib = nv.lib()
lib.bpb_conv_s1_set_trace.argtypes = [C.c_void_p]
nv.check(lib.bpb_conv_s1_set_trace(buf.data_ptr()))
nv.call('bpb_plan_run', C.cast(one, C.c_void_p), 1, nv.stream())     # a traced launch right behind another one, like in the plan
nv.call('bpb_plan_run', C.cast(one, C.c_void_p), 1, nv.stream())
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nblk * 4, 8)
nv.check(lib.bpb_conv_s1_set_trace(None))
t0, t1, t2, t3, t4, hw, xcc, pi = [t[:, i].astype(np.int64) for i in range(8)]
# every XCD has its own s_memtime base: times are taken relative to the first wave start ON THE SAME XCD (the dispatcher starts
# all XCDs within a fraction of a microsecond of each other)
xid = xcc & 15
for x_ in np.unique(xid):
    m_ = xid == x_
    b_ = t0[m_].min()
    for a_ in (t0, t1, t2, t3, t4):
        a_[m_] -= b_
base = 0
span = t4.max() - base
print('waves per XCD:', {int(x_): int((xid == x_).sum()) for x_ in np.unique(xid)}, ' end of the last wave per XCD:',
      {int(x_): int(t4[xid == x_].max()) for x_ in np.unique(xid)})
print('%s: %d workgroups, %.1f GFLOP, %.1f us per launch (%.1f TF) untraced; traced launch spans %d ticks'
      % (what, nblk, flops * 1e-9, us, flops / us * 1e-6, span))
print('tile/ck per problem:', [(p.mt_r, p.lwn, p.nt, p.CK, p.n_mtiles * p.n_ntiles) for p in probs])
ticks_per_us = span / us          # (the traced launch is a little slower than the untraced average: an estimate)
print('~%.0f ticks per us if the traced launch took the untraced time' % ticks_per_us)
pc = lambda a, q: np.percentile(a, q)
order = sorted(range(len(probs)), key=lambda k_: -(probs[k_].R * probs[k_].R * probs[k_].Cin * probs[k_].mt_r * probs[k_].nt))    # grid order: heaviest first
probs = [probs[k_] for k_ in order]
print('problem   waves   start(p50,p95)   end(p50,p95,max)   prologue   1st-chunk wait   mfma loop   epilogue   lifetime   (ticks, medians; [p10..p90])')
for k in sorted(set(pi.tolist())):
    m = pi == k
    row = ['%4d(C%d)' % (k, probs[k].Cin), '%6d' % m.sum(), '%7d %7d' % (pc(t0[m] - base, 50), pc(t0[m] - base, 95)),
           '%7d %7d %7d' % (pc(t4[m], 50), pc(t4[m], 95), t4[m].max())]
    for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t0, t4)):
        d = (b - a)[m]
        row.append('%6d [%d..%d]' % (pc(d, 50), pc(d, 10), pc(d, 90)))
    print('  '.join(row))
# per-SIMD concurrency of the MFMA loops
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = ((xcc & 15) << 12) | (se << 8) | (sh << 7) | (cu << 2) | simd
keys = np.unique(key)
print('%d distinct SIMDs hosted waves (1024 on the chip)' % len(keys))
grid = 400
occ_any = np.zeros(grid)
occ_mfma = np.zeros(grid)
edges = np.linspace(0, span, grid + 1)
busy_frac, mfma_frac, nw = [], [], []
for kk in keys:
    m = key == kk
    nw.append(m.sum())
    a_any = np.zeros(grid)
    a_m = np.zeros(grid)
    for w0, w2, w3, w4 in zip(t0[m] - base, t2[m] - base, t3[m] - base, t4[m] - base):
        i0, i4 = np.searchsorted(edges, [w0, w4])
        a_any[max(0, i0 - 1):i4] += 1
        i2, i3 = np.searchsorted(edges, [w2, w3])
        a_m[max(0, i2 - 1):i3] += 1
    occ_any += a_any
    occ_mfma += a_m
    busy_frac.append((a_any > 0).mean())
    mfma_frac.append((a_m > 0).mean())
print('waves per SIMD over the launch: min %d median %d max %d' % (min(nw), np.median(nw), max(nw)))
print('fraction of the launch span with >= 1 resident wave per SIMD: mean %.3f ; with >= 1 wave inside its MFMA loop: mean %.3f (min %.3f)'
      % (np.mean(busy_frac), np.mean(mfma_frac), np.min(mfma_frac)))
print('timeline (launch span in 20 slices): mean resident waves per SIMD | mean waves inside the MFMA loop per SIMD')
for q in range(20):
    sl = slice(q * grid // 20, (q + 1) * grid // 20)
    print('  %3d%%  %.2f  %.2f' % (q * 5, occ_any[sl].mean() / len(keys), occ_mfma[sl].mean() / len(keys)))
# the MFMA loop's own efficiency: MFMAs * 64 cycles / loop duration, per problem
for k, p in enumerate(probs):
    m = pi == k
    mf = p.R * p.R * p.Cin // 2 * p.mt_r * p.nt
    d = (t3 - t2)[m]
    print('problem %d: %d MFMAs per wave = %d pipe cycles; median loop %d ticks -> one wave holds %.2f of its SIMD pipe while in the loop'
          % (k, mf, mf * 64, pc(d, 50), mf * 64 / pc(d, 50)))
