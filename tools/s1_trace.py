"""GPU: phase timeline of the lean conv kernel inside ONE grouped launch (the four-branch HRNet module step), from a
measurement build of csrc/conv_s1.hip (-DBPB_S1_TRACE: every wave stamps s_memtime at entry / prologue done / first chunk
landed / MFMA loop done / exit and records the SIMD it ran on).  Answers: where does a wave's lifetime go, how many waves of
a SIMD are in their MFMA loop at the same time, how long is the launch's tail.

    python tools/s1_trace.py [x4|x3|x2|b0|b1|b2|b3 | l1a|l1b|l1c|l2a|l2b|l3a|l3b|l4a|l4b (1x1 shapes)] [tile mt,lwn,nt] [ck]
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
src = os.environ.get('S1_TRACE_SRC') or os.path.join(B.CSRC, 'conv_s1.hip')
if os.environ.get('S1_TRACE_SRC'):
    tag += '_' + os.path.splitext(os.path.basename(src))[0]
    trace_lib = os.path.join(objdir, 'libbpbreid_hip_trace%s.so' % tag)
    trace_obj = os.path.join(objdir, 'conv_s1_trace%s.o' % tag)
B.build()
if not os.path.exists(trace_obj) or os.path.getmtime(trace_obj) < os.path.getmtime(src):
    subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip',
                           '-DBPB_S1_TRACE', '-DS1_ABL=' + ABL, '-DS1_NA=' + NA, '-c', src, '-o', trace_obj, '-I', B.CSRC,
                           '-I', os.path.join(ROOT, 'include'), '-Wno-unused-value'])
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
ONE = {'l1a': (64, 32, 256, 64, 1), 'l1b': (64, 32, 64, 256, 1), 'l1c': (64, 32, 64, 64, 1),          # layer 1 (HRNet and ResNet-50)
       'l2a': (32, 16, 128, 512, 1), 'l2b': (32, 16, 512, 128, 1), 'l3a': (16, 8, 256, 1024, 1), 'l3b': (16, 8, 1024, 256, 1),
       'l4a': (16, 8, 512, 2048, 1), 'l4b': (16, 8, 2048, 512, 1)}                                      # ResNet-50 layers 2-4
shapes = {'x4': BR, 'x3': BR[:3], 'x2': BR[:2], 'b0': BR[:1], 'b1': BR[1:2], 'b2': BR[2:3], 'b3': BR[3:4],
          **{k_: [v_] for k_, v_ in ONE.items()}}[what]
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
nblk_grid = sum(p.n_mtiles * p.n_ntiles * (2 if p.split else 1) for p in probs)
buf = torch.zeros(nblk_grid * 4 * 16, dtype=torch.int64, device=dev)
lib = nv.lib()
lib.bpb_conv_s1_set_trace.argtypes = [C.c_void_p]
nv.check(lib.bpb_conv_s1_set_trace(buf.data_ptr()))
nv.call('bpb_plan_run', C.cast(one, C.c_void_p), 1, nv.stream())     # a traced launch right behind another one, like in the plan
nv.call('bpb_plan_run', C.cast(one, C.c_void_p), 1, nv.stream())
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nblk_grid * 4, 16)
nv.check(lib.bpb_conv_s1_set_trace(None))
t0, t1, t2, t3, t4, hw, xcc, pi_raw, r0, r4, t10, t11, t12 = [t[:, i].astype(np.int64) for i in range(13)]
pi = pi_raw & 0xff
first_half = (pi_raw & 0x100) != 0
# s_memtime (t0..t4) counts shader cycles from a base of its own per CU: only differences within a wave are used.  The launch
# timeline comes from s_memrealtime (r0 entry, r4 exit: the chip-wide 100 MHz clock), converted to shader cycles per wave.
CYC = 2400.0 / 100.0                      # nominal shader cycles per realtime tick (tools/mfma_peak.py: 2.39 GHz sustained)
start = (r0 - r0.min()) * CYC
end = (r4 - r0.min()) * CYC
span = end.max()
print('%s: %d workgroups, %.1f GFLOP, %.1f us per launch (%.1f TF) untraced; the traced launch spans %.1f us from the first wave entry to the last exit'
      % (what, nblk_grid, flops * 1e-9, us, flops / us * 1e-6, span / 2400.0))
print('tile/ck/blocks per problem:', [(p.mt_r, p.lwn, p.nt, p.CK, p.n_mtiles * p.n_ntiles * (2 if p.split else 1)) for p in probs])
pc = lambda a, q: np.percentile(a, q)
order = sorted(range(len(probs)), key=lambda k_: -(probs[k_].R * probs[k_].R * probs[k_].Cin * probs[k_].mt_r * probs[k_].nt))    # grid order: heaviest first
probs = [probs[k_] for k_ in order]
print('problem   waves   entry us (p5,p50,p95)   exit us (p50,p95,max)   prologue   1st-chunk wait   mfma loop   epilogue   lifetime   (shader cycles, medians; [p10..p90])')
for k in sorted(set(pi.tolist())):
    for fh in (True, False):
        m = (pi == k) & (first_half == fh)
        if not m.any():
            continue
        row = ['%4d(C%d%s)' % (k, probs[k].Cin, ' 1st half' if fh else ''), '%6d' % m.sum(),
               '%6.1f %6.1f %6.1f' % tuple(pc(start[m], q) / 2400.0 for q in (5, 50, 95)),
               '%6.1f %6.1f %6.1f' % (pc(end[m], 50) / 2400.0, pc(end[m], 95) / 2400.0, end[m].max() / 2400.0)]
        for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t0, t4)):
            d = (b - a)[m]
            row.append('%6d [%d..%d]' % (pc(d, 50), pc(d, 10), pc(d, 90)))
        print('  '.join(row))
print('prologue in detail (medians [p10..p90]): entry -> descriptor in SGPRs -> halo DMA offsets -> all offsets -> first DMA issued')
for k in sorted(set(pi.tolist())):
    m = (pi == k) & ~first_half
    row = ['%4d(C%d)' % (k, probs[k].Cin)]
    for a, b in ((t0, t10), (t10, t11), (t11, t1), (t1, t12)):
        d = (b - a)[m]
        row.append('%6d [%d..%d]' % (pc(d, 50), pc(d, 10), pc(d, 90)))
    print('  '.join(row))
# per-SIMD residency over the launch (realtime timeline)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = ((xcc & 15) << 12) | (se << 8) | (sh << 7) | (cu << 2) | simd
keys = np.unique(key)
print('%d distinct SIMDs hosted waves (1024 on the chip)' % len(keys))
grid = 400
edges = np.linspace(0, span, grid + 1)
occ_any = np.zeros(grid)
occ_mfma = np.zeros(grid)
# MFMA-loop interval of a wave on the realtime axis: entry + (t2 - t0) .. entry + (t3 - t0)
m2, m3 = start + (t2 - t0), start + (t3 - t0)
i0 = np.clip(np.searchsorted(edges, start) - 1, 0, grid)
i4 = np.searchsorted(edges, end)
j2 = np.clip(np.searchsorted(edges, m2) - 1, 0, grid)
j3 = np.searchsorted(edges, m3)
for a, b, c_, d_ in zip(i0, i4, j2, j3):
    occ_any[a:b] += 1
    if d_ > c_ and True:
        occ_mfma[c_:d_] += 1
nw = [int((key == kk).sum()) for kk in keys]
print('waves per SIMD over the launch: min %d median %d max %d' % (min(nw), np.median(nw), max(nw)))
print('timeline (launch span in 20 slices of %.1f us): mean resident waves per SIMD | mean waves inside their MFMA loop per SIMD' % (span / 2400.0 / 20))
for q in range(20):
    sl = slice(q * grid // 20, (q + 1) * grid // 20)
    print('  %3d%%  %.2f  %.2f' % (q * 5, occ_any[sl].mean() / 1024.0, occ_mfma[sl].mean() / 1024.0))
tot_life = float((end - start).sum())
print('sum of wave lifetimes / (1024 SIMDs x span) = %.2f resident waves per SIMD on average' % (tot_life / (1024.0 * span)))
# the MFMA loop's own efficiency: MFMAs * 64 cycles / loop duration, per problem
for k, p in enumerate(probs):
    m = (pi == k) & ~first_half
    mf = p.R * p.R * p.Cin // 2 * p.mt_r * p.nt // (2 if p.split else 1)
    d = (t3 - t2)[m]
    print('problem %d: %d MFMAs per wave = %d pipe cycles; median loop %d cycles -> one wave holds %.2f of its SIMD pipe while in the loop'
          % (k, mf, mf * 64, pc(d, 50), mf * 64 / pc(d, 50)))
