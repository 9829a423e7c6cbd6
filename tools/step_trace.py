"""One train step as the GPU saw it: every kernel between the last two Adam launches of a rocprofv3 kernel trace (rocpd
sqlite), in launch order, with its start offset, duration and the idle gap before it; then the totals per region (forward
plan / head + losses / backward head / backward plan) so that latency chains between the big launches show up.
    python tools/step_trace.py <results.db> [out.txt]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
qcol = next((x for x in ('stream_id', 'queue_id', 'queue') if x in cols), None)       # which stream / queue a launch ran on
rows = list(c.execute('select start, end, name%s from kernels order by start' % (', ' + qcol if qcol else '')))
queues = sorted({r[3] for r in rows}) if qcol else []
adam = [i for i, r in enumerate(rows) if 'bpb_adam_kernel' in r[2]]
lo, hi = adam[-2] + 1, adam[-1] + 1
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
t0, prev = rows[lo][0], rows[lo][0]
busy = gap = 0
short = lambda n: n.split('(')[0].replace('void ', '')[:58]
per_q = {}
for row in rows[lo:hi]:
    s, e, n = row[:3]
    q = row[3] if qcol else 0
    out.write('%9.1f us  %7.1f us  gap %6.1f  %s%s\n' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3,
                                                        ('q%s  ' % queues.index(q)) if len(queues) > 1 else '', short(n)))
    busy += e - s
    gap += max(0, s - prev)
    prev = max(prev, e)
    per_q[q] = per_q.get(q, 0) + e - s
out.write('step wall %.2f ms  kernel time %.2f ms  idle gaps %.2f ms  launches %d\n' % ((prev - t0) / 1e6, busy / 1e6, gap / 1e6, hi - lo))
if len(per_q) > 1:
    # two streams (round 4: weight gradients beside the backward chain): kernel time per stream, and how much of it ran concurrently
    ev = sorted([(r[0], 1) for r in rows[lo:hi]] + [(r[1], -1) for r in rows[lo:hi]])
    depth, last, both, any_ = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth >= 2:
            both += t - last
        if depth >= 1:
            any_ += t - last
        depth += d
        last = t
    out.write('streams: %s  -- at least one kernel running %.2f ms, two or more %.2f ms (kernel time %.2f ms)\n'
              % ('  '.join('q%d %.2f ms' % (queues.index(q), v / 1e6) for q, v in sorted(per_q.items(), key=lambda kv: queues.index(kv[0]))),
                 any_ / 1e6, both / 1e6, busy / 1e6))
# the stretch between the forward plan's last launch (concat) and the backward plan's first convolution kernel
names = [short(r[2]) for r in rows[lo:hi]]
try:
    a = max(i for i, n in enumerate(names) if 'bilinear_concat_multi_fwd' in n)
    b = min(i for i, n in enumerate(names) if i > a and 'bilinear_concat_multi_bwd' in n)
    seg = rows[lo + a + 1:lo + b]
    out.write('head + losses + head backward: %d launches, wall %.2f ms, kernel time %.2f ms\n'
              % (len(seg), (seg[-1][1] - seg[0][0]) / 1e6, sum(r[1] - r[0] for r in seg) / 1e6))
except ValueError:
    pass

# the loss stretch (round-4 review, item 4): from the head's last forward launch (the slab reduce of the BN-neck classifiers' grouped GEMM,
# which follows the last BatchNorm1d forward) to the first grouped GEMM of the head's backward -- identity CE x3, part triplet, pixel CE, the
# weighted sum, its fan-out and the gradient scales.  It must hold library kernels only (no at::native::*, no runtime copy kernels).
foreign_kernel = lambda n: n.startswith('at::') or n.startswith('__amd') or 'elementwise_kernel' in n
try:
    a = max(i for i, n in enumerate(names) if 'bpb_bn1d_fwd' in n)
    a = min(i for i, n in enumerate(names) if i > a and 'bpb_gemm_reduce_grouped' in n)        # the forward classifiers are done
    b = min(i for i, n in enumerate(names) if i > a and 'bpb_gemm_grouped' in n)
    seg = rows[lo + a + 1:lo + b]
    foreign = [short(r[2]) for r in seg if foreign_kernel(short(r[2]))]
    out.write('loss stretch (behind the head\'s last forward launch .. first bpb_gemm_grouped of the backward): %d launches, wall %.3f ms, '
              'kernel time %.3f ms, kernels that are not the library\'s: %d %s\n'
              % (len(seg), (seg[-1][1] - seg[0][0]) / 1e6 if seg else 0.0, sum(r[1] - r[0] for r in seg) / 1e6, len(foreign), sorted(set(foreign))))
    whole = [short(r[2]) for r in rows[lo:hi] if foreign_kernel(short(r[2]))]
    out.write('whole step: %d launches that are not the library\'s: %s\n' % (len(whole), sorted(set(whole))))
except ValueError:
    pass
