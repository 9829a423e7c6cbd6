"""Concurrency analysis of a rocprofv3 kernel trace (rocpd sqlite): GPU busy fraction, average number of kernels in flight,
and the per-kernel share of the *exclusive* (nothing else running) time.   python tools/timeline.py <results.db> [tail_frac]"""
import sqlite3
import sys
from collections import defaultdict

import numpy as np

c = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = list(c.execute('select start, end, name, queue_id from kernels order by start'))
s = np.array([r[0] for r in rows], dtype=np.int64)
e = np.array([r[1] for r in rows], dtype=np.int64)
t0 = s[int(len(s) * (1 - frac))]
idx = np.nonzero(s >= t0)[0]
ev = sorted([(s[i], 1, i) for i in idx] + [(e[i], -1, i) for i in idx])
active, last = set(), ev[0][0]
busy = 0
hist = defaultdict(int)
excl = defaultdict(int)
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        hist[len(active)] += dt
        if active:
            busy += dt
        if len(active) == 1:
            excl[rows[next(iter(active))][2].split('(')[0][:60]] += dt
    last = t
    if d > 0:
        active.add(i)
    else:
        active.discard(i)
wall = ev[-1][0] - ev[0][0]
print('kernels %d  queues %s' % (len(idx), sorted(set(rows[i][3] for i in idx))))
print('wall %.2f ms  busy %.2f ms (%.1f%%)  sum of durations %.2f ms' % (wall / 1e6, busy / 1e6, 100 * busy / wall, (e[idx] - s[idx]).sum() / 1e6))
for k in sorted(hist):
    print('  %d kernels in flight: %.2f ms (%.1f%%)' % (k, hist[k] / 1e6, 100 * hist[k] / wall))
print('exclusive time by kernel:')
for k, v in sorted(excl.items(), key=lambda kv: -kv[1])[:12]:
    print('  %-62s %.2f ms' % (k, v / 1e6))
