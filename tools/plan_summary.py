"""Aggregate a --dump-plan-timing file (bench.py) by kernel label: launches, ms, TFLOP/s or GB/s per plan (forward, backward,
forward_eval).  python tools/plan_summary.py plan_timing.json"""
import collections
import json
import sys

d = json.load(open(sys.argv[1]))
for name, rows in d.items():
    agg = collections.OrderedDict()
    for r in rows:
        lab = r['label']
        base = lab.rsplit(' x', 1)[0] if lab.rsplit(' x', 1)[-1].isdigit() else lab
        a = agg.setdefault(base, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += r['ms']
        a[2] += r['flops']
        a[3] += r['bytes']
    print('%s: %d launches, %.3f ms' % (name, len(rows), sum(v[1] for v in agg.values())))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if v[1] <= 0:
            continue
        rate = '%6.1f TF' % (v[2] / v[1] * 1e-9) if v[2] else '%6.0f GB/s' % (v[3] / v[1] * 1e-6)
        print('  %-72s n=%3d  %7.3f ms  avg %6.1f us  %s' % (k[:72], v[0], v[1], 1e3 * v[1] / v[0], rate))
