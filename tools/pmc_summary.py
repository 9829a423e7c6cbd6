"""Per-kernel average of a rocprofv3 --pmc counter (csv counter_collection output or the rocpd sqlite database).

    python tools/pmc_summary.py <rocprof output dir> <COUNTER> [out.csv]
"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict

root, counter = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else None
acc = defaultdict(lambda: [0, 0.0])
files = glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)
if files:
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') == counter:
                    a = acc[row['Kernel_Name']]
                    a[0] += 1
                    a[1] += float(row['Counter_Value'])
else:
    for f in glob.glob(os.path.join(root, '**', '*.db'), recursive=True):
        c = sqlite3.connect(f)
        names = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
        view = 'counters_collection' if 'counters_collection' in names else None
        if view is None:
            print('no counters_collection view; objects:', names)
            continue
        cols = [r[1] for r in c.execute('pragma table_info(%s)' % view)]
        kcol = 'kernel_name' if 'kernel_name' in cols else [x for x in cols if 'kernel' in x and 'name' in x][0]
        ncol = 'counter_name' if 'counter_name' in cols else [x for x in cols if 'counter' in x and 'name' in x][0]
        vcol = 'value' if 'value' in cols else [x for x in cols if 'value' in x][0]
        for k, v in c.execute('select %s, %s from %s where %s = ?' % (kcol, vcol, view, ncol), (counter,)):
            a = acc[k]
            a[0] += 1
            a[1] += float(v)
rows = sorted(((k, n, s, s / n) for k, (n, s) in acc.items()), key=lambda r: -r[2])
if out:
    with open(out, 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['Name', 'Dispatches', counter + '_total', counter + '_avg_per_dispatch'])
        w.writerows(rows)
for r in rows[:12]:
    print('%-70s n=%6d total=%.4e avg=%.4e' % (r[0][:70], r[1], r[2], r[3]))
