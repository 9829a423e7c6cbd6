"""Combine the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv counter_collection output) into per-kernel HBM bytes
per launch:  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950 correction, MI355X_MICROARCH.md section HBM).

    python tools/pmc_hbm.py <fetch dir> <write dir> profiles/r01_pmc_hbm.json
"""
import csv
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def norm(name):
    return name.replace('void ', '').split('(')[0].replace(' ', '')


def collect(root, counter):
    acc = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') == counter:
                    k = norm(row['Kernel_Name'])
                    acc[k][0] += 1
                    acc[k][1] += float(row['Counter_Value'])
    if files:
        return acc
    for f in glob.glob(os.path.join(root, '**', '*.db'), recursive=True):      # rocpd sqlite output (the default format)
        c = sqlite3.connect(f)
        names = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
        if 'counters_collection' not in names:
            continue
        cols = [r[1] for r in c.execute('pragma table_info(counters_collection)')]
        kcol = 'kernel_name' if 'kernel_name' in cols else [x for x in cols if 'kernel' in x and 'name' in x][0]
        ncol = 'counter_name' if 'counter_name' in cols else [x for x in cols if 'counter' in x and 'name' in x][0]
        vcol = 'value' if 'value' in cols else [x for x in cols if 'value' in x][0]
        # one row per (dispatch, counter instance): sum the instances of a dispatch, count dispatches
        dcol = 'dispatch_id' if 'dispatch_id' in cols else None
        if dcol:
            q = 'select %s, sum(%s) from counters_collection where %s = ? group by %s, %s' % (kcol, vcol, ncol, dcol, kcol)
        else:
            q = 'select %s, %s from counters_collection where %s = ?' % (kcol, vcol, ncol)
        for k, v in c.execute(q, (counter,)):
            acc[norm(k)][0] += 1
            acc[norm(k)][1] += float(v)
    return acc


fetch, write = collect(sys.argv[1], 'FETCH_SIZE'), collect(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in sorted(set(fetch) | set(write)):
    nf, sf = fetch.get(k, [0, 0.0])
    nw, sw = write.get(k, [0, 0.0])
    f_avg = sf / nf if nf else 0.0
    w_avg = sw / nw if nw else 0.0
    out[k] = {'launches': max(nf, nw), 'fetch_size_kb_per_launch': f_avg, 'write_size_kb_per_launch': w_avg,
              'hbm_bytes_per_launch': (2.0 * f_avg + w_avg) * 1024.0}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bpbreid_amd.build import source_id                       # noqa: E402
out['_build'] = {'source_id': source_id(), 'note': 'sha256 prefix of csrc/ + include/bpbreid_hip.h of the build these counters '
                                                   'were collected on; bench.py quotes `traffic` only when it matches its own build'}
json.dump(out, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
del out['_build']
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:10]:
    print('%-60s n=%5d  %.2f MB/launch' % (k[:60], v['launches'], v['hbm_bytes_per_launch'] / 1e6))
