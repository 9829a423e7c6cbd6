"""Turn a rocprofv3 (ROCm 7.x, rocpd sqlite output) result database into the --stats style per-kernel summary CSV.

    python tools/rocprof_summary.py gpurun_out/prof/r01_results.db profiles/r01_bench_kernel_stats.csv
"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute('select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc'))
with open(out, 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(['Name', 'Calls', 'TotalDurationUs', 'AverageUs', 'Percentage'])
    for r in rows:
        w.writerow([r[0], r[1], '%.3f' % r[2], '%.3f' % r[3], '%.4f' % r[4]])
print('wrote', out, len(rows), 'kernels; total GPU kernel time %.1f ms' % (sum(r[2] for r in rows) / 1e3))
if len(sys.argv) > 3:
    # optional third argument: per-kernel average durations as JSON, stamped with the build's source hash -- bench.py quotes
    # `roofline.frac_in_step` (the dominant kernel's duration INSIDE the two-stream step) from it when the hash is its own build's
    import json
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bpbreid_amd.build import source_id
    norm = lambda s_: s_.replace('void ', '').split('(')[0].replace(' ', '')
    js = {norm(r[0]): {'calls': r[1], 'average_us': r[3]} for r in rows}
    js['_build'] = {'source_id': source_id(), 'from': os.path.basename(out)}
    json.dump(js, open(sys.argv[3], 'w'), indent=1, sort_keys=True)

