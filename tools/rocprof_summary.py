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
