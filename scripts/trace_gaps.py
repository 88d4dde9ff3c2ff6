#!/usr/bin/env python3
"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace run (largest first)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]))
rows.sort()
t_last = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3     # analyse the last fraction of the run
t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1 - t_last)
rows = [r for r in rows if r[0] >= t0]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print('kernels %d  span %.2f ms  busy %.2f ms  idle %.2f ms' % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
gaps = []
for a, b in zip(rows, rows[1:]):
    g = b[0] - a[1]
    if g > 0: gaps.append((g, a[2], b[2]))
gaps.sort(reverse=True)
from collections import defaultdict
agg = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    agg[(a, b)][0] += g; agg[(a, b)][1] += 1
for (a, b), (g, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print('%9.1f us  n=%4d  %-50s -> %s' % (g / 1e3, n, a, b))
