#!/usr/bin/env python3
"""Per-kernel launch count / total / average duration from a rocprofv3 kernel trace dir."""
import csv, glob, os, sys
from collections import defaultdict
src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(src, '**', '*_kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name') or r.get('kernel_name')
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        a = acc[name]; a[0] += 1; a[1] += d
tot = sum(a[1] for a in acc.values()) or 1.0
with open(dst, 'w') as o:
    o.write('kernel,launches,total_us,avg_us,percent\n')
    for name, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        o.write('"%s",%d,%.1f,%.2f,%.2f\n' % (name, n, t, t / n, 100 * t / tot))
print('wrote', dst)
