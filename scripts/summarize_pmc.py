#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter per kernel name (one row per dispatch and counter)."""
import csv, glob, os, sys
from collections import defaultdict
src, counter, dst = sys.argv[1], sys.argv[2], sys.argv[3]
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') != counter:
            continue
        a = acc[r['Kernel_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
with open(dst, 'w') as o:
    o.write('kernel,dispatches,%s_sum,%s_per_dispatch\n' % (counter, counter))
    for name, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        o.write('"%s",%d,%.1f,%.1f\n' % (name, n, v, v / n))
print('wrote', dst)
