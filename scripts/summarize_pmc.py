#!/usr/bin/env python3
"""Per (kernel, grid size) class: dispatch count, sum and per-dispatch value of one rocprofv3 --pmc counter.

    python scripts/summarize_pmc.py <rocprof output dir> <COUNTER> <out.csv>

A kernel is launched with different grids inside one bench run (the guess bank's STFT, prepare()'s, the step's own;
a convolution's 20 x 516 layers and its small last layer), so per-kernel averages mix unequal launches: the rows here
keep them apart (Grid_Size = total work-items), and scripts/make_pmc_traffic.py picks the step's own class."""
import csv, glob, os, sys
from collections import defaultdict
src, counter, dst = sys.argv[1], sys.argv[2], sys.argv[3]
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') != counter:
            continue
        grid = r.get('Grid_Size') or r.get('Grid_Size_X') or '0'
        a = acc[(r['Kernel_Name'], int(float(grid)))]; a[0] += 1; a[1] += float(r['Counter_Value'])
with open(dst, 'w') as o:
    o.write('kernel,grid_size,dispatches,%s_sum,%s_per_dispatch\n' % (counter, counter))
    for (name, grid), (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        o.write('"%s",%d,%d,%.1f,%.1f\n' % (name, grid, n, v, v / n))
print('wrote', dst)
