#!/bin/bash
# HBM-side traffic of the STFT kernel alone: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of scripts/stft_bench.py.
#   bash scripts/pmc_stft.sh [windows=256]   -> gpurun_out/pmc_stft/summary.txt
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_stft
mkdir -p $OUT; rm -rf /tmp/ps_f /tmp/ps_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/ps_f -- python3 $ROOT/scripts/stft_bench.py ${1:-256} > $OUT/f.log 2> $OUT/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/ps_w -- python3 $ROOT/scripts/stft_bench.py ${1:-256} > $OUT/w.log 2> $OUT/w.err
python3 - <<P > $OUT/summary.txt
import csv, glob
W=int('${1:-256}')
for tag, d in (('FETCH_SIZE', '/tmp/ps_f'), ('WRITE_SIZE', '/tmp/ps_w')):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'stft_mag_kernel' not in k: continue
        a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
    for k, (n, v) in acc.items():
        print('%s %-44s launches %d  KB per launch %.1f  MB per window %.3f%s' % (tag, k, n, v / n, v / n * 1024 / W / 1e6 * (2 if tag == 'FETCH_SIZE' else 1), ' (x2 gfx950 correction)' if tag == 'FETCH_SIZE' else ''))
P
cat $OUT/summary.txt
tail -2 $OUT/w.log
