#!/bin/bash
# ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE of the STFT kernel at the step's batch (1024 windows of 516 frames), separate --pmc
# passes of scripts/stft_bench.py (nothing but --pmc), per dispatch.   bash scripts/pmc_stft.sh [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-run}; OUT=$R/gpurun_out/pmc_stft_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ps_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/ps_$c -- python3 $R/scripts/stft_bench.py 1024 > $OUT/$c.log 2> $OUT/$c.err
  python3 $R/scripts/summarize_pmc.py /tmp/ps_$c $c $OUT/$c.csv >> $OUT/$c.err 2>&1
done
python3 - <<PY > $OUT/pmc_stft_1024.txt
import csv
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for r in csv.DictReader(open('$OUT/%s.csv' % c)):
        if 'stft_mag_kernel<2048, false>' in r['kernel']:
            out[c] = float(r['%s_per_dispatch' % c]); out['n_' + c] = int(r['dispatches'])
# MI355X_MICROARCH.md (HBM section), as scripts/make_pmc_traffic.py applies it: both counters are in KB; on gfx950 FETCH_SIZE
# reports half of the bytes of wide coalesced reads and is doubled; WRITE_SIZE as is.  Bytes per window, in kB here:
rd, wr = 2 * out['FETCH_SIZE'] * 1024 / 1024 / 1e3, out['WRITE_SIZE'] * 1024 / 1024 / 1e3
print('stft_mag_kernel<2048, mag only>, 1024 windows x 516 frames, per dispatch (%d / %d dispatches):' % (out['n_FETCH_SIZE'], out['n_WRITE_SIZE']))
print('  read  %.3f MB per window (audio: 1.055)' % (rd / 1e3))
print('  write %.3f MB per window (magnitudes: 2.130 with the pad columns)' % (wr / 1e3))
print('  total %.3f MB per window (algorithmic 3.170)' % ((rd + wr) / 1e3))
PY
cat $OUT/pmc_stft_1024.txt
