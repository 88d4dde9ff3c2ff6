#!/bin/bash
# ON THE GPU BOX: SQ counters of the FFT-domain conv kernels (separate --pmc passes, no tracing domains).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_fftconv; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pf_sq /tmp/pf_sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS \
  --output-format csv -d /tmp/pf_sq -- python3 $R/scripts/conv_microbench.py timing ${PMC_B:-512} 3 > $OUT/b.log 2> $OUT/b.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d /tmp/pf_sq2 -- python3 $R/scripts/conv_microbench.py timing ${PMC_B:-512} 3 > $OUT/b2.log 2> $OUT/b2.err
python3 - <<'PY' > $OUT/summary.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for d in ('/tmp/pf_sq', '/tmp/pf_sq2'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'fc_' not in k: continue
            k = k[:40]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
for k in acc:
    print(k, 'dispatches', n[k])
    for c, v in sorted(acc[k].items()): print('   %-26s %.4g  (per dispatch %.4g)' % (c, v, v / max(n[k], 1)))
PY
cat $OUT/summary.txt
