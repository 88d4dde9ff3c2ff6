#!/bin/bash
# ON THE GPU BOX: SQ counters of the conv kernels (one --pmc pass, no tracing domains)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_conv; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pmc_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS \
  --output-format csv -d /tmp/pmc_sq -- python3 $R/scripts/conv_microbench.py timing ${PMC_B:-128} ${PMC_MODES:-0,1} > $OUT/micro.log 2> $OUT/micro.err
python3 - <<'PY' > $OUT/sq_summary.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for f in glob.glob('/tmp/pmc_sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_' not in k: continue
        k = k[:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
for k in acc:
    print(k, 'dispatches', n[k])
    for c, v in sorted(acc[k].items()): print('   %-28s %.4g' % (c, v))
PY
cat $OUT/micro.log; cat $OUT/sq_summary.txt
