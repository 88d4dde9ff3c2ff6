#!/bin/bash
# ON THE GPU BOX: SQ counters of the conv kernels + GRBM_GUI_ACTIVE (separate --pmc passes, no
# tracing domains) -> MFMA utilisation and effective clock per kernel.
#   PMC_MODES=2 PMC_B=512 bash scripts/pmc_conv.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_conv; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pmc_sq /tmp/pmc_grbm /tmp/pmc_kt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS \
  --output-format csv -d /tmp/pmc_sq -- python3 $R/scripts/conv_microbench.py timing ${PMC_B:-128} ${PMC_MODES:-0,1} > $OUT/micro.log 2> $OUT/micro.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_grbm -- python3 $R/scripts/conv_microbench.py timing ${PMC_B:-128} ${PMC_MODES:-0,1} > $OUT/micro_grbm.log 2> $OUT/micro_grbm.err
rocprofv3 --kernel-trace --output-format csv -d /tmp/pmc_kt -- python3 $R/scripts/conv_microbench.py timing ${PMC_B:-128} ${PMC_MODES:-0,1} > $OUT/micro_kt.log 2> $OUT/micro_kt.err
python3 - <<'PY' > $OUT/sq_summary.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for d in ('/tmp/pmc_sq', '/tmp/pmc_grbm'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'conv_' not in k: continue
            k = k[:60]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
wall = defaultdict(float); nk = defaultdict(int)
for f in glob.glob('/tmp/pmc_kt/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if 'conv_' not in k: continue
        wall[k] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-9; nk[k] += 1
print('MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); clock = GRBM_GUI_ACTIVE / 8 / wall (un-profiled pass)')
for k in acc:
    a = acc[k]
    print(k, 'dispatches', n[k])
    for c, v in sorted(a.items()): print('   %-28s %.4g' % (c, v))
    if a.get('GRBM_GUI_ACTIVE') and a.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        cyc = a['GRBM_GUI_ACTIVE'] / 8.0
        print('   -> MFMA pipe busy %.1f %% of SIMD cycles' % (100.0 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc)))
        if wall.get(k): print('   -> effective clock %.2f GHz (kernel-trace wall %.3f ms over %d dispatches)' % (cyc / wall[k] / 1e9, wall[k] * 1e3, nk[k]))
# machine-readable copy for bench.py's roofline.from_profiles (the dominant kernel = most MFMA-busy cycles)
import json, os, subprocess
dom = max((k for k in acc if acc[k].get('GRBM_GUI_ACTIVE') and acc[k].get('SQ_VALU_MFMA_BUSY_CYCLES')),
          key=lambda k: acc[k]['SQ_VALU_MFMA_BUSY_CYCLES'], default=None)
if dom:
    a = acc[dom]; cyc = a['GRBM_GUI_ACTIVE'] / 8.0
    R = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
    commit = os.environ.get('AMT_HEAD_COMMIT')          # .git does not travel to the GPU box: the caller passes it
    d = dict(kernel=dom, mfma_pipe_busy=round(a['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc), 4),
             held_clock_ghz=round(cyc / wall[dom] / 1e9, 3) if wall.get(dom) else None,
             lds_wait_share=round(a.get('SQ_WAIT_INST_LDS', 0) / a['SQ_WAVE_CYCLES'], 4) if a.get('SQ_WAVE_CYCLES') else None,
             dispatches=n[dom], commit=commit,
             source='scripts/pmc_conv.sh: separate rocprofv3 --pmc passes (8 SQ counters; GRBM_GUI_ACTIVE) + a kernel-trace pass '
                    'of scripts/conv_microbench.py timing %s mode %s' % (os.environ.get('PMC_B', '128'), os.environ.get('PMC_MODES', '0,1')),
             formulas='busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); clock = GRBM_GUI_ACTIVE / 8 / kernel-trace wall')
    json.dump(d, open(os.path.join(R, 'gpurun_out', 'pmc_conv', 'pmc_derived.json'), 'w'), indent=1)
PY
cat $OUT/micro.log; cat $OUT/sq_summary.txt
