#!/bin/bash
# ON THE GPU BOX: SQ counters + GRBM_GUI_ACTIVE + un-profiled wall time of every convolution-path kernel of the timing head
# (conv mode 3: fc_* row / GEMM kernels of the 20 x 516 layers, pk_* of the 10 x 64 layers, the direct split-fp16 classes),
# separate --pmc passes and a kernel-trace pass of scripts/chunk_probe.py (no tracing domains beside --pmc).
#   AMT_HEAD_COMMIT=<sha> PMC_B=1024 bash scripts/pmc_kernels.sh
# -> gpurun_out/pmc_kernels/{summary.txt, pmc_derived.json}; copy both to profiles/ (pmc_derived.json feeds bench.py's from_profiles).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_kernels; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pk_sq1 /tmp/pk_sq2 /tmp/pk_kt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS \
  --output-format csv -d /tmp/pk_sq1 -- python3 $R/scripts/chunk_probe.py ${PMC_B:-1024} > $OUT/b1.log 2> $OUT/b1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d /tmp/pk_sq2 -- python3 $R/scripts/chunk_probe.py ${PMC_B:-1024} > $OUT/b2.log 2> $OUT/b2.err
rocprofv3 --kernel-trace --output-format csv -d /tmp/pk_kt -- python3 $R/scripts/chunk_probe.py ${PMC_B:-1024} > $OUT/b3.log 2> $OUT/b3.err
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, json, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
name = lambda k: k.split('(')[0].replace('void ', '')
want = lambda k: any(s in k for s in ('fc_', 'pk_', 'conv_f16x3s', 'conv1_mfma'))
for d in ('/tmp/pk_sq1', '/tmp/pk_sq2'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = name(r['Kernel_Name'])
            if not want(k): continue
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
wall = defaultdict(float); nk = defaultdict(int)
for f in glob.glob('/tmp/pk_kt/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = name(r['Kernel_Name'])
        if want(k):
            wall[k] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-9; nk[k] += 1
out = {}
print('per kernel, summed over its dispatches of scripts/chunk_probe.py %s (timing head, conv mode 3); SQ_WAVE_CYCLES / SQ_WAIT_* / '
      'SQ_ACTIVE_INST_* count quad-cycles' % os.environ.get('PMC_B', '1024'))
for k in sorted(acc, key=lambda k: -wall.get(k, 0)):
    a = acc[k]
    print(k, 'dispatches', n[k], 'un-profiled wall %.3f ms per dispatch' % (wall[k] / max(nk[k], 1) * 1e3))
    for c, v in sorted(a.items()): print('   %-28s %.4g' % (c, v))
    wc = a.get('SQ_WAVE_CYCLES') or 0
    d = dict(dispatches=n[k], ms_per_dispatch=round(wall[k] / max(nk[k], 1) * 1e3, 4))
    if wc:
        d.update(issue_share=round(a['SQ_ACTIVE_INST_ANY'] / wc, 4), wait_any_share=round(a['SQ_WAIT_ANY'] / wc, 4),
                 wait_inst_share=round(a['SQ_WAIT_INST_ANY'] / wc, 4), valu_share=round(a['SQ_ACTIVE_INST_VALU'] / wc, 4),
                 lds_wait_share=round(a.get('SQ_WAIT_INST_LDS', 0) / wc, 4))
    if a.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_bank_conflict_share'] = round(a['SQ_LDS_BANK_CONFLICT'] / a['SQ_LDS_IDX_ACTIVE'], 4)
    if a.get('GRBM_GUI_ACTIVE'):
        cyc = a['GRBM_GUI_ACTIVE'] / 8.0
        if a.get('SQ_VALU_MFMA_BUSY_CYCLES'):
            d['mfma_pipe_busy'] = round(a['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc), 4)
        if wall.get(k) and nk[k] == n[k]:
            d['held_clock_ghz'] = round(cyc / wall[k] / 1e9, 3)
    print('   ->', json.dumps(d))
    out[k] = d
out['_provenance'] = dict(commit=os.environ.get('AMT_HEAD_COMMIT'), source='scripts/pmc_kernels.sh: two rocprofv3 --pmc passes (8 counters '
                          'each) + a kernel-trace pass of scripts/chunk_probe.py %s' % os.environ.get('PMC_B', '1024'),
                          formulas='shares = counter / SQ_WAVE_CYCLES; lds_bank_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; '
                                   'mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); held_clock = '
                                   'GRBM_GUI_ACTIVE / 8 / un-profiled wall')
json.dump(out, open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', os.getcwd()), 'gpurun_out', 'pmc_kernels', 'pmc_derived.json'), 'w'), indent=1)
PY
tail -5 $OUT/b1.log; head -60 $OUT/summary.txt
