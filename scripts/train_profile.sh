#!/bin/bash
# Kernel-time breakdown of the training step (rocprofv3 --kernel-trace --stats of scripts/train_bench.py).
#   bash scripts/train_profile.sh [head=timing] [batch=8] [steps=5]   -> gpurun_out/train_prof/kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/train_prof
rm -rf /tmp/prof_tr && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o run -- python3 $ROOT/scripts/train_bench.py ${1:-timing} ${2:-8} ${3:-5} > $OUT/bench.txt 2> $OUT/err.txt
cp $(find /tmp/prof_tr -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
tail -1 $OUT/bench.txt
python3 - <<P
import csv
rows=list(csv.DictReader(open('$OUT/kernel_stats.csv')))
steps=int('${3:-5}')+2
tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step %.2f' % (tot/1e6/steps))
for r in rows[:24]:
    print('%-64s %6d  %8.3f ms/step  %8.1f us avg  %5s%%' % (r['Name'][:64], int(r['Calls'])//steps, int(r['TotalDurationNs'])/1e6/steps, float(r['AverageNs'])/1e3, r['Percentage'][:5]))
P
