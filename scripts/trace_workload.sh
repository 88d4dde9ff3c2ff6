#!/bin/bash
# ON THE GPU BOX: kernel-trace summary of one workload.  bash scripts/trace_workload.sh c5 512
R=${GRAFT_REPO_ROOT:-$(pwd)}; WL=${1:-c5}; WIN=${2:-512}; OUT=$R/gpurun_out/trace_$WL; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/prof_wl
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_wl -- python3 $R/bench.py --workload $WL --windows $WIN --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.txt
python3 $R/scripts/summarize_trace.py /tmp/prof_wl $OUT/summary.csv >> $OUT/err.txt 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/summary.csv")))
for r in rows:
    if "at::native" in r["kernel"] or float(r["total_us"])<2000: continue
    print("%-62s n=%5s tot=%9.1f avg=%8.1f" % (r["kernel"][:62], r["launches"], float(r["total_us"]), float(r["avg_us"])))
PY
cut -c1-300 $OUT/bench.json
