#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:
#   bash scripts/collect_profiles.sh r01
# Produces small summaries under gpurun_out/profiles_<tag>/ (raw traces stay in /tmp):
#   kernel_stats.csv       rocprofv3 --kernel-trace --stats of the default bench.py (C3)
#   pmc_fetch.csv / pmc_write.csv   per-kernel FETCH_SIZE / WRITE_SIZE sums (separate --pmc passes)
#   bench_*.json           the bench lines printed under the profiler
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# one stream (the product's default): a kernel's duration in the trace is that of a kernel that owns the chip
export AMT_TIMING_STREAMS=1
cd /tmp
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- \
    python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_kernel_trace.json 2> $OUT/ks.err
cp /tmp/prof_ks/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
python3 $R/scripts/summarize_trace.py /tmp/prof_ks $OUT/kernel_trace_summary.csv >> $OUT/ks.err 2>&1
# counters in their own passes, nothing but --pmc (+ kernel names come with the counter csv)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- \
    python3 $R/bench.py --steps 1 --warmup 0 --windows 1024 --no-cpu-baseline --no-extras > $OUT/bench_pmc_fetch.json 2> $OUT/pf.err
python3 $R/scripts/summarize_pmc.py /tmp/prof_f FETCH_SIZE $OUT/pmc_fetch.csv >> $OUT/pf.err 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -- \
    python3 $R/bench.py --steps 1 --warmup 0 --windows 1024 --no-cpu-baseline --no-extras > $OUT/bench_pmc_write.json 2> $OUT/pw.err
python3 $R/scripts/summarize_pmc.py /tmp/prof_w WRITE_SIZE $OUT/pmc_write.csv >> $OUT/pw.err 2>&1
python3 $R/scripts/make_pmc_traffic.py $OUT/pmc_fetch.csv $OUT/pmc_write.csv 1024 $TAG > $OUT/pmc_traffic.log 2>&1
cp $R/profiles/pmc_traffic.json $OUT/pmc_traffic.json
# C5's per-GPU shard: 2048 windows, all heads, 5 iterations (kernel trace only)
rm -rf /tmp/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- \
    python3 $R/bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_c5_kernel_trace.json 2> $OUT/c5.err
cp /tmp/prof_c5/*/*_kernel_stats.csv $OUT/kernel_stats_c5.csv 2>/dev/null
ls -la $OUT
