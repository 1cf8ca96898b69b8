#!/bin/bash
set -e
TAG=${1:-r03b}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bf16 -- python bench.py --dtype bf16 --no-cpu-baseline --no-roofline > $OUT/bench_bf16_profiled.json 2> $OUT/bench_bf16_profiled.err
python tools/prof_summary.py stats $OUT/bf16_kernel_stats.csv $OUT/bf16_kernel_trace.csv $OUT/${TAG}_bench_bf16_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 ($TAG)"
python tools/dispatch_counts.py $OUT/bf16_kernel_stats.csv 7 > $OUT/${TAG}_dispatch_counts_bf16.txt
rm -f $OUT/*_kernel_trace.csv
