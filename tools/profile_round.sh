#!/bin/bash
# Run on the GPU box (through gpurun): collects the rocprofv3 summaries that profiles/ keeps.
# Before the call: git rev-parse --short HEAD > profiles/HEAD_COMMIT (the snapshot carries no .git; the PMC summary records that id).
#   tools/profile_round.sh r04
# The training step is profiled WITHOUT the roofline legs (their ~500 extra Winograd launches used to inflate the step's
# kernel table, round-3 verdict); the legs get their own profile (roofline_*).
set -e
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python bench.py --no-direct --no-cpu-baseline --no-roofline > $OUT/${TAG}_bench_profiled.json 2> $OUT/bench_profiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bf16 -- python bench.py --dtype bf16 --no-cpu-baseline --no-roofline --no-graph-leg > $OUT/${TAG}_bench_bf16_profiled.json 2> $OUT/bench_bf16_profiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o roof -- python bench.py --roofline-only > $OUT/${TAG}_roofline_profiled.json 2> $OUT/roofline_profiled.err
python tools/prof_summary.py stats $OUT/bf16_kernel_stats.csv $OUT/bf16_kernel_trace.csv $OUT/${TAG}_bench_bf16_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 --no-roofline ($TAG)" $OUT/${TAG}_bench_bf16_profiled.json
python tools/prof_summary.py stats $OUT/roof_kernel_stats.csv $OUT/roof_kernel_trace.csv $OUT/${TAG}_roofline_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --roofline-only ($TAG): north-star conv + the step's dominant kernels, no training step"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ns -- python tools/northstar_conv.py 20 > $OUT/ns_stats.log 2>&1
python tools/prof_summary.py stats $OUT/ns_kernel_stats.csv $OUT/ns_kernel_trace.csv $OUT/${TAG}_northstar_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python tools/northstar_conv.py 20 ($TAG): north-star conv only"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o ns_fetch -- python tools/northstar_conv.py 10 > $OUT/ns_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o ns_write -- python tools/northstar_conv.py 10 > $OUT/ns_write.log 2>&1
python tools/prof_summary.py stats $OUT/bench_kernel_stats.csv $OUT/bench_kernel_trace.csv $OUT/${TAG}_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --no-roofline ($TAG): the training step only" $OUT/${TAG}_bench_profiled.json
python tools/prof_summary.py step $OUT/bench_kernel_trace.csv $OUT/${TAG}_bench_last_step.txt.gz 7
python tools/dispatch_counts.py $OUT/bench_kernel_stats.csv 7 > $OUT/${TAG}_dispatch_counts_f32.txt
python tools/dispatch_counts.py $OUT/bf16_kernel_stats.csv 7 > $OUT/${TAG}_dispatch_counts_bf16.txt
python tools/prof_summary.py pmc $OUT/ns_fetch_counter_collection.csv $OUT/ns_write_counter_collection.csv $OUT/northstar_conv_pmc.json c4conv "" $TAG
# the 3-D path: NVNet3D step under the profiler + the per-layer table with roofline fractions, six-product kernels on / off
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b3d -- python tools/bench3d.py > $OUT/${TAG}_bench3d_profiled.json 2> $OUT/bench3d.err
python tools/prof_summary.py stats $OUT/b3d_kernel_stats.csv $OUT/b3d_kernel_trace.csv $OUT/${TAG}_bench3d_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python tools/bench3d.py ($TAG): NVNet3D step on 4x4x128^3"
python tools/bench3d.py > $OUT/${TAG}_bench3d.json 2>> $OUT/bench3d.err
MRDIS_SPLIT6=0 python tools/bench3d.py > $OUT/${TAG}_bench3d_split6_off.json 2>> $OUT/bench3d.err
python tools/bench3d.py --layers > $OUT/${TAG}_bench3d_layers.txt 2>> $OUT/bench3d.err
# the graph-replayed step under the profiler (kernel time per step must equal the eager step's)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o graph -- python bench.py --graph --no-direct --no-cpu-baseline --no-roofline > $OUT/${TAG}_bench_graph_profiled.json 2> $OUT/bench_graph_profiled.err
python tools/dispatch_counts.py $OUT/graph_kernel_stats.csv 8 > $OUT/${TAG}_dispatch_counts_graph_f32.txt
rm -f $OUT/*_kernel_trace.csv $OUT/*_counter_collection.csv
ls -la $OUT
