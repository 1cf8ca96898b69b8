#!/bin/bash
# rocprofv3 kernel statistics of tools/layer_bench.py on a subset of layers:  tools/prof_layers.sh TAG DTYPE "sp1.,sp2."
set -e
TAG=${1:-layers}; DT=${2:-bf16}; ONLY=${3:-sp1.,sp2.}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lb -- python tools/layer_bench.py --dtype $DT --only "$ONLY" --iters 20 > $OUT/layer_bench.txt 2>&1
python tools/prof_summary.py stats $OUT/lb_kernel_stats.csv $OUT/lb_kernel_trace.csv $OUT/${TAG}_kernel_stats.md "layer_bench --dtype $DT --only $ONLY"
rm -f $OUT/*_kernel_trace.csv
