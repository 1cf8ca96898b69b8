#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the kernels of one layer of tools/layer_bench.py:
#   tools/pmc_hbm_layers.sh TAG "--dtype bf16 --only sp6.gamma" KERNEL_SUBSTR [KERNEL_SUBSTR ...]
set -e
TAG=$1; ARGS=$2; shift; shift
OUT=gpurun_out/pmch_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o f -- python3 tools/layer_bench.py $ARGS --iters 2 > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o w -- python3 tools/layer_bench.py $ARGS --iters 2 > $OUT/w.log 2>&1
for k in "$@"; do
  python tools/prof_summary.py pmc $OUT/f_counter_collection.csv $OUT/w_counter_collection.csv $OUT/${TAG}_$k.json $k ""
done
rm -f $OUT/*_kernel_trace.csv
