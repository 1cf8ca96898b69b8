#!/bin/bash
# rocprofv3 kernel stats of tools/layer_bench.py for a few layers (run on the GPU box):  tools/prof_layers_one.sh TAG bf16|f32 layer,layer,...
TAG=$1; DT=$2; LAYERS=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/lb_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lb -- python3 tools/layer_bench.py --dtype $DT --only $LAYERS --iters 10 > $OUT/out.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/lb_kernel_stats.csv")))
for r in rows[:45]:
    print(f'{r["Name"][:120]:120s} {r["Calls"]:>6s} {float(r["TotalDurationNs"]) / 1e3:10.1f} us  avg {float(r["AverageNs"]) / 1e3:8.1f}')
PY
rm -f $OUT/lb_kernel_trace.csv
