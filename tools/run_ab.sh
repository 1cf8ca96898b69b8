#!/bin/bash
# Same-box A/B of one environment switch on the bench step: `bash tools/run_ab.sh MRDIS_GROUPED [bf16]` runs
# bench.py (step only) with VAR=1 / VAR=0 alternately and prints the timed ms/step of each run.
set -o pipefail
VAR=${1:-MRDIS_GROUPED}; DT=${2:-f32}
mkdir -p gpurun_out/ab
for v in 1 0 1 0; do
  env $VAR=$v python bench.py --dtype $DT --no-cpu-baseline --no-roofline --no-direct > gpurun_out/ab/${VAR}_${DT}_$v.json 2> gpurun_out/ab/${VAR}_${DT}_$v.err
  echo "$DT $VAR=$v $(grep -o 'timed: [0-9.]* ms' gpurun_out/ab/${VAR}_${DT}_$v.err)"
done
