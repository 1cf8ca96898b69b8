#!/bin/bash
# same-box A/B of two checkouts of the tree: the current one and an older one unpacked under ab/ (git archive <commit> | tar -x -C ab; make there)
set -o pipefail
DT=${1:-f32}
mkdir -p gpurun_out/oldnew
for r in 1 2; do
  (cd ab && python bench.py --dtype $DT --no-cpu-baseline --no-roofline --no-direct > ../gpurun_out/oldnew/old_$DT.json 2> ../gpurun_out/oldnew/old_$DT.err); echo "$DT old: $(grep -o 'timed: [0-9.]* ms' gpurun_out/oldnew/old_$DT.err)"
  python bench.py --dtype $DT --no-cpu-baseline --no-roofline --no-direct > gpurun_out/oldnew/new_$DT.json 2> gpurun_out/oldnew/new_$DT.err; echo "$DT new: $(grep -o 'timed: [0-9.]* ms' gpurun_out/oldnew/new_$DT.err)"
done
