#!/bin/bash
# One parameterised runner for the GPU box (through gpurun), replacing the per-round one-off scripts:
#   bash tools/run_round.sh TAG step [step ...]
# steps: tests | tests:<pytest -k expression> | smoke | bench | bench_fast | bench_bf16 | bench_drop | rehearsal | layers | layers_bf16 | ew | aten | stepprof
# Every step writes under gpurun_out/TAG/ and stops the script when it fails (no GPU step after a failed one).
set -o pipefail
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
for step in "$@"; do
  echo "== $step"
  case $step in
    tests)      python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log ;;
    tests:*)    python -m pytest tests -m gpu -x -q -k "${step#tests:}" > $OUT/pytest_k.log 2>&1; rc=$?; tail -15 $OUT/pytest_k.log ;;
    smoke)      python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?; tail -1 $OUT/smoke.log ;;
    bench)      python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err; rc=$?; grep "timed:" $OUT/bench_f32.err ;;
    bench_fast) python bench.py --no-cpu-baseline --no-roofline --no-direct > $OUT/bench_f32_fast.json 2> $OUT/bench_f32_fast.err; rc=$?; grep "timed:" $OUT/bench_f32_fast.err ;;
    bench_bf16) python bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; rc=$?; grep "timed:" $OUT/bench_bf16.err ;;
    bench_drop) python bench.py --drop --no-cpu-baseline --no-roofline --no-direct > $OUT/bench_drop.json 2> $OUT/bench_drop.err; rc=$?; grep "timed:" $OUT/bench_drop.err ;;
    rehearsal)  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/ddp_rehearsal.py > $OUT/ddp_rehearsal.txt 2>&1; rc=$?; grep "ddp rehearsal" $OUT/ddp_rehearsal.txt ;;
    layers)     python tools/layer_bench.py > $OUT/layer_bench_f32.txt 2>&1; rc=$?; tail -5 $OUT/layer_bench_f32.txt ;;
    layers_bf16) python tools/layer_bench.py --dtype bf16 > $OUT/layer_bench_bf16.txt 2>&1; rc=$?; tail -5 $OUT/layer_bench_bf16.txt ;;
    ew)         python tools/ew_bench.py > $OUT/ew_bench.txt 2>&1 && python tools/elem_once.py > $OUT/elem_once.txt 2>&1; rc=$?; tail -8 $OUT/elem_once.txt ;;
    aten)       HOSTPROF_SHAPE=32,4,256,256 python tools/aten_sources.py > $OUT/aten_sources.txt 2>&1; rc=$?; head -3 $OUT/aten_sources.txt ;;
    stepprof)   ( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o sp -- python bench.py --no-direct --no-cpu-baseline --no-roofline > $OUT/stepprof.json 2> $OUT/stepprof.err ); rc=$?
                [ $rc -eq 0 ] && python tools/prof_summary.py step $OUT/sp_kernel_trace.csv $OUT/${TAG}_last_step.txt.gz 7 && python tools/dispatch_counts.py $OUT/sp_kernel_stats.csv 7 > $OUT/${TAG}_dispatch_counts.txt; rm -f $OUT/sp_kernel_trace.csv; grep "timed:" $OUT/stepprof.err ;;
    stepprof_bf16) ( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o spb -- python bench.py --dtype bf16 --no-direct --no-cpu-baseline --no-roofline > $OUT/stepprof_bf16.json 2> $OUT/stepprof_bf16.err ); rc=$?
                [ $rc -eq 0 ] && python tools/prof_summary.py step $OUT/spb_kernel_trace.csv $OUT/${TAG}_last_step_bf16.txt.gz 7 && python tools/prof_summary.py stats $OUT/spb_kernel_stats.csv $OUT/spb_kernel_trace.csv $OUT/${TAG}_bench_bf16_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 --no-roofline ($TAG)" $OUT/stepprof_bf16.json && python tools/dispatch_counts.py $OUT/spb_kernel_stats.csv 7 > $OUT/${TAG}_dispatch_counts_bf16.txt; rm -f $OUT/spb_kernel_trace.csv; grep "timed:" $OUT/stepprof_bf16.err ;;
    bconv_abl)  python tools/bconv_abl.py > $OUT/bconv_abl.txt 2>&1; rc=$?; cat $OUT/bconv_abl.txt ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
  [ $rc -eq 0 ] || { echo "step $step failed (rc $rc)"; exit $rc; }
done
