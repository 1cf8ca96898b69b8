set -o pipefail
TAG=${1:-full}
mkdir -p gpurun_out/$TAG
python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; rc=$?; tail -4 gpurun_out/$TAG/pytest.log; [ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1; tail -1 gpurun_out/$TAG/smoke.log
python bench.py > gpurun_out/$TAG/bench_f32.json 2> gpurun_out/$TAG/bench_f32.err; grep "timed:" gpurun_out/$TAG/bench_f32.err
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/$TAG/bench_bf16.json 2> gpurun_out/$TAG/bench_bf16.err; grep "timed:" gpurun_out/$TAG/bench_bf16.err
