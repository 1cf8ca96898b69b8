set -o pipefail
mkdir -p gpurun_out/r03i
python -m pytest tests -m gpu -x -q > gpurun_out/r03i/pytest.log 2>&1; rc=$?; tail -8 gpurun_out/r03i/pytest.log; [ $rc -eq 0 ] || exit $rc
python bench.py --no-cpu-baseline > gpurun_out/r03i/bench_f32.json 2> gpurun_out/r03i/bench_f32.err; tail -2 gpurun_out/r03i/bench_f32.err | cut -c1-400
