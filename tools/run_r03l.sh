set -o pipefail
mkdir -p gpurun_out/r03l
python -m pytest tests/test_gpu_scale.py tests/test_gpu_ops.py -m gpu -x -q -k "bf16" > gpurun_out/r03l/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/r03l/pytest.log; [ $rc -eq 0 ] || exit $rc
python tools/layer_bench.py --dtype bf16 > gpurun_out/r03l/lb_bf16.txt 2>&1; head -16 gpurun_out/r03l/lb_bf16.txt; tail -1 gpurun_out/r03l/lb_bf16.txt
python bench.py --dtype bf16 --no-cpu-baseline --no-roofline > gpurun_out/r03l/bench_bf16.json 2> gpurun_out/r03l/bench_bf16.err; tail -1 gpurun_out/r03l/bench_bf16.err
