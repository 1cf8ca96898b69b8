set -o pipefail
mkdir -p gpurun_out/r03c
python -m pytest tests/test_gpu_scale.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -x -q -k "bf16 or grouped or mixing" > gpurun_out/r03c/pytest.log 2>&1; rc=$?; tail -12 gpurun_out/r03c/pytest.log; [ $rc -eq 0 ] || exit $rc
python tools/layer_bench.py --dtype bf16 --only si,1x1 > gpurun_out/r03c/layer_bench_bf16.txt 2>&1; tail -12 gpurun_out/r03c/layer_bench_bf16.txt
python bench.py --dtype bf16 --no-cpu-baseline --no-roofline > gpurun_out/r03c/bench_bf16.json 2> gpurun_out/r03c/bench_bf16.err; tail -2 gpurun_out/r03c/bench_bf16.err
