set -o pipefail
mkdir -p gpurun_out/r03k
python -m pytest tests/test_gpu_scale.py tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "bf16" > gpurun_out/r03k/pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r03k/pytest.log; [ $rc -eq 0 ] || exit $rc
python tools/layer_bench.py --dtype bf16 --only gamma,out,up_ > gpurun_out/r03k/lb_bf16.txt 2>&1; head -14 gpurun_out/r03k/lb_bf16.txt
MRDIS_DEBUG_NOPACK=1 python tools/layer_bench.py --dtype bf16 --only sp6.gamma,sp5.gamma,sp4.gamma > gpurun_out/r03k/lb_bf16_narrow.txt 2>&1; head -5 gpurun_out/r03k/lb_bf16_narrow.txt
python bench.py --dtype bf16 --no-cpu-baseline --no-roofline > gpurun_out/r03k/bench_bf16.json 2> gpurun_out/r03k/bench_bf16.err; tail -1 gpurun_out/r03k/bench_bf16.err
