set -o pipefail
mkdir -p gpurun_out/r03b
python tools/layer_bench.py --dtype bf16 > gpurun_out/r03b/layer_bench_bf16.txt 2>&1 && head -40 gpurun_out/r03b/layer_bench_bf16.txt &&
true
