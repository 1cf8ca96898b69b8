set -o pipefail
mkdir -p gpurun_out/r03d
python -m pytest tests -m gpu -x -q > gpurun_out/r03d/pytest.log 2>&1; rc=$?; tail -12 gpurun_out/r03d/pytest.log; [ $rc -eq 0 ] || exit $rc
python bench.py --no-cpu-baseline --no-roofline --no-direct > gpurun_out/r03d/bench_f32.json 2> gpurun_out/r03d/bench_f32.err; tail -1 gpurun_out/r03d/bench_f32.err
MRDIS_UP2_STATS=0 python bench.py --no-cpu-baseline --no-roofline --no-direct > gpurun_out/r03d/bench_f32_nostats.json 2> gpurun_out/r03d/bench_f32_nostats.err; tail -1 gpurun_out/r03d/bench_f32_nostats.err
python bench.py --dtype bf16 --no-cpu-baseline --no-roofline > gpurun_out/r03d/bench_bf16.json 2> gpurun_out/r03d/bench_bf16.err; tail -1 gpurun_out/r03d/bench_bf16.err
MRDIS_UP2_STATS=0 python bench.py --dtype bf16 --no-cpu-baseline --no-roofline > gpurun_out/r03d/bench_bf16_nostats.json 2> gpurun_out/r03d/bench_bf16_nostats.err; tail -1 gpurun_out/r03d/bench_bf16_nostats.err
