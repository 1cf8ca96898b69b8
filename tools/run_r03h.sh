set -o pipefail
mkdir -p gpurun_out/r03h
python -m pytest tests/test_gpu_model.py tests/test_train_entry.py -m gpu -x -q > gpurun_out/r03h/pytest_model.log 2>&1; rc=$?; tail -8 gpurun_out/r03h/pytest_model.log; [ $rc -eq 0 ] || exit $rc
python bench.py --no-cpu-baseline --no-roofline --no-direct > gpurun_out/r03h/bench_f32.json 2> gpurun_out/r03h/bench_f32.err; tail -1 gpurun_out/r03h/bench_f32.err
MRDIS_GROUPED_ENC=0 python bench.py --no-cpu-baseline --no-roofline --no-direct > gpurun_out/r03h/bench_f32_off.json 2> gpurun_out/r03h/bench_f32_off.err; tail -1 gpurun_out/r03h/bench_f32_off.err
python bench.py --dtype bf16 --no-cpu-baseline --no-roofline > gpurun_out/r03h/bench_bf16.json 2> gpurun_out/r03h/bench_bf16.err; tail -1 gpurun_out/r03h/bench_bf16.err
MRDIS_GROUPED_ENC=0 python bench.py --dtype bf16 --no-cpu-baseline --no-roofline > gpurun_out/r03h/bench_bf16_off.json 2> gpurun_out/r03h/bench_bf16_off.err; tail -1 gpurun_out/r03h/bench_bf16_off.err
