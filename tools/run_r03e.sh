set -o pipefail
mkdir -p gpurun_out/r03e
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "winograd_filter_image or wino" > gpurun_out/r03e/pytest_ops.log 2>&1; rc=$?; tail -5 gpurun_out/r03e/pytest_ops.log; [ $rc -eq 0 ] || exit $rc
python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "golden or mixing or grouped" > gpurun_out/r03e/pytest_model.log 2>&1; rc=$?; tail -5 gpurun_out/r03e/pytest_model.log; [ $rc -eq 0 ] || exit $rc
python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "winograd_vs_direct" > gpurun_out/r03e/pytest_scale.log 2>&1; rc=$?; tail -5 gpurun_out/r03e/pytest_scale.log; [ $rc -eq 0 ] || exit $rc
python bench.py --no-cpu-baseline --no-roofline --no-direct > gpurun_out/r03e/bench_f32.json 2> gpurun_out/r03e/bench_f32.err; tail -1 gpurun_out/r03e/bench_f32.err
MRDIS_WINO_U=0 python bench.py --no-cpu-baseline --no-roofline --no-direct > gpurun_out/r03e/bench_f32_nou.json 2> gpurun_out/r03e/bench_f32_nou.err; tail -1 gpurun_out/r03e/bench_f32_nou.err
python bench.py --no-cpu-baseline --no-roofline --no-direct > gpurun_out/r03e/bench_f32_b.json 2> gpurun_out/r03e/bench_f32_b.err; tail -1 gpurun_out/r03e/bench_f32_b.err
MRDIS_WINO_U=0 python bench.py --no-cpu-baseline --no-roofline --no-direct > gpurun_out/r03e/bench_f32_nou_b.json 2> gpurun_out/r03e/bench_f32_nou_b.err; tail -1 gpurun_out/r03e/bench_f32_nou_b.err
