set -o pipefail
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_scale.py --deselect tests/test_gpu_3d.py --deselect tests/test_gpu_ops.py > gpurun_out/r03a/pytest.log 2>&1; rc=$?; tail -15 gpurun_out/r03a/pytest.log; [ $rc -eq 0 ] || exit $rc
python bench.py --no-cpu-baseline > gpurun_out/r03a/bench_f32.json 2> gpurun_out/r03a/bench_f32.err && tail -3 gpurun_out/r03a/bench_f32.err &&
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r03a/bench_bf16.json 2> gpurun_out/r03a/bench_bf16.err && tail -3 gpurun_out/r03a/bench_bf16.err &&
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/ddp_rehearsal.py > gpurun_out/r03a/ddp_rehearsal.txt 2>&1; tail -8 gpurun_out/r03a/ddp_rehearsal.txt
