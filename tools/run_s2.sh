set -o pipefail
mkdir -p gpurun_out/s2
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stride2_weight_gradient or bf16_storage_random or conv_bf16_mfma" > gpurun_out/s2/t1.log 2>&1; echo "ops rc=$?"; tail -5 gpurun_out/s2/t1.log
timeout -k 10 900 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "zoo_at_bench_scale_bf16 and (down or mod)" > gpurun_out/s2/t2.log 2>&1; echo "zoo rc=$?"; tail -5 gpurun_out/s2/t2.log
python tools/layer_bench.py --dtype bf16 --only ana.down,mod.conv > gpurun_out/s2/lb.txt 2>&1; cat gpurun_out/s2/lb.txt
