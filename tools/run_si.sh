set -o pipefail
mkdir -p gpurun_out/si
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "si_layer_weight or wgrad_c4 or stride2_weight" > gpurun_out/si/t1.log 2>&1; echo "ops rc=$?"; tail -5 gpurun_out/si/t1.log
timeout -k 10 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_model.py -x -q -m gpu -k "bf16" > gpurun_out/si/t2.log 2>&1; echo "bf16 rc=$?"; tail -5 gpurun_out/si/t2.log
bash tools/run_ab.sh MRDIS_NOTHING bf16
