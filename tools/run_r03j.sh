set -o pipefail
mkdir -p gpurun_out/r03j
python tools/layer_bench.py > gpurun_out/r03j/layer_bench_f32.txt 2>&1; head -14 gpurun_out/r03j/layer_bench_f32.txt; tail -1 gpurun_out/r03j/layer_bench_f32.txt
python tools/layer_bench.py --dtype bf16 > gpurun_out/r03j/layer_bench_bf16.txt 2>&1; tail -1 gpurun_out/r03j/layer_bench_bf16.txt
bash tools/pmc_sq.sh r03_f32 python3 tools/layer_bench.py --only sp6.gamma,sp5.gamma,sp4.gamma --iters 2 > gpurun_out/r03j/pmc_f32.txt 2>&1
bash tools/pmc_sq.sh r03_bf16 python3 tools/layer_bench.py --dtype bf16 --only sp6.gamma,sp5.gamma,sp4.gamma --iters 2 > gpurun_out/r03j/pmc_bf16.txt 2>&1
python tools/pmc_report.py gpurun_out/pmc_r03_f32 "fp32 path, round 3 (r03): Winograd filter images on (wino2_kernel<.., UIMG>)" > gpurun_out/r03j/pmc_report.md
python tools/pmc_report.py gpurun_out/pmc_r03_bf16 "compute_dtype bf16 (MRDIS_DT_BF16), round 3 (r03): bconv3_kernel / bwgrad2_kernel on bf16 views" >> gpurun_out/r03j/pmc_report.md
cat gpurun_out/r03j/pmc_report.md
