set -o pipefail
mkdir -p gpurun_out/r03b
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03b/gpu_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r03b/gpu_tests.log
tail -4 gpurun_out/r03b/gpu_tests.log
timeout -k 10 300 python profiles/op_bench.py --storage f16 --width 8192 --height 6144 --only "PPG|Wiener.process C=3" > gpurun_out/r03b/op_bench_50mp_f16.json 2>/dev/null; cat gpurun_out/r03b/op_bench_50mp_f16.json
timeout -k 10 300 python profiles/op_bench.py --storage f32 > gpurun_out/r03b/op_bench_f32.json 2>/dev/null
timeout -k 10 300 python profiles/bounds_probe.py > gpurun_out/r03b/bounds_probe.json 2>/dev/null; grep -A4 laplacian_12mp gpurun_out/r03b/bounds_probe.json
timeout -k 10 300 python profiles/fp16_chain_error.py > gpurun_out/r03b/fp16_chain_error.json 2>/dev/null; tail -c 300 gpurun_out/r03b/fp16_chain_error.json
