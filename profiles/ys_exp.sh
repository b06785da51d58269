set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_gpu_tests.log
tail -4 gpurun_out/r3_gpu_tests.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err; tail -c 1500 gpurun_out/r3_bench_a.json
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --streams 1 --no-cpu-baseline > gpurun_out/r3_bench_a_s1.json 2> gpurun_out/r3_bench_a_s1.err
