set -o pipefail
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "wiener or Wiener or fusion or smoke or pipeline or image_processor or process_image_set or white_balance or bilateral_tile" > gpurun_out/ys2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/ys2_tests.log
timeout -k 10 200 python profiles/op_bench.py --only Wiener > gpurun_out/ys2_f16.json 2> gpurun_out/ys2_f16.err && \
timeout -k 10 200 python profiles/op_bench.py --only Wiener --storage f32 > gpurun_out/ys2_f32.json 2> gpurun_out/ys2_f32.err
tail -5 gpurun_out/ys2_tests.log
